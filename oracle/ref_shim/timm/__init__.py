"""Minimal stand-in for timm==1.0.16 (oracle tooling only). See ../README.md."""
