"""Minimal stand-in for monai==1.4.0 (oracle tooling only). See ../README.md."""
