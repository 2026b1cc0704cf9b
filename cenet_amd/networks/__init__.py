"""Drop-in for the reference's `networks` package (src/networks/__init__.py:1): `from networks import CENet`."""
from .cenet.net import CENet  # noqa: F401
