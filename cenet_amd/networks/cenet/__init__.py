from .net import CENet  # noqa: F401
