from . import DropPath  # noqa: F401
