from . import trunc_normal_, trunc_normal_tf_  # noqa: F401
