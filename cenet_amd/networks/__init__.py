"""Drop-in for the reference's `networks` package (src/networks/__init__.py:1): `from networks import CENet`."""
from .cenet.net import CENet  # noqa: F401

# the reference's print_param_flops (utils/utils.py:171-181) runs fvcore's FlopCountAnalysis on the model: give its operator table
# an entry for the opaque forward operator (no-op when fvcore is not installed)
from .. import flops as _flops  # noqa: E402

_flops.register_with_fvcore()
