"""cenet_amd.ops — autograd operators of the CENet hot path, forward AND backward on hand-written HIP kernels.

One module per operator family (round 6: the former 3 100-line ops.py), imported here in dependency order; every public and private name is
re-exported, so `from cenet_amd import ops; ops.linear(...)` / `ops._WgradCfg` / `ops._ZeroWs` read as they always did:
  ops.infra          shared plumbing of the operator modules: tensor allocation helpers, zero-at-rest workspaces, parameter / gradient views into the
  ops.linear         GEMM-backed operators: Linear / MultiLinear in token layout, 1x1 convolutions on NCHW, dense k x k convolutions (implicit GEMM, direct
  ops.norm           LayerNorm over the last dimension and train-mode BatchNorm (+ fused activation)
  ops.depthwise      depthwise 3x3 convolutions (token layout and NCHW) and the fused PVT Mlp half (csrc/pvt_mlp.hip)
  ops.attention      attention operators: spatial-reduction attention, Non-local attention, differential attention heads + combine
  ops.glue           layout / glue operators: token <-> NCHW, concat / split (+ merged depthwise branches), grouped 1x1, add + activation, SiLU product,
  ops.decoder_fused  resampling operators and the channel-local fused chains of the decoder and the head (csrc/chanloc.hip, csrc/res_tail.hip): EUCB front,
  ops.gates          CCU and SRM gates (cfam.py:251-264, 93-101)
  ops.dseb_loss      DSEB combine (dseb.py:40-50,63-76,156-163) and the fused segmentation loss (utils/core.py:44-131,161-188)
"""
from .infra import *  # noqa: F401,F403
from .linear import *  # noqa: F401,F403
from .norm import *  # noqa: F401,F403
from .depthwise import *  # noqa: F401,F403
from .attention import *  # noqa: F401,F403
from .glue import *  # noqa: F401,F403
from .decoder_fused import *  # noqa: F401,F403
from .gates import *  # noqa: F401,F403
from .dseb_loss import *  # noqa: F401,F403

__all__ = [n for n in dir() if not n.startswith("__")]
