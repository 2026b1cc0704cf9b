"""CENet — mirrors reference src/networks/cenet/net.py:8-64 (same ctor signature, attributes and state-dict keys)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import kern
from .decoders import Decoder
from .encoder import get_encoder2d
from .out import OutHead


class CENet(nn.Module):
    def __init__(self, input_channels=1, num_classes=1, scale_factors=[0.8, 0.4], diffatt_num_heads=[2, 2, 2],
                 encoder='pvt_v2_b2', enc_pretrain=False, freeze_bb=False, skip_mode="cat", dec_up_block='eucb',
                 out_merge_mode="cat", out_up_block="eucb", out_up_ks=3, writer=None, base_ptdir='.'):
        super().__init__()
        self.writer = writer
        self.backbone, channels = get_encoder2d(input_channels=input_channels, encoder=encoder, pretrain=enc_pretrain,
                                                freeze_bb=freeze_bb, base_ptdir=base_ptdir)
        self.decoder = Decoder(channels=channels, scale_factors=scale_factors, skip_mode=skip_mode,
                               num_heads=diffatt_num_heads, up_block=dec_up_block, writer=writer)
        self.out = OutHead(dec_in_spatial=56, dec_in_channels=channels[-1], x_in_spatial=224,
                           x_in_channels=input_channels, out_channels=num_classes, merge_mode=out_merge_mode,
                           up_block=out_up_block, up_ks=out_up_ks)

    def forward(self, x):
        if torch.jit.is_tracing():
            # utils/utils.py:171-185 (print_param_flops, called on the model at main_acdc.py:128) runs fvcore's FlopCountAnalysis,
            # a torch.jit.trace of the model.  A tracer turns tensor sizes into traced values (which cannot be passed to the C
            # ABI) and could only record the custom kernels as opaque calls anyway: the forward runs with tracing suspended and
            # the result is tied to the input by a zero-valued traced term, so the trace completes (the parameter count it
            # reports is right, its FLOP count is ~0 — INTEGRATION.md).
            state = torch._C._get_tracing_state()
            torch._C._set_tracing_state(None)
            try:
                out = self._forward(x.detach())
            finally:
                torch._C._set_tracing_state(state)
            return out + torch.zeros_like(x).sum().to(out.dtype)
        return self._forward(x)

    def _forward(self, x):
        # throughput mode (kern.set_compute_bf16): the whole network runs on bf16 tensors — the input is rounded once here
        # and every kernel downstream follows the element type of its input; logits come back as bf16
        # The reference's AMP switch (main_acdc.py:192-199,243-249: `with autocast('cuda')` + GradScaler) selects the same
        # mode: under an enabled CUDA autocast region the forward runs on bf16 tensors (fp32 islands as in
        # multihead_diffattn.py:108, rms_norm.py:19: softmax / norm statistics / accumulators; parameters stay fp32).  bf16 has
        # fp32's exponent range, so the GradScaler the caller wraps around the step is harmless but not needed.
        if torch.compiler.is_compiling():
            # main_acdc.py:188-191 (--compile): the operators below are ctypes launches of the HIP library inside
            # autograd.Functions; a tracing compiler cannot see through them and fullgraph=True cannot be honoured
            raise RuntimeError("cenet_amd.CENet does not support torch.compile: its operators are hand-written HIP kernels "
                               "launched through a C ABI (capture the step with cenet_amd.graph.GraphedStep instead)")
        amp = x.is_cuda and torch.is_autocast_enabled("cuda")
        if (kern.get_compute_bf16() or amp) and x.dtype == torch.float32:
            x = kern.cast(x, torch.bfloat16)
        # grayscale input: the 3-channel replication of net.py:55 is a zero-stride channel read in patch_embed1
        x1, x2, x3, x4 = self.backbone(x)
        sync = getattr(self, "_grad_sync", None)
        if sync is not None and self.training and x4.requires_grad:
            # gradient-arena segments (cenet_amd.optim.cenet_segments) become final when backward reaches these
            # tensors: head+decoder at x4, stage4 at x3, stage3 at x2, stage2 at x1 (stage1: end of backward)
            for i, t in enumerate((x4, x3, x2, x1)):
                t.register_hook(sync.hook(i))
        dec = self.decoder(x4, [x3, x2, x1])
        return self.out(dec, x)
