"""Probe keys recorded by the whole-model goldens (TEST INFRASTRUCTURE; no reference import)."""
# parameters whose gradients / post-step values are recorded for the whole-model goldens
PROBE_KEYS = [
    "backbone.patch_embed1.proj.weight", "backbone.block1.0.attn.q.weight", "backbone.block1.0.attn.sr.weight",
    "backbone.block2.1.mlp.dwconv.dwconv.weight", "backbone.block3.5.mlp.fc2.weight", "backbone.block4.2.attn.kv.bias",
    "backbone.norm4.weight", "decoder.dec4.layer_scale_1", "decoder.dec4.mca.ccu.fc1.weight",
    "decoder.dec3.mca.value.dlps.1.depthwise.weight", "decoder.dec2.mca.denoising_module.w",
    "decoder.dec1.mlp.srm.dwc.weight", "decoder.up2.up_dwc.1.weight", "decoder.skip_enhancer3.diffattn.lambda_q1",
    "decoder.skip_enhancer2.boundary.w", "decoder.skip_enhancer1.diffattn.q_proj.weight",
    "decoder.skip_enhancer1.mixer.weight", "out.w", "out.rb.0.conv2.conv.weight", "out.up.up.1.weight",
    "out.out.0.conv1.conv.weight", "out.out.1.conv.conv.bias",
]
PROBE_BUFFERS = ["decoder.dec4.norm1.running_mean", "decoder.dec1.mca.ccu.bn.running_var",
                 "out.rb.0.norm1.running_var", "decoder.up1.up_dwc.2.running_mean",
                 "decoder.dec2.mlp.srm.bn.running_mean"]
