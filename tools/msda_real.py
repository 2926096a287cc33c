"""The fused MSDA launch exactly as the pixel decoder issues it (30 frames, init-rule offsets), timed (dev tool)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd.pixel_decoder import MSDeformAttnPixelDecoder, r50_input_shape  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
pd = MSDeformAttnPixelDecoder(r50_input_shape(), transformer_dropout=0.0, transformer_nheads=8,
                              transformer_dim_feedforward=1024, transformer_enc_layers=1, conv_dim=256, mask_dim=256,
                              norm="GN", transformer_in_features=["res3", "res4", "res5"], common_stride=4).to(dev).eval()
with torch.no_grad():
    at = pd.transformer.encoder.layers[0].self_attn
    at.sampling_offsets.weight.normal_(0, 0.01)           # trained-like: offsets vary a little around the init ring
    shapes_py = [(23, 40), (46, 80), (92, 160)]
    ss, lsi = pd.transformer._shape_tensors(shapes_py, dev)
    ref = pd.transformer.encoder.reference_points_unpadded(shapes_py, dev)
    N, S, C = 30, 19320, 256
    src = torch.randn(N, S, C, device=dev)
    run = lambda: at(src, ref, src, ss, lsi, None, spatial_shapes_py=shapes_py)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    from dvis_plus_amd import functions as Fn
    orig = Fn.msda_fused_forward
    ev = []

    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        o = orig(*a, **k)
        e1.record()
        ev.append((e0, e1))
        return o
    Fn.msda_fused_forward = timed
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    print(f"fused MSDA in-module: {ms * 1e3:.1f} us/launch ({N} frames) = {ms * 1e3 / N:.1f} us/frame-layer, "
          f"{61824000 * N / ms / 1e6:.0f} GB/s algorithmic")
