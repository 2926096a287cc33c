"""fp16-value MSDA experiment (dvis_msda_fused_forward_h16): time and error vs the fp32 kernel at the real encoder
shape (30 frames of 720p, init-rule offsets).  Dev tool."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd import functions as Fn, native  # noqa: E402
from dvis_plus_amd.pixel_decoder import MSDeformAttnPixelDecoder, r50_input_shape  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
pd = MSDeformAttnPixelDecoder(r50_input_shape(), transformer_dropout=0.0, transformer_nheads=8,
                              transformer_dim_feedforward=1024, transformer_enc_layers=1, conv_dim=256, mask_dim=256,
                              norm="GN", transformer_in_features=["res3", "res4", "res5"], common_stride=4).to(dev).eval()


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


with torch.no_grad():
    at = pd.transformer.encoder.layers[0].self_attn
    at.sampling_offsets.weight.normal_(0, 0.01)
    shapes_py = [(23, 40), (46, 80), (92, 160)]
    ss, lsi = pd.transformer._shape_tensors(shapes_py, dev)
    ref = pd.transformer.encoder.reference_points_unpadded(shapes_py, dev).contiguous()
    N, S, C, M, L, P = 30, 19320, 256, 8, 3, 4
    src = torch.randn(N, S, C, device=dev)
    value = at.value_proj(src).view(N, S, M, C // M)
    w, b = at._fused_projection()
    proj = F.linear(src.view(N * S, C), w, b)
    n_off = M * L * P * 2
    o32 = Fn.msda_fused_forward(value, ss, lsi, ref, proj[:, :n_off], proj[:, n_off:], L, P)
    v16 = value.to(torch.float16).contiguous()
    out = torch.empty_like(o32)
    lib = native.lib()
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    offs, lgs = proj[:, :n_off], proj[:, n_off:]

    def h16():
        rc = lib.dvis_msda_fused_forward_h16(p(v16), p(ss), p(lsi), p(ref), 1, p(offs), offs.stride(0), p(lgs), lgs.stride(0),
                                             N, S, M, C // M, L, S, P, p(out), st)
        assert rc == 0, native.last_error() if hasattr(native, "last_error") else rc
    h16()
    err = (out - o32).abs().max().item()
    t32 = t(lambda: Fn.msda_fused_forward(value, ss, lsi, ref, offs, lgs, L, P))
    t16 = t(h16)
    tcv = t(lambda: value.to(torch.float16))
    print(f"fp32 kernel {t32 / N:6.1f} us/frame-layer | fp16-value kernel {t16 / N:6.1f} us/frame-layer "
          f"(+ {tcv / N:4.1f} us to make the fp16 copy) | max|diff| {err:.2e} at max|value| {value.abs().max().item():.2f}")
