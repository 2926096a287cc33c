"""A/B of mask_gemm.hip build variants (dev tool, GPU box): python tools/exp/mask_variants/time.py"""
import ctypes
import glob
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda", 0)
B, Q, C, H, W = 30, 100, 256, 184, 320
torch.manual_seed(0)
emb = torch.randn(B, Q, C, device=dev)
mf = torch.randn(B, C, H, W, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = None
for path in sorted(glob.glob(os.path.join(HERE, "libmask_*.so")), key=lambda q: (0 if "base" in q else 1, q)):
    lib = ctypes.CDLL(path)
    lib.dvis_attn_mask.restype = lib.dvis_mask_logits.restype = ctypes.c_int
    line = os.path.basename(path).ljust(24)
    outs = []
    for (h, w) in ((23, 40), (46, 80), (92, 160)):
        mask = torch.empty(B, Q, h * w, dtype=torch.uint8, device=dev)
        allowed = torch.empty(B, Q, dtype=torch.int32, device=dev)
        run = lambda: lib.dvis_attn_mask(p(emb), p(mf), B, Q, C, H, W, h, w, p(mask), p(allowed), st)
        assert run() == 0
        line += f" mask {h}x{w} {t(run):7.1f} us"
        outs.append(mask.clone())
    out = torch.empty(B, Q, H * W, device=dev)
    run = lambda: lib.dvis_mask_logits(p(emb), p(mf), B, Q, C, ctypes.c_int64(H * W), p(out), st)
    assert run() == 0
    line += f"  logits Q=100 {t(run):7.1f} us"
    outs.append(out.clone())
    if ref is None:
        ref = outs
    same = [bool(torch.equal(a, b)) for a, b in zip(outs, ref)]
    diff = float((outs[3] - ref[3]).abs().max())
    nm = [int((a != b).sum()) for a, b in zip(outs[:3], ref[:3])]
    print(line, " same as first:", same, "differing mask bytes", nm, "max|logit diff|", diff)
