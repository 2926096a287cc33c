"""The decoder's masked cross-attention (100 queries, d = 32, the three pixel levels at 30 frames): split-f16 key-split kernel against the
fp32 one and against fp64 (development).  python tools/exp/cross_attn_x3.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


with torch.no_grad():
    for Lk, B in ((920, 30), (3680, 30), (14720, 30), (920, 2), (14720, 1)):
        Lq, H, d = 100, 8, 32
        q = torch.randn(Lq, B, H * d, device=dev)
        k = torch.randn(Lk, B, H * d, device=dev) * 1.5
        v = torch.randn(Lk, B, H * d, device=dev)
        mask = torch.rand(B, Lq, Lk, device=dev) < 0.6
        mask[:, 3] = True                                    # a fully blocked row
        allowed = (~mask).sum(-1).to(torch.int32)
        res = {}
        for name, sw in (("x3", True), ("f32", False)):
            Fn.X3_CROSS_ATTN = sw
            res[name] = Fn.attention(q, k, v, H, mask, allowed)
            res[name + "_us"] = timed(lambda: Fn.attention(q, k, v, H, mask, allowed))
        Fn.X3_GUARD.check_now(torch.device(dev))
        # fp64 reference
        qd, kd, vd = (t.double().reshape(t.shape[0], B, H, d).permute(1, 2, 0, 3) for t in (q, k, v))
        s = qd @ kd.transpose(-1, -2) / d ** 0.5
        m = mask.clone()
        m[allowed == 0] = False
        s = s.masked_fill(m[:, None], float("-inf"))
        ref = (torch.softmax(s, -1) @ vd).permute(2, 0, 1, 3).reshape(Lq, B, H * d)
        sc = float(ref.abs().max())
        e3, e32 = float((res["x3"].double() - ref).abs().max()) / sc, float((res["f32"].double() - ref).abs().max()) / sc
        same = torch.equal(res["x3"], Fn.attention(q, k, v, H, mask, allowed)) if True else None
        print(f"Lk {Lk:6d} B {B:2d}: x3 {res['x3_us']:7.1f} us  f32 {res['f32_us']:7.1f} us   err vs fp64: x3 {e3:.2e}  f32 {e32:.2e}   max|x3 - f32| {float((res['x3'] - res['f32']).abs().max()) / sc:.2e}")
