"""mask_gemm.hip on the headline shapes: 30 frames, 100 queries, 256 channels, stride-4 map 184x320; attention masks at
the three decoder levels (23x40, 46x80, 92x160) and the full-resolution logits of 20 selected queries."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd import functions as Fn   # noqa: E402

dev = torch.device("cuda", 0)
B, Q, C, H, W = 30, 100, 256, 184, 320
emb = torch.randn(B, Q, C, device=dev)
mf = torch.randn(B, C, H, W, device=dev)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = 0.
for (h, w) in ((23, 40), (46, 80), (92, 160)):
    us = t(lambda: Fn.attn_mask(emb, mf, (h, w)))
    fl = 2 * B * Q * C * h * w * 4
    tot += us
    print(f"attn_mask -> {h}x{w}: {us:8.1f} us  {fl / us / 1e6:6.1f} TF/s (useful flops, 100 of 112 padded rows)")
print(f"three levels: {tot:.1f} us (x3 per clip = {3 * tot / 1e3:.2f} ms)")
for q in (20, 100):
    us = t(lambda: Fn.mask_logits(emb[:, :q].contiguous(), mf))
    print(f"mask_logits Q={q}: {us:8.1f} us  {2 * B * q * C * H * W / us / 1e6:6.1f} TF/s")
