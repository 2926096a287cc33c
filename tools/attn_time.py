"""Time the attention kernel at the shapes of the pipeline (dev tool, GPU box):  python tools/attn_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd import functions as Fn  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


CASES = [  # name, Lq, Lk, B, heads, d, masked
    ("decoder cross-attn, level 2 (92x160)", 100, 14720, 30, 8, 32, True),
    ("decoder cross-attn, level 1 (46x80)", 100, 3680, 30, 8, 32, True),
    ("decoder cross-attn, level 0 (23x40)", 100, 920, 30, 8, 32, True),
    ("decoder self-attn", 100, 100, 30, 8, 32, False),
    ("refiner time-attn (T=30, 100 queries)", 30, 30, 100, 8, 64, False),
    ("tracker cross-attn (batch 1)", 100, 100, 1, 8, 64, False),
    ("ViT-L block, 10 frames", 3681, 3681, 10, 16, 64, False),
]
for name, Lq, Lk, B, H, d, masked in CASES:
    C = H * d
    q = torch.randn(Lq, B, C, device=dev)
    k = torch.randn(Lk, B, C, device=dev)
    v = torch.randn(Lk, B, C, device=dev)
    mask = allowed = None
    if masked:
        mask = (torch.rand(B, Lq, Lk, device=dev) < 0.7).to(torch.uint8)
        allowed = (mask == 0).sum(-1).to(torch.int32)
    us = timeit(lambda: Fn.attention(q, k, v, H, mask, allowed))
    flops = 4.0 * B * H * Lq * Lk * d
    print(f"{name:42s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  ({flops / us / 1e6 / 157.3:.2f} of the fp32-MFMA peak)")
