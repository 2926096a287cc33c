"""Timing ablations of x3_ffn_kernel (development builds with the DVIS_X3_ABL switch only): what each part costs.
bits: 1 no hidden-activation split, 2 no barriers, 4 no LDS-DMA, 8 no epilogue."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, ".")
from dvis_plus_amd import functions as Fn  # noqa: E402

DEV = "cuda:0"
M = 19320 * 30
x = torch.randn(M, 256, device=DEV)
l1, l2, norm = nn.Linear(256, 1024).to(DEV), nn.Linear(1024, 256).to(DEV), nn.LayerNorm(256).to(DEV)


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


with torch.no_grad():
    for abl in (0, 1, 2, 4, 8, 6, 7, 15):
        os.environ["DVIS_X3_ABL"] = str(abl)
        t = timeit(lambda: Fn.x3_ffn_ln(x, l1, l2, norm))
        print(f"abl {abl:2d}: {t:.3f} ms")
