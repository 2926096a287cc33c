"""bias_act epilogue bandwidth at the R50 shapes (dev tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd import functions as Fn  # noqa: E402

dev = "cuda:0"


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    for (C, H, W, res) in [(64, 368, 640, False), (64, 184, 320, False), (256, 184, 320, True), (128, 92, 160, False),
                           (512, 92, 160, True), (1024, 46, 80, True), (2048, 23, 40, True)]:
        x = torch.randn(30, C, H, W, device=dev)
        r = torch.randn_like(x) if res else None
        b = torch.randn(C, device=dev)
        ms = t(lambda: Fn.bias_act_(x, b, r, True))
        gb = x.numel() * 4 * (3 if res else 2) / 1e9
        print(f"C={C:4d} {H}x{W} res={res}: {ms:6.3f} ms  {gb / ms:6.2f} TB/s")
