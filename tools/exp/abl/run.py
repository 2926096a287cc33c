"""Timing-only ablations of the key-partitioned attention kernel (results are WRONG by construction): which part of a
step costs what.  Variant libraries are built by hand with -DDVIS_ABL_*; see DESIGN.md section 3.4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from dvis_plus_amd import native
native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"lib_{sys.argv[1]}.so")
import torch
from dvis_plus_amd import functions as Fn
dev = "cuda:0"
B, H, d, Lq, Lk = 30, 8, 32, 100, 14720
q = torch.randn(Lq, B, H * d, device=dev); k = torch.randn(Lk, B, H * d, device=dev); v = torch.randn(Lk, B, H * d, device=dev)
mask = (torch.rand(B, Lq, Lk, device=dev) < 0.6).to(torch.uint8)
allowed = (mask == 0).sum(-1).int()
f = lambda: Fn.attention(q, k, v, H, mask, allowed)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(f"{sys.argv[1]:8s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
