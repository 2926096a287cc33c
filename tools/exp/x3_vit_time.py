"""The ViT-L block's four GEMMs (30 frames x 3681 tokens) on the split-f16 linear kernel: time and fp32-equivalent TFLOP/s."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn
dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M = frames * 3681
g = torch.Generator().manual_seed(0)
tot = 0.0
for name, K, N, act in (("qkv", 1024, 3072, None), ("proj", 1024, 1024, None), ("fc1", 1024, 4096, "gelu"), ("fc2", 4096, 1024, None)):
    w = (torch.randn(N, K, generator=g) * 0.02).to(dev)
    b = torch.zeros(N, device=dev)
    x = torch.randn(M, K, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev) if name in ("proj", "fc2") else None
    with torch.no_grad():
        for _ in range(2):
            Fn.linear(x, w, b, tall=True, act=act, residual=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            Fn.linear(x, w, b, tall=True, act=act, residual=res)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    print(f"{name:5s} M={M} K={K} N={N}: {ms:7.3f} ms  {2.0 * M * K * N / ms / 1e9:6.1f} TFLOP/s fp32-equivalent  ({6.0 * M * K * N / ms / 1e9 / 2500:.3f} of the f16 matrix peak)")
    del x, w, res
print(f"block total {tot:.2f} ms")
