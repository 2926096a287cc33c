"""Latency of the tracker-sized ops inside a hipGraph (dev tool): library GEMMs, attention, add+LN, an empty-ish kernel."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd import functions as Fn  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def graph_time(fn, reps=50):
    """us per call when `reps` dependent-free calls are replayed from one graph (launch overhead excluded)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)


with torch.no_grad():
    for (M, N, K) in [(100, 512, 512), (100, 1536, 512), (100, 2048, 512), (100, 512, 2048), (100, 3072, 512)]:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev)
        b = torch.randn(N, device=dev)
        t_lib = graph_time(lambda: F.linear(x, w, b))
        t_relu = graph_time(lambda: Fn.linear(x, w, b, relu=True))
        print(f"linear {M}x{N}x{K}:  library {t_lib:6.2f} us   with fused ReLU epilogue {t_relu:6.2f} us")
    q = torch.randn(100, 1, 512, device=dev)
    k = torch.randn(100, 1, 512, device=dev)
    v = torch.randn(100, 1, 512, device=dev)
    print(f"attention Lq=Lk=100 B=1 h=8 d=64: {graph_time(lambda: Fn.attention(q, k, v, 8)):6.2f} us")
    q3 = torch.randn(30, 100, 512, device=dev)
    print(f"attention Lq=Lk=30 B=100 h=8 d=64: {graph_time(lambda: Fn.attention(q3, q3, q3, 8)):6.2f} us")
    q4 = torch.randn(100, 30, 256, device=dev)
    print(f"attention Lq=Lk=100 B=30 h=8 d=32: {graph_time(lambda: Fn.attention(q4, q4, q4, 8)):6.2f} us")
    ln = torch.nn.LayerNorm(512).to(dev)
    xx = torch.randn(100, 1, 512, device=dev)
    print(f"add+LN 100x512 fused: {graph_time(lambda: Fn.add_layer_norm(xx, xx, ln)):6.2f} us   torch: {graph_time(lambda: ln(xx + xx)):6.2f} us")
    print(f"empty-ish kernel (relu 100x512): {graph_time(lambda: torch.relu(xx)):6.2f} us")
