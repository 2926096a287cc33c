"""Encoder seam A/B (cheap form): the residual handed to the LIBRARY GEMM as its C operand (beta = 1,
torch.addmm) so that the following LayerNorm pass reads one tensor instead of two — vs linear + add_layernorm(y, res)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn
dev = "cuda:0"
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    M = 579600
    norm = torch.nn.LayerNorm(256).to(dev)
    for K in (256, 1024):
        x = torch.randn(M, K, device=dev); w = torch.randn(256, K, device=dev) / K ** 0.5; b = torch.randn(256, device=dev)
        res = torch.randn(M, 256, device=dev)
        a = t(lambda: Fn.add_layer_norm(torch.nn.functional.linear(x, w, b), res, norm))
        g1 = t(lambda: torch.nn.functional.linear(x, w, b))
        c = t(lambda: Fn.add_layer_norm(torch.addmm(res, x, w.t()), None, norm))
        g2 = t(lambda: torch.addmm(res, x, w.t()))
        own = t(lambda: Fn.add_layer_norm(Fn.gemm_nt(x, w, b, res=res), None, norm))
        g3 = t(lambda: Fn.gemm_nt(x, w, b, res=res))
        print(f"K={K}: linear {g1:.0f} + add_ln(y, res) = {a:.0f} us | addmm(res) {g2:.0f} + ln(y) = {c:.0f} us | own gemm(+bias+res) {g3:.0f} + ln(y) = {own:.0f} us", flush=True)
