"""Marginal cost of one link of a launch-bound chain: a hipGraph of `n` back-to-back calls on one stream, each with its OWN
weights (n x 4 MB > the L2s: weights come from MALL / HBM as in the tracker), for dvis_gemm_ln's configurations and modes
next to dvis_gemm_nt (+ dvis_add_layernorm where the fused form replaces it).
    python tools/gemm_ln_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd import functions as Fn, native  # noqa: E402

DEV = "cuda:0"
NL = 24


def graph_time(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20 / NL


def main():
    torch.manual_seed(0)
    lib = native.lib()
    with torch.no_grad():
        for M, N, K in [(100, 512, 512), (100, 1536, 512), (100, 2048, 512), (100, 3072, 512), (100, 512, 2048)]:
            a = torch.randn(M, K, device=DEV)
            add = torch.randn(M, K, device=DEV)
            res = torch.randn(M, N, device=DEV)
            ws = [torch.randn(N, K, device=DEV) for _ in range(NL)]
            bs = [torch.randn(N, device=DEV) for _ in range(NL)]
            n1, n2 = torch.nn.LayerNorm(K).to(DEV), torch.nn.LayerNorm(K).to(DEV)
            base = graph_time(lambda: [Fn.gemm_nt(a, ws[i], bs[i], res=res) for i in range(NL)])
            ln = graph_time(lambda: [Fn.add_layer_norm(a, add, n1) for i in range(NL)]) if K <= 1024 else float("nan")
            line = f"M={M} N={N} K={K}: gemm_nt {base:5.1f} us  (add_layernorm {ln:4.1f})  gemm_ln cfg:"
            print(line, flush=True)
            if not lib.dvis_gemm_ln_supported(M, N, K, 0):
                continue
            for c in range(lib.dvis_gemm_ln_num_configs()):
                out = []
                for mode in ("plain", "ln1", "ln1+add+ln2") if K <= 512 else ("plain",):
                    kw = dict(norm1=n1 if "ln1" in mode else None, add=add if "add" in mode else None,
                              norm2=n2 if "ln2" in mode else None)
                    try:
                        t = graph_time(lambda: [Fn.gemm_ln(a, ws[i], bs[i], res=res, config=c, **kw) for i in range(NL)])
                    except RuntimeError:
                        t = float("nan")
                    out.append(t)
                if out[0] == out[0]:
                    print(f"     cfg {c}: " + "  ".join(f"{t:5.1f}" for t in out) + "   (plain / ln1 / ln1+add+ln2)", flush=True)


if __name__ == "__main__":
    main()
