"""Time dvis_gemm_nt's tile configurations on the tracker / refiner GEMM shapes, next to the library GEMM.
    python tools/gemm_time.py [--big]      (--big: the segmenter's large-M shapes too)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd import functions as Fn, native  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters       # us


def main():
    shapes = [(100, 512, 512), (100, 1536, 512), (100, 2048, 512), (100, 512, 2048), (200, 512, 512), (200, 2048, 512),
              (3000, 512, 512), (3000, 1536, 512), (3000, 2048, 512), (3000, 512, 2048), (3000, 512, 2560),
              (3000, 512, 1536), (3000, 6144, 512), (3000, 125, 1024), (3000, 256, 512)]
    if "--big" in sys.argv:
        shapes += [(579600, 256, 256), (579600, 320, 256), (579600, 1024, 256), (579600, 256, 1024),
                   (441600, 768, 256), (3000, 2048, 256), (3000, 256, 2048), (3000, 768, 256)]
    ncfg = native.lib().dvis_gemm_num_configs()
    for M, N, K in shapes:
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(N, K, device=DEV)
        b = torch.randn(N, device=DEV)
        iters = 200 if M <= 3000 else 10
        lib = timeit(lambda: torch.nn.functional.linear(a, w, b), iters)
        # captured in a graph: what a node costs inside the tracker's hipGraph (no launch overhead)
        res = []
        for c in range(ncfg):
            rt_ct = None
            try:
                t = timeit(lambda: Fn.gemm_nt(a, w, b, config=c), iters)
            except RuntimeError as e:
                t = float("nan")
            res.append(t)
        auto = timeit(lambda: Fn.gemm_nt(a, w, b), iters)
        best = min(range(ncfg), key=lambda c: res[c] if res[c] == res[c] else 1e30)
        fl = 2.0 * M * N * K
        print(f"M={M:6d} N={N:5d} K={K:5d}  lib {lib:8.1f} us ({fl / lib / 1e6:6.1f} TF)  auto {auto:8.1f} us "
              f"({fl / auto / 1e6:6.1f} TF)  best cfg {best} {res[best]:8.1f} us  all: "
              + " ".join(f"{t:.1f}" for t in res), flush=True)


if __name__ == "__main__":
    main()
