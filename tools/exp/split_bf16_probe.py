"""Extra-line experiment: the deformable encoder's FFN GEMMs (579 600 x 256 -> 1024 -> 256 per
30-frame layer, the single largest library kernel of a clip) with fp32 operands SPLIT into bf16 pieces and multiplied on the
bf16 matrix cores with fp32 accumulation, through the LIBRARY's bf16 GEMM (torch.mm(..., out_dtype=float32)) — i.e. the
arithmetic and an upper bound of what an own split kernel could reach, before writing one.
  x = hi + lo (2 pieces, 16 mantissa bits):  a.w ~ hi.hi + hi.lo + lo.hi                       (3 products)
  x = hi + mid + lo (3 pieces, 24 bits):     + hi.lo2 + lo2.hi + mid.mid                         (6 products)
either as separate GEMMs or as ONE GEMM over K-concatenated operands ([Ah Ah Al] x [Wh Wl Wh]^T).
Reports time per variant (GEMM only, and with the split passes) and max relative error vs fp64 on a row sample.
    python tools/exp/split_bf16_probe.py [frames]"""
import sys

import torch

DEV = "cuda:0"


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters       # ms


def split(x, pieces):
    out, r = [], x
    for _ in range(pieces):
        p = r.bfloat16()
        out.append(p)
        r = r - p.float()
    return out


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    M, K, N = 19320 * frames, 256, 1024
    g = torch.Generator(device=DEV).manual_seed(0)
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) * 0.06
    S = 4096                                                  # error on a row sample (fp64 reference)
    ref = a[:S].double() @ w.double().t()
    scale = (a[:S].double().abs() @ w.double().abs().t())

    def err(c):
        return float(((c[:S].double() - ref).abs() / scale).max())
    mm = lambda x, y: torch.mm(x, y.t(), out_dtype=torch.float32)
    res = []
    t = timeit(lambda: a @ w.t())
    res.append(("fp32 library GEMM", t, t, err(a @ w.t())))
    for pieces, pairs in ((2, [(0, 0), (0, 1), (1, 0)]), (3, [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)])):
        ap, wp = split(a, pieces), split(w, pieces)
        t_split = timeit(lambda: split(a, pieces))

        def sep():
            c = mm(ap[pairs[-1][0]], wp[pairs[-1][1]])         # smallest terms first
            for i, j in reversed(pairs[:-1]):
                c += mm(ap[i], wp[j])
            return c
        t = timeit(sep)
        res.append((f"{len(pairs)} products, separate bf16 GEMMs (+ adds)", t, t + t_split, err(sep())))
        acat = torch.cat([ap[i] for i, _ in pairs], 1)
        wcat = torch.cat([wp[j] for _, j in pairs], 1)
        t_cat = timeit(lambda: torch.cat([ap[i] for i, _ in pairs], 1))
        t = timeit(lambda: mm(acat, wcat))
        res.append((f"{len(pairs)} products, ONE bf16 GEMM over K' = {acat.shape[1]}", t, t + t_split + t_cat, err(mm(acat, wcat))))
        del acat, wcat
    fl = 2.0 * M * N * K
    print(f"FFN linear1 shape M={M} K={K} N={N} ({fl / 1e9:.0f} GFLOP); max |c - fp64| / sum|a||w| on {S} rows")
    for name, t, tt, e in res:
        print(f"  {name:58s} GEMM {t:7.3f} ms ({fl / t / 1e9:7.1f} TF fp32-equivalent)  with split passes {tt:7.3f} ms   rel err {e:.2e}")


if __name__ == "__main__":
    main()
