"""Library GEMM time for the fused offsets|logits projection (N*S rows x 256 -> 288) at small and large batches, and
for padded output widths (dev tool: hipBLASLt's heuristic picks a slow macro-tile for some (rows, 288) shapes)."""
import torch
import torch.nn.functional as F

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
    return e0.elapsed_time(e1) / reps * 1e3


with torch.no_grad():
    for frames in (1, 2, 4, 8, 15, 30):
        rows = frames * 19320
        x = torch.randn(rows, 256, device=dev)
        line = f"frames={frames:2d} rows={rows:6d}:"
        for n in (288, 320, 384, 512):
            w = torch.randn(n, 256, device=dev)
            b = torch.randn(n, device=dev)
            us = t(lambda: F.linear(x, w, b))
            line += f"  N={n}: {us:8.1f} us ({2.0 * rows * 256 * n / us / 1e6:6.1f} TF/s)"
        w1, b1, w2, b2 = (torch.randn(192, 256, device=dev), torch.randn(192, device=dev),
                          torch.randn(96, 256, device=dev), torch.randn(96, device=dev))
        us = t(lambda: (F.linear(x, w1, b1), F.linear(x, w2, b2)))
        line += f"  192+96 split: {us:8.1f} us"
        print(line, flush=True)
