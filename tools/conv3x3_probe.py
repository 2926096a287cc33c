"""The pixel decoder's 3x3 256->256 convolution on the stride-4 map (184x320): MIOpen runs it as implicit GEMM at
~120 TF/s direct-equivalent while the same layer on smaller maps gets its Winograd kernel (>200 TF/s equivalent).
Which map sizes get Winograd?"""
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.02


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (N, H, W, pad) in [(8, 46, 80, 1), (8, 92, 160, 1), (8, 184, 160, 1), (8, 92, 320, 1), (8, 184, 320, 1), (32, 46, 320, 1),
                       (32, 48, 322, 0), (120, 48, 322, 0), (120, 46, 320, 1), (60, 94, 322, 0), (60, 92, 320, 1),
                       (240, 25, 322, 0), (480, 48, 82, 0), (30, 184, 320, 1)]:
    x = torch.randn(N, 256, H, W, device=dev)
    ms = t(lambda: F.conv2d(x, w, None, 1, pad))
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    fl = 2 * N * Ho * Wo * 256 * 256 * 9
    print(f"N={N:3d} {H}x{W} pad={pad}: {ms:8.3f} ms, {fl / ms / 1e9:6.1f} TF/s direct-equivalent", flush=True)
    del x
