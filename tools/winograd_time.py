"""Own Winograd F(2x2, 3x3) fp32-MFMA convolution against the library's kernels at the shapes of the path (dev tool, GPU box):
    python tools/winograd_time.py [frames]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd import functions as Fn  # noqa: E402

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


CASES = [("FPN output conv 256->256 @184x320", 256, 256, 184, 320),
         ("res2 conv2 64->64 @184x320", 64, 64, 184, 320),
         ("res3 conv2 128->128 @92x160", 128, 128, 92, 160),
         ("res4 conv2 256->256 @46x80", 256, 256, 46, 80),
         ("res5 conv2 512->512 @23x40", 512, 512, 23, 40)]
print(f"{N} frames; direct FLOPs = 2*9*C*K*H*W*N, Winograd F(2x2,3x3) multiplies = /2.25; MFMA peak 157.3 TF (137.6 at 2.1 GHz)")
for name, C, K, H, W in CASES:
    x = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.02
    b = torch.randn(K, device=dev)
    t_lib = timeit(lambda: Fn.bias_act_(F.conv2d(x, w, None, 1, 1), b, None, True))
    t_own = timeit(lambda: Fn.conv3x3_bias_act(x, w, b, True, winograd=True))
    fl = 2.0 * 9 * C * K * H * W * N
    err = float((Fn.conv3x3_bias_act(x, w, b, True, winograd=True) - Fn.bias_act_(F.conv2d(x, w, None, 1, 1), b, None, True)).abs().max())
    print(f"{name:36s} library + bias_act {t_lib:8.1f} us   own {t_own:8.1f} us  ({t_lib / t_own:4.2f}x)   "
          f"own = {fl / t_own / 1e6:6.1f} direct-equivalent TFLOP/s, {fl / 2.25 / t_own / 1e6 / 157.3:.2f} of the MFMA peak on its own "
          f"multiplies; |own - library| max {err:.2e}")

print("stride 2 (csrc/conv3x3s2.hip, direct fp32 MFMA):")
for name, C, K, H, W in [("res3 first conv2 128->128 @184x320 -> 92x160", 128, 128, 184, 320),
                         ("res4 first conv2 256->256 @92x160 -> 46x80", 256, 256, 92, 160),
                         ("res5 first conv2 512->512 @46x80 -> 23x40", 512, 512, 46, 80)]:
    x = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.02
    b = torch.randn(K, device=dev)
    t_lib = timeit(lambda: Fn.bias_act_(F.conv2d(x, w, None, 2, 1), b, None, True))
    t_own = timeit(lambda: Fn.conv3x3s2_bias_act(x, w, b, True, own=True))
    fl = 2.0 * 9 * C * K * (H // 2) * (W // 2) * N
    print(f"{name:48s} library + bias_act {t_lib:8.1f} us   own {t_own:8.1f} us  ({t_lib / t_own:4.2f}x)   own = {fl / t_own / 1e6:6.1f} TFLOP/s = "
          f"{fl / t_own / 1e6 / 157.3:.2f} of the MFMA peak")

x = torch.randn(N, 3, 736, 1280, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) * 0.1
t_lib = timeit(lambda: F.conv2d(x, w, None, 2, 3))
t_own = timeit(lambda: Fn.conv7x7s2_stem(x, w, own=True))
fl = 2.0 * 147 * 64 * 368 * 640 * N
print(f"stem 7x7 / 2, 3 -> 64 @736x1280 (csrc/conv7x7s2.hip)      library {t_lib:8.1f} us   own {t_own:8.1f} us  ({t_lib / t_own:4.2f}x)   own = {fl / t_own / 1e6:6.1f} TFLOP/s = "
      f"{fl / t_own / 1e6 / 157.3:.2f} of the MFMA peak")

print("compute-bound 1x1 layers (csrc/conv1x1_mfma.hip) against the library's batched GEMM + bias_act:")
for name, C, K, H, W, with_res in [("res3 conv1 512->128 @92x160", 512, 128, 92, 160, False), ("res4 conv1 1024->256 @46x80", 1024, 256, 46, 80, False),
                                   ("res4 conv3 256->1024 @46x80 (+shortcut)", 256, 1024, 46, 80, True), ("res5 conv1 2048->512 @23x40", 2048, 512, 23, 40, False),
                                   ("res5 conv3 512->2048 @23x40 (+shortcut)", 512, 2048, 23, 40, True), ("res3 first conv1 256->128 @184x320", 256, 128, 184, 320, False)]:
    x = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(K, C, 1, 1, device=dev) * 0.02
    b = torch.randn(K, device=dev)
    r = torch.randn(N, K, H, W, device=dev) if with_res else None
    with torch.no_grad():
        t_lib = timeit(lambda: Fn.bias_act_(Fn.conv1x1(x, w), b, r, True))
        t_own = timeit(lambda: Fn.conv1x1_mfma(x, w, b, r, True))
    fl = 2.0 * C * K * H * W * N
    print(f"{name:44s} library + bias_act {t_lib:8.1f} us   own {t_own:8.1f} us  ({t_lib / t_own:4.2f}x)   own = {fl / t_own / 1e6:6.1f} TFLOP/s")

print("memory-bound 1x1 layers: csrc/conv1x1.hip (weights on chip) against the streaming MFMA kernel:")
for name, C, K, H, W, with_res in [("res2 conv1 256->64 @184x320", 256, 64, 184, 320, False), ("res3 conv1 (first) 256->128 @184x320", 256, 128, 184, 320, False),
                                   ("res3 conv3 128->512 @92x160 (+shortcut)", 128, 512, 92, 160, True)]:
    x = torch.randn(N, C, H, W, device=dev)
    w = torch.randn(K, C, 1, 1, device=dev) * 0.02
    b = torch.randn(K, device=dev)
    r = torch.randn(N, K, H, W, device=dev) if with_res else None
    with torch.no_grad():
        t_a = timeit(lambda: Fn.conv1x1_bias_act(x, w, b, r, True))
        t_b = timeit(lambda: Fn.conv1x1_mfma(x, w, b, r, True))
    print(f"{name:44s} conv1x1_bias_act {t_a:8.1f} us   conv1x1_mfma {t_b:8.1f} us")
