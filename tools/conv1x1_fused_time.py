"""The R50 bottleneck's 1x1 layers at 30 frames of 736x1280: library contraction + dvis_bias_act pass vs the fused kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd import functions as Fn, native   # noqa: E402

dev = torch.device("cuda", 0)
N = 30


def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


LAYERS = [("res2 conv1 256->64", 256, 64, 184, 320, False), ("res2.0 conv1 64->64", 64, 64, 184, 320, False),
          ("res2 conv3 64->256 +res", 64, 256, 184, 320, True), ("res2.0 shortcut 64->256", 64, 256, 184, 320, False),
          ("res3.0 conv1 256->128", 256, 128, 184, 320, False), ("res3 conv1 512->128", 512, 128, 92, 160, False),
          ("res3 conv3 128->512 +res", 128, 512, 92, 160, True), ("res4.0 conv1 512->256", 512, 256, 92, 160, False),
          ("res4 conv1 1024->256", 1024, 256, 46, 80, False), ("res4 conv3 256->1024 +res", 256, 1024, 46, 80, True)]
with torch.no_grad():
    for name, K, M, H, W, with_res in LAYERS:
        x = torch.randn(N, K, H, W, device=dev)
        w = torch.randn(M, K, 1, 1, device=dev) / K ** 0.5
        b = torch.randn(M, device=dev)
        res = torch.randn(N, M, H, W, device=dev) if with_res else None
        old = t(lambda: Fn.bias_act_(Fn.conv1x1(x, w), b, res, True))
        sup = native.lib().dvis_conv1x1_supported(K, M, H * W)
        new = t(lambda: Fn.conv1x1_bias_act(x, w, b, res, True)) if sup else float("nan")
        gb = 4 * N * H * W * (K + M * (2 if with_res else 1)) / 1e9
        print(f"{name:28s} library + epilogue {old:6.2f} ms   fused {new:6.2f} ms   ({gb:5.2f} GB compulsory"
              f"{'' if not sup else f' -> {gb / new:5.2f} TB/s'})", flush=True)
