"""Operand-image forms of csrc/conv1x1_x3.hip against the fp32-map forms at the benchmark's shapes (30 frames):
the FPN's top-down sum + 3x3 output convolution, and the res3 - res5 bottleneck interiors.
    python tools/conv_image_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd import functions as Fn      # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    torch.manual_seed(0)
    with torch.no_grad():
        # FPN: lateral (T, 256, 184, 320) + up(top (T, 256, 92, 160)) -> 3x3 256 -> 256
        lat = torch.randn(T, 256, 184, 320, device=DEV)
        top = torch.randn(T, 256, 92, 160, device=DEV)
        aff = (torch.rand(T * 256, device=DEV) + 0.5, torch.randn(T * 256, device=DEV))
        w = torch.randn(256, 256, 3, 3, device=DEV) * 0.02
        b = torch.randn(256, device=DEV)
        a = timed(lambda: Fn.upsample_add(lat, top, aff))
        c = timed(lambda: Fn.conv3x3_x3(lat, w, b, None, False))
        ai = timed(lambda: Fn.upsample_add_image(lat, top, aff))
        img = Fn.upsample_add_image(lat, top, aff)
        ci = timed(lambda: Fn.conv_x3_image(img, w, b))
        d = float((Fn.conv_x3_image(img, w, b) - Fn.conv3x3_x3(Fn.upsample_add(lat, top, aff), w, b, None, False)).abs().max())
        print(f"FPN: top-down sum {a:.3f} ms -> as image {ai:.3f} ms; 3x3 256 -> 256 {c:.3f} ms -> from the image {ci:.3f} ms   (max |diff| {d:.2e})")
        del lat, top, img
        # bottleneck interiors: (C_in, mid, H, W)
        for name, C, M, H, W in (("res3", 512, 128, 92, 160), ("res4", 1024, 256, 46, 80), ("res5", 2048, 512, 23, 40)):
            x = torch.randn(T, C, H, W, device=DEV).relu()
            w1 = torch.randn(M, C, 1, 1, device=DEV) * (2.0 / C) ** 0.5
            w2 = torch.randn(M, M, 3, 3, device=DEV) * (2.0 / (9 * M)) ** 0.5
            w3 = torch.randn(C, M, 1, 1, device=DEV) * (2.0 / M) ** 0.5
            t1 = timed(lambda: Fn.conv1x1_x3(x, w1, None, None, True))
            a1 = Fn.conv1x1_x3(x, w1, None, None, True)
            t2 = timed(lambda: Fn.conv3x3_x3(a1, w2, None, None, True))
            a2 = Fn.conv3x3_x3(a1, w2, None, None, True)
            t3 = timed(lambda: Fn.conv1x1_x3(a2, w3, None, x, True))
            i1 = timed(lambda: Fn.conv_x3_image(x, w1, None, None, relu=True, out_image=True))
            m1 = Fn.conv_x3_image(x, w1, None, None, relu=True, out_image=True)
            i2 = timed(lambda: Fn.conv_x3_image(m1, w2, None, None, relu=True, out_image=True))
            m2 = Fn.conv_x3_image(m1, w2, None, None, relu=True, out_image=True)
            i3 = timed(lambda: Fn.conv_x3_image(m2, w3, None, x, relu=True))
            print(f"{name} identity block ({C} / {M} at {H} x {W}): maps {t1:.3f} + {t2:.3f} + {t3:.3f} = {t1 + t2 + t3:.3f} ms; "
                  f"images {i1:.3f} + {i2:.3f} + {i3:.3f} = {i1 + i2 + i3:.3f} ms")


if __name__ == "__main__":
    main()
