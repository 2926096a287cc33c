"""Time a ViT-L block (block.py:36-104 shapes: 1024 features, 16 heads, 3681 tokens per 736 x 1280 frame) with and without row images,
and its four GEMMs in both forms.  python tools/row_image_time.py [frames]"""
import sys
import time

import torch

from dvis_plus_amd import functions as Fn
from dvis_plus_amd.vit_adapter import Block


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = "cuda:0"
    torch.manual_seed(0)
    with torch.no_grad():
        blk = Block(1024, 16, qkv_bias=True, init_values=1.0).to(dev).eval()
        x = torch.randn(frames, 3681, 1024, device=dev)
        M = frames * 3681
        for images in (True, False):
            Fn.X3_ROW_IMAGES = images
            print(f"block, {frames} frames, row images {'on ' if images else 'off'}: {timed(lambda: blk(x)):8.3f} ms")
        Fn.X3_ROW_IMAGES = True
        Fn.X3_GUARD.check_now(torch.device(dev))
        x2 = x.reshape(M, 1024)
        img = Fn.x3_rows_image(x2)
        a, m = blk.attn, blk.mlp
        flops = lambda n, k: 2.0 * M * n * k
        for name, w, b, act in (("proj  1024 -> 1024", a.proj.weight, a.proj.bias, None), ("fc1   1024 -> 4096 GELU", m.fc1.weight, m.fc1.bias, "gelu")):
            t0 = timed(lambda: Fn.x3_tile_linear(x2, w, b, act=act))
            t1 = timed(lambda: Fn.x3_tile_linear(img, w, b, act=act))
            f = flops(*w.shape)
            print(f"{name:26s} fp32 rows {t0:7.3f} ms ({f / t0 / 1e9:6.1f} TFLOP/s)   row image {t1:7.3f} ms ({f / t1 / 1e9:6.1f} TFLOP/s)")
        hid = Fn.x3_tile_linear(img, m.fc1.weight, m.fc1.bias, act="gelu")
        hid32 = Fn.x3_tile_linear(x2, m.fc1.weight, m.fc1.bias, act="gelu")
        t0 = timed(lambda: Fn.x3_tile_linear(hid32, m.fc2.weight, m.fc2.bias, residual=x2))
        t1 = timed(lambda: Fn.x3_tile_linear(hid, m.fc2.weight, m.fc2.bias, residual=x2))
        f = flops(1024, 4096)
        print(f"{'fc2   4096 -> 1024 + res':26s} fp32 rows {t0:7.3f} ms ({f / t0 / 1e9:6.1f} TFLOP/s)   row image {t1:7.3f} ms ({f / t1 / 1e9:6.1f} TFLOP/s)")
        t0 = timed(lambda: Fn.add_layer_norm(x, None, blk.norm1))
        t1 = timed(lambda: Fn.layer_norm_rows_image(x, blk.norm1))
        print(f"LayerNorm: fp32 out {t0:7.3f} ms   row image out {t1:7.3f} ms")
        t0 = timed(lambda: Fn.x3_qkv_attention(x, a.qkv.weight, a.qkv.bias, 16))
        xi = Fn.RowImage(img.data, x.shape, img.exp, 0)
        t1 = timed(lambda: Fn.x3_qkv_attention(xi, a.qkv.weight, a.qkv.bias, 16, out_image=True))
        print(f"qkv + attention: fp32 in / out {t0:7.3f} ms   images {t1:7.3f} ms")
        Fn.X3_GUARD.check_now(torch.device(dev))


if __name__ == "__main__":
    main()
