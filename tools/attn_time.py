"""Time the attention kernel at the shapes of the pipeline (dev tool, GPU box):  python tools/attn_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd import functions as Fn  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timeit(fn, iters=20):
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


CASES = [  # name, Lq, Lk, B, heads, d, masked
    ("decoder cross-attn, level 2 (92x160)", 100, 14720, 30, 8, 32, True),
    ("decoder cross-attn, level 1 (46x80)", 100, 3680, 30, 8, 32, True),
    ("decoder cross-attn, level 0 (23x40)", 100, 920, 30, 8, 32, True),
    ("decoder self-attn", 100, 100, 30, 8, 32, False),
    ("refiner time-attn (T=30, 100 queries)", 30, 30, 100, 8, 64, False),
    ("tracker cross-attn (batch 1)", 100, 100, 1, 8, 64, False),
    ("ViT-L block, 10 frames", 3681, 3681, 10, 16, 64, False),
]
for name, Lq, Lk, B, H, d, masked in CASES:
    C = H * d
    q = torch.randn(Lq, B, C, device=dev)
    k = torch.randn(Lk, B, C, device=dev)
    v = torch.randn(Lk, B, C, device=dev)
    mask = allowed = None
    if masked:
        mask = (torch.rand(B, Lq, Lk, device=dev) < 0.7).to(torch.uint8)
        allowed = (mask == 0).sum(-1).to(torch.int32)
    us = timeit(lambda: Fn.attention(q, k, v, H, mask, allowed))
    flops = 4.0 * B * H * Lq * Lk * d
    print(f"{name:42s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  ({flops / us / 1e6 / 157.3:.2f} of the fp32-MFMA peak)")


# ---------------------------------------------------------------------------------------------------------------
# Would skipping fully blocked 16-query x 16-key tiles pay?    The masked cross-attention blocks a
# key where the previous layer's mask logit is < 0 (sigmoid < 0.5), and a query whose mask is EMPTY attends everywhere
# (dvis_Plus/video_mask2former_transformer_decoder.py:297).  A key tile can only be skipped for a query tile if ALL 16
# queries of the tile block ALL 16 keys and none of the 16 is an empty-mask (attend-everywhere) row.  Spatially coherent
# masks (blobs covering 5-20 % of the pixels, like trained instance / stuff masks), with a fraction of empty-mask queries
# (the "no object" queries of a trained Mask2Former: most of the 100), unsorted and sorted by emptiness:
def coherent_masks(B, Q, h, w, empty_frac, g):
    yy, xx = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    cy = torch.rand(B, Q, 1, 1, device=dev, generator=g) * h
    cx = torch.rand(B, Q, 1, 1, device=dev, generator=g) * w
    area = (0.05 + 0.15 * torch.rand(B, Q, 1, 1, device=dev, generator=g)) * h * w          # 5 - 20 % of the pixels
    ry = torch.sqrt(area / 3.14159 * (0.5 + torch.rand(B, Q, 1, 1, device=dev, generator=g)))
    rx = area / 3.14159 / ry
    inside = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0                             # allowed region (an ellipse)
    empty = torch.rand(B, Q, device=dev, generator=g) < empty_frac
    inside = inside & ~empty[:, :, None, None]
    return (~inside).flatten(2).to(torch.uint8), empty                                      # 1 = blocked


def skippable_fraction(mask, empty, sort):
    B, Q, Lk = mask.shape
    if sort:   # queries of a frame ordered so that the attend-everywhere rows share tiles
        order = empty.to(torch.int32).argsort(dim=1, stable=True)
        mask = mask.gather(1, order[:, :, None].expand(-1, -1, Lk))
        empty = empty.gather(1, order)
    QT, KT = (Q + 15) // 16, (Lk + 15) // 16
    pad_q, pad_k = QT * 16 - Q, KT * 16 - Lk
    m = torch.nn.functional.pad(mask, (0, pad_k, 0, pad_q), value=1).bool()
    e = torch.nn.functional.pad(empty, (0, pad_q), value=False)
    tile_blocked = m.view(B, QT, 16, KT, 16).all(4).all(2)                                   # (B, QT, KT)
    tile_has_global_row = e.view(B, QT, 16).any(2)
    return float((tile_blocked & ~tile_has_global_row[:, :, None]).float().mean())


print("\nfraction of 16 x 16 (query, key) tiles that a tile-skipping kernel could drop, level 2 (92 x 160), 100 queries:")
g = torch.Generator(device=dev).manual_seed(0)
for empty_frac in (0.0, 0.5, 0.8):
    mask, empty = coherent_masks(30, 100, 92, 160, empty_frac, g)
    blocked = float(mask.float().mean())
    print(f"  blobs of 5-20 % of the pixels, {int(empty_frac * 100):2d} % empty-mask queries: {blocked:.2f} of the mask bits blocked; "
          f"skippable tiles: queries as they come {skippable_fraction(mask, empty, False):.3f}, sorted by emptiness "
          f"{skippable_fraction(mask, empty, True):.3f}")
    q = torch.randn(100, 30, 256, device=dev)
    k = torch.randn(14720, 30, 256, device=dev)
    v = torch.randn(14720, 30, 256, device=dev)
    allowed = (mask == 0).sum(-1).to(torch.int32)
    us = timeit(lambda: Fn.attention(q, k, v, 8, mask, allowed))
    print(f"      this kernel on those masks: {us:.1f} us (it does not skip: dead keys cost the same as live ones)")
mask = (torch.rand(30, 100, 14720, device=dev) < 0.7).to(torch.uint8)
print(f"  i.i.d. 70 % random masks (what the benchmark's random-init heads produce): "
      f"{skippable_fraction(mask, torch.zeros(30, 100, dtype=torch.bool, device=dev), False):.3f}")
