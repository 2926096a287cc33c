"""Probe for DESIGN.md section 10 item 7 (not product code): the decoder's attention masks from a feature PYRAMID.
    python tools/exp/attn_mask_pyramid_probe.py
dvis_attn_mask contracts the FOUR centre pixels of every s x s block of the stride-4 mask features and averages the four
logits (the reference's order: contract -> bilinear down-size -> threshold): 4 x the products of the level's own pixel
count, once per decoder layer (9 launches per clip, each reading the 1.8 GB map).  The down-sizing is linear, so the four
FEATURE pixels can be averaged once per clip into three small maps and every layer contracts its own level's map:
    pooled_s = 0.25 * ((f_a + f_b) + (f_c + f_d)),   mask = (einsum(embed, pooled_s) < 0)
This probe measures, with the existing kernels only (torch slicing for the pooling, dvis_mask_logits for the contraction):
  * how many mask bits differ from dvis_attn_mask's (the rounding ORDER changes: bits can flip where |logit| ~ 1e-6 scale),
  * what the contraction costs on the pooled maps against dvis_attn_mask on the full map.
A product version would pool in one kernel (one read of the map, 0.6 GB written) and threshold in the contraction's epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn   # noqa: E402

dev = torch.device("cuda", 0)
B, Q, C, H, W = 30, 100, 256, 184, 320
g = torch.Generator(device=dev).manual_seed(0)
emb = torch.randn(B, Q, C, device=dev, generator=g)
mf = torch.randn(B, C, H, W, device=dev, generator=g)


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pool(f, s):
    """the four centre pixels of every s x s block, added in the reference's order ((a + b) + (c + d)) * 0.25"""
    o = s // 2 - 1
    a, b = f[:, :, o::s, o::s], f[:, :, o::s, o + 1::s]
    c, d = f[:, :, o + 1::s, o::s], f[:, :, o + 1::s, o + 1::s]
    return (((a + b) + (c + d)) * 0.25).contiguous()


with torch.no_grad():
    tot_old = tot_new = 0.0
    t_pool = 0.0
    for s, (h, w) in ((8, (23, 40)), (4, (46, 80)), (2, (92, 160))):
        mask, allowed = Fn.attn_mask(emb, mf, (h, w))
        p = pool(mf, s)
        logits = Fn.mask_logits(emb, p).view(B, Q, h * w)
        new = (logits < 0).to(torch.uint8)                       # 1 = blocked, as dvis_attn_mask writes it
        diff = int((new != mask).sum())
        ref = torch.einsum("bqc,bchw->bqhw", emb.double(), p.double()).view(B, Q, h * w)
        near = float(ref.abs()[new != mask].max()) if diff else 0.0
        us_old = t(lambda: Fn.attn_mask(emb, mf, (h, w)))
        us_new = t(lambda: Fn.mask_logits(emb, p))
        us_pool = t(lambda: pool(mf, s), 3)
        tot_old += us_old
        tot_new += us_new
        t_pool += us_pool
        print(f"level {h}x{w} (s = {s}): {diff} of {mask.numel()} mask bits differ (largest |fp64 logit| among them {near:.2e}); "
              f"dvis_attn_mask {us_old:.1f} us, contraction on the pooled map {us_new:.1f} us (+ fp32 logits written: the product "
              f"form thresholds in the epilogue), torch pooling {us_pool:.1f} us")
    print(f"per clip (3 layers per level): dvis_attn_mask {3 * tot_old / 1e3:.2f} ms; pyramid {3 * tot_new / 1e3:.2f} ms + pooling "
          f"once {t_pool / 1e3:.2f} ms (torch slicing; a fused kernel reads the 1.8 GB map once: ~0.4 ms)")
