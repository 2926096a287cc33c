"""MSDeformAttn backward (dvis_msda_backward), timed at the R50 720p encoder shape and the ViT-L extractor shape.  (Round 4
also timed a lane-owned 16-byte form through DVIS_MSDA_BWD_VEC: 4x slower, removed — profiles/r04_msda_bwd_time.txt.)  Algorithmic bytes = the forward's
(value + loc + w + out read as grad_out) + the three gradient tensors written (SURVEY.md section 8(d) + (f)-1).
    python tools/msda_bwd_time.py"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = "cuda:0"


def case(name, N, shapes, M, D, Lq, P=4):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(0)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    ss = torch.tensor(shapes, dtype=torch.int64, device=DEV)
    lsi = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    value = torch.randn(N, S, M, D, generator=g).to(DEV)
    # the callers' geometry: queries = the pixels of the (23, 40) / (46, 80) / (92, 160) pyramid in raster order per level, each
    # looking at its own normalised centre + the init-rule ring (head m along angle 2 pi m / M, point p at p + 1 pixels) + a
    # learned part of ~0.2 pixel; weights a softmax
    import math
    qshapes = [(92, 160), (46, 80), (23, 40)] if Lq == 19320 else shapes
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w_) + 0.5) / w_, indexing="ij"), -1)
                     .flip(-1).reshape(-1, 2) for h, w_ in qshapes])[:Lq]                       # (Lq, 2) as (x, y)
    ang = torch.arange(M) * (2 * math.pi / M)
    d = torch.stack([ang.cos(), ang.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    ring = d[:, None, None, :] * torch.arange(1, P + 1)[None, None, :, None]                      # (M, 1, P, 2) pixels
    off_px = ring.expand(M, L, P, 2)[None, None] + 0.2 * torch.randn(N, Lq, M, L, P, 2, generator=g)
    wh = torch.tensor([[w_, h] for h, w_ in shapes], dtype=torch.float32)                        # (L, 2) as (W, H)
    loc = (ref[None, :, None, None, None, :] + off_px / wh[None, None, None, :, None, :]).to(DEV).contiguous()
    w = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P).to(DEV).contiguous()
    go = torch.randn(N, Lq, M * D, generator=g).to(DEV)
    run = lambda: Fn.ms_deform_attn_backward(value, ss, lsi, loc, w, go, deterministic=False)
    det = lambda: Fn.ms_deform_attn_backward(value, ss, lsi, loc, w, go, deterministic=True)
    fwd = lambda: Fn.ms_deform_attn_forward(value, ss, lsi, loc, w)
    out = {}
    for tag, fn in (("backward", run), ("deterministic", det), ("forward", fwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[tag] = e0.elapsed_time(e1) / 10 * 1e3
    fwd_bytes = 4 * (N * S * M * D + N * Lq * M * L * P * 3 + N * Lq * M * D)
    bwd_bytes = fwd_bytes + 4 * (N * S * M * D + N * Lq * M * L * P * 3)
    us = out["backward"] - 0.0
    print(f"{name}: backward {out['backward']:.1f} us per launch ({N} frames; includes the zero-fill of grad_value) = "
          f"{out['backward'] / N:.1f} us/frame, algorithmic {bwd_bytes / N / 1e6:.1f} MB/frame -> "
          f"{bwd_bytes / (us * 1e-6) / 1e9:.0f} GB/s = {bwd_bytes / (us * 1e-6) / 8e12:.3f} of 8 TB/s;  forward (unfused op) "
          f"{out['forward'] / N:.1f} us/frame;  deterministic backward (64-bit fixed-point atomics + max / convert passes) "
          f"{out['deterministic'] / N:.1f} us/frame", flush=True)


def main():
    if os.environ.get("DVIS_MSDA_BWD_CHILD") != "1":
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, DVIS_MSDA_BWD_CHILD="1"),
                       stdin=subprocess.DEVNULL)
        return
    case("R50 720p encoder layer (S = Lq = 19320, M = 8, D = 32, L = 3, P = 4)", 4, [(92, 160), (46, 80), (23, 40)], 8, 32, 19320)
    case("ViT-L extractor (value 46 x 80, M = 16, D = 64, L = 1; Lq = 19320)", 4, [(46, 80)], 16, 64, 19320)


if __name__ == "__main__":
    main()
