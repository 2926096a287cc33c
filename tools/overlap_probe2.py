"""Decoder-span / tracker overlap probe (dev tool)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_clip  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

dev = torch.device("cuda", 0)
m = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0).to(dev)
clip = synthetic_clip(30, dev)
to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)


def wall(fn, n=3):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def host(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) * 1e3


with torch.no_grad():
    images, _ = m.preprocess(clip)
    ms, mf = m.encode(images)
    for chunk in (30, 15, 10, 6):
        def dec():
            for s in range(0, 30, chunk):
                m.decode([x[s:s + chunk] for x in ms], mf[s:s + chunk])
        w = wall(dec)
        print(f"decoder 30 frames in spans of {chunk:2d}: {w:7.2f} ms (host enqueue {host(dec):.2f} ms)", flush=True)
    e, e_nn, lg = m.decode(ms, mf)
    for n in (30, 15, 10):
        def trk():
            for s in range(0, 30, n):
                m.tracker(to_bctq(e[s:s + n]), None, resume=s > 0, frame_embeds_no_norm=to_bctq(e_nn[s:s + n]),
                          need_masks=False)
        print(f"tracker 30 frames in spans of {n:2d}: {wall(trk):7.2f} ms (host {host(trk):.2f} ms)", flush=True)
    enc = lambda: m.encode(images)
    print(f"encode(30): {wall(enc):.2f} ms (host enqueue {host(enc):.2f} ms)")
    for r in (1, 2, 3):
        m.pipeline_rounds = r
        full = lambda: m([{"image": clip, "height": 720, "width": 1280}])
        print(f"forward rounds={r}: {wall(full):.2f} ms")
