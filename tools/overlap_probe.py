"""Does the tracker's hipGraph overlap with the segmenter on a second stream? (dev tool)
Times: segmenter(T) alone at several batch sizes, tracker(T) alone, and both concurrently on two streams."""
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


with torch.no_grad():
    for chunk in (30, 15, 10, 6, 3):
        def seg():
            for s in range(0, 30, chunk):
                images, _ = m.preprocess(clip[s:s + chunk])
                m.segment(images)
        print(f"segmenter 30 frames in batches of {chunk:2d}: {wall(seg):7.2f} ms", flush=True)
    images, _ = m.preprocess(clip[:15])
    e, e_nn, lg, mf = m.segment(images)
    trk = lambda: m.tracker(to_bctq(e), None, resume=False, frame_embeds_no_norm=to_bctq(e_nn), need_masks=False)
    seg15 = lambda: m.segment(m.preprocess(clip[15:])[0])
    t_trk, t_seg = wall(trk), wall(seg15)
    side = torch.cuda.Stream()

    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        seg15()
        with torch.cuda.stream(side):
            trk()
        main.wait_stream(side)

    def both_rev():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            trk()
        seg15()
        main.wait_stream(side)
    print(f"tracker(15) alone {t_trk:.2f} ms, segmenter(15) alone {t_seg:.2f} ms, sum {t_trk + t_seg:.2f}")
    print(f"concurrent (segmenter enqueued first) {wall(both):.2f} ms; (tracker enqueued first) {wall(both_rev):.2f} ms")
