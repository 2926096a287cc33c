"""Per-stage wall time of the offline pipeline at the BASELINE shape (dev tool):  python tools/stage_times.py [T]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_clip  # noqa: E402
from dvis_plus_amd import postprocess as PP  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
m = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0).to(dev)
clip = synthetic_clip(T, dev)


def timed(name, fn, acc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return out


def run(acc):
    with torch.no_grad():
        images, img_size = timed("preprocess", lambda: m.preprocess(clip), acc)
        feats = timed("backbone", lambda: m.backbone(images), acc)
        pd = m.sem_seg_head.pixel_decoder
        mf, _, ms = timed("pixel_decoder", lambda: pd.forward_features(feats), acc)
        out = timed("decoder", lambda: m.sem_seg_head.predictor(ms, mf), acc)
        to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
        embds = out["pred_embds"][0].permute(1, 2, 0)
        embds_nn = out["pred_embds_without_norm"][0].permute(1, 2, 0)
        track = timed("tracker", lambda: m.tracker(to_bctq(embds), None, resume=False,
                                                   frame_embeds_no_norm=to_bctq(embds_nn), need_masks=False), acc)
        ref = timed("refiner", lambda: m.refiner(track["pred_embds"], to_bctq(embds_nn), None, need_masks=False), acc)
        cls, aux = PP.mean_logits(ref["pred_logits"], track["pred_logits"])
        s = torch.softmax(cls, -1).max(-1)[0].sort(descending=True)[0]
        thr = float((s[19] + s[20]) / 2)
        mask_fn = lambda idx: m.refiner.predict_masks(ref["mask_embed"], mf.unsqueeze(0), idx)[0]
        timed("postprocess(20 cand.)", lambda: PP.inference_video_vps(
            cls, mask_fn, img_size, (720, 1280), images.shape[-2:], 124, 58, thr, 0.8, aux, num_frames=T), acc)


run({})
run({})
acc = {}
N = 3
for _ in range(N):
    run(acc)
tot = sum(acc.values())
for k, v in acc.items():
    print(f"{k:24s} {v / N:9.2f} ms/clip  {v / N / T:7.3f} ms/frame  {100 * v / tot:5.1f}%")
print(f"{'total':24s} {tot / N:9.2f} ms/clip  -> {T / (tot / N / 1e3):.1f} frames/s (stage-synchronised)")
