"""What does phase B cost when it runs NEXT to the following clip's phase A (stream())?  (dev tool, GPU box)
Variants of stream() over the same clips: full; phase B replaced by a no-op (outputs of a previous pass returned); phase B with the
tracker only / the refiner + masks + post-processing only (the other part's outputs replayed)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

dev = torch.device("cuda", 0)
model = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0).to(dev)
model.allow_input_threshold = True
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
videos = [{"image": bench.synthetic_clip(30, dev, seed=1234 + i), "height": 720, "width": 1280} for i in range(K)]
thr = bench.calibrate_threshold(model, videos[:1], 20)
for v in videos:
    v["object_mask_threshold"] = thr


def run():
    for _ in model.stream(iter(videos[:2])):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = list(model.stream(iter(videos)))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3, outs


full, outs = run()
print(f"stream(), full                                   {full:7.2f} ms per clip")
orig_round, orig_core = model._track_round, model._track_core
cached = {}


def noop(sts):
    return [dict(outs[0]) for _ in sts]


model._track_round = noop
t, _ = run()
print(f"phase B = nothing (phase A alone, streamed)      {t:7.2f} ms per clip   -> phase B costs {full - t:5.2f} ms next to phase A")
model._track_round = orig_round

# tracker only: refiner replaced by a cheap stand-in is not possible without changing shapes; instead time the parts alone
model.tracker_only = True
core_out = {}


def core_tracker_only(embds, embds_nn):
    to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
    track = model.tracker(to_bctq(embds), None, resume=False, frame_embeds_no_norm=to_bctq(embds_nn), need_masks=False)
    if "ref" not in core_out:
        core_out["ref"] = orig_core(embds, embds_nn)
    return core_out["ref"]


model._track_core = core_tracker_only
t2, _ = run()
print(f"phase B = all-gather + tracker + masks + post    {t2:7.2f} ms per clip   (refiner skipped: {full - t2:5.2f} ms)")


def core_refiner_only(embds, embds_nn):
    if "trk" not in core_out:
        to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
        core_out["trk"] = model.tracker(to_bctq(embds), None, resume=False, frame_embeds_no_norm=to_bctq(embds_nn), need_masks=False)
    track = core_out["trk"]
    to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
    from dvis_plus_amd import postprocess as PP
    ref = model.refiner(track["pred_embds"], to_bctq(embds_nn), None, need_masks=False)
    cls, aux = PP.mean_logits(ref["pred_logits"], track["pred_logits"])
    return ref["mask_embed"], cls, aux


model._track_core = core_refiner_only
t3, _ = run()
print(f"phase B = all-gather + refiner + masks + post    {t3:7.2f} ms per clip   (tracker skipped: {full - t3:5.2f} ms)")
