"""Where do the 160 s of tests/test_pipeline_720p_gpu.py::test_T30_natural_logit_scale... go?  (round 6: GPU suite time)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import pipeline_parity as PPar  # noqa: E402
from oracle import dvis_torch as O  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

acc = {}


def timed(mod, name):
    fn = getattr(mod, name)

    def w(*a, **k):
        t0 = time.time()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.time() - t0
        return r
    setattr(mod, name, w)


for n in ("pixel_decoder_forward", "decoder_forward", "refiner_forward", "inference_video_vps", "preprocess", "post_processing"):
    timed(O, n)
timed(O.Tracker, "forward")
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count(), flush=True)
DEV = "cuda:0"
t0 = time.time()
m = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0)
PPar.perturb_msda(m.sem_seg_head.pixel_decoder)
PPar.sharpen_masks(m, 2.0)
sd = PPar.cpu_state(m)
m = m.to(DEV)
clip = bench.synthetic_clip(30, torch.device(DEV), seed=1234)
video = {"image": clip, "height": 720, "width": 1280}
print("build + clip %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
m.object_mask_threshold = bench.calibrate_threshold(m, [video], 20)
m.overlap_threshold = 0.0
m.debug_stages = {}
out = m([video])
torch.cuda.synchronize()
print("product (calibration + run) %.1f s" % (time.time() - t0), flush=True)
bb = PPar.gpu_backbone(m)
tb = [0.0]


def bb_timed(x):
    t = time.time()
    r = bb(x)
    tb[0] += time.time() - t
    return r


PPar.gpu_backbone = lambda m_: bb_timed
t0 = time.time()
ref, stages = PPar.run_oracle(m, sd, [f for f in clip.cpu()], offline=True, task="vps", attn_masks=True,
                              object_mask_threshold=m.object_mask_threshold, overlap_threshold=0.0, out_hw=(720, 1280))
print("oracle total %.1f s; gpu backbone + D2H %.1f s; parts %s" % (time.time() - t0, tb[0], {k: round(v, 1) for k, v in acc.items()}), flush=True)
t0 = time.time()
with torch.no_grad():
    all_logits = m.debug_stages["mask_fn"](None)
rows = PPar.error_budget(m.debug_stages, stages, all_logits, "t30")
print("error budget %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
PPar.compare_vps(out, ref, stages, "t30", tol_logit=PPar.TOL_LOGIT)
print("compare_vps %.1f s" % (time.time() - t0), flush=True)
# the same oracle with the SEGMENTER batched over 10 frames (frames are the batch; the tracker still walks windows of 3)
for ws in (10,):
    acc.clear()
    t0 = time.time()
    with torch.no_grad():
        O.dvis_plus_forward(sd, bb_timed, [f for f in clip.cpu()][:12], offline=True, task="vps", window_size=ws, stages={},
                            object_mask_threshold=m.object_mask_threshold, overlap_threshold=0.0, out_hw=(720, 1280))
    print("12 frames, window %d: %.1f s parts %s" % (ws, time.time() - t0, {k: round(v, 1) for k, v in acc.items()}), flush=True)
acc.clear()
t0 = time.time()
with torch.no_grad():
    O.dvis_plus_forward(sd, bb_timed, [f for f in clip.cpu()][:12], offline=True, task="vps", window_size=3, stages={},
                        object_mask_threshold=m.object_mask_threshold, overlap_threshold=0.0, out_hw=(720, 1280))
print("12 frames, window 3: %.1f s parts %s" % (time.time() - t0, {k: round(v, 1) for k, v in acc.items()}), flush=True)
