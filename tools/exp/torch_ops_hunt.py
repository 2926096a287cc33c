"""Which torch (aten) ops still launch device kernels in a clip, and from where (dev tool).
    python tools/exp/torch_ops_hunt.py [stage]      stage: all | backbone | pixel_decoder | decoder
One clip at the benchmark's shape under torch.profiler: every aten op that owns device time, with its input shapes and the
innermost frame of dvis_plus_amd/ that issued it, sorted by device time.  The package's own kernels (ctypes launches) do not
appear: they are not aten ops."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synthetic_clip  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda", 0)
m = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0).to(dev).eval()
clip = synthetic_clip(30, dev)


def run():
    with torch.no_grad():
        images, _ = m.preprocess(clip)
        feats = m.backbone(images)
        if stage == "backbone":
            return
        mf, _, ms = m.sem_seg_head.pixel_decoder.forward_features(feats)
        if stage == "pixel_decoder":
            return
        m.sem_seg_head.predictor(ms, mf)


run()
run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    run()
    torch.cuda.synchronize()

rows = collections.defaultdict(lambda: [0.0, 0])
for ev in prof.events():
    t = getattr(ev, "self_device_time_total", 0.0) or 0.0
    if t <= 0 or not ev.name.startswith("aten::"):
        continue
    where = "?"
    for fr in (ev.stack or []):
        if "dvis_plus_amd/" in fr:
            where = fr.split("dvis_plus_amd/")[-1]
            break
    shapes = str([s for s in (ev.input_shapes or []) if s])[:90]
    r = rows[(ev.name, where, shapes)]
    r[0] += t
    r[1] += 1
tot = sum(r[0] for r in rows.values())
print(f"stage = {stage}: {tot / 1e3:.2f} ms of device time in aten ops")
for (name, where, shapes), (t, n) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{t / 1e3:8.3f} ms {n:4d} x  {name:28s} {where:60s} {shapes}")
