"""Run one sub-stage repeatedly (for rocprofv3 kernel stats):  python tools/pd_only.py {backbone|pixel_decoder|decoder|tracker} [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_clip  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "pixel_decoder"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
m = build_dvis_plus_r50("offline").to(dev)
clip = synthetic_clip(30, dev)
with torch.no_grad():
    images, _ = m.preprocess(clip)
    feats = m.backbone(images)
    pd = m.sem_seg_head.pixel_decoder
    mf, _, ms = pd.forward_features(feats)
    out = m.sem_seg_head.predictor(ms, mf)
    torch.cuda.synchronize()
    for _ in range(reps):
        if what == "backbone":
            m.backbone(images)
        elif what == "pixel_decoder":
            pd.forward_features(feats)
        elif what == "decoder":
            m.sem_seg_head.predictor(ms, mf)
    torch.cuda.synchronize()
