"""How many host threads should the CPU oracle use on the GPU box (256 logical CPUs, torch default 128)?  (round 6: suite time)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pipeline_parity as PPar  # noqa: E402
from oracle import dvis_torch as O  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

torch.manual_seed(0)
m = build_dvis_plus_r50("offline", task="vps")
sd = PPar.cpu_state(m)
pd, pr = O._sub(sd, "sem_seg_head.pixel_decoder."), O._sub(sd, "sem_seg_head.predictor.")
for nf in (3, 6):
    feats = {k: torch.randn(nf, c, 736 // s, 1280 // s) for k, c, s in (("res2", 256, 4), ("res3", 512, 8), ("res4", 1024, 16), ("res5", 2048, 32))}
    for nt in (128, 64, 32, 16, 8):
        torch.set_num_threads(nt)
        with torch.no_grad():
            t0 = time.time()
            mf, _, ms = O.pixel_decoder_forward(pd, feats, 8, 6)
            t1 = time.time()
            O.decoder_forward(pr, ms, mf, 8, 9)
            t2 = time.time()
        print(f"{nf} frames, {nt} threads: pixel decoder {t1 - t0:.1f} s, decoder {t2 - t1:.1f} s", flush=True)
