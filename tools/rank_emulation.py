"""What ONE rank of an N-rank job does per clip, measured on a 1-GPU box (dev tool):

    python tools/rank_emulation.py [--worlds 1,2,4,8] [--clips 16] [--frames 30]

`clip_shard.EmulatedShard` makes this process rank 0 of `world`: its ceil(T / world) frames of every clip go through
backbone / pixel decoder / decoder, the all-gather of the per-frame queries is replaced by a local copy of the same size,
tracker + refiner run as the schedule says, masks + post-processing cover the rank's frames.  Every rank of the real job
does this same work at the same time, so clips / second of this process is the job's throughput when the collectives cost
nothing — an UPPER bound on what `bench.py --gpus N` can report, and the number to tune a schedule against without a node:

  * north_star's split (frames sharded, ONE all-gather per clip, tracker + refiner replicated), tracker_batch 1 / 2 / 4;
  * tracker-owner rounds (`world` clips per round; clip j's tracker on rank j alone): this process plays a rank that
    owns one clip of every full round.
What is left out: RCCL latency and bandwidth (14.7 MB received per clip at 8 ranks: ~0.1 ms at xGMI rates), rank skew.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dvis_plus_amd.clip_shard import ClipShard, EmulatedShard  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--clips", type=int, default=16)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--candidates", type=int, default=20)
    ap.add_argument("--batches", default="1,2,4")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0).to(device)
    model.allow_input_threshold = True
    T = args.frames
    videos = [{"image": bench.synthetic_clip(T, device, seed=1234 + i), "height": 720, "width": 1280}
              for i in range(args.clips)]
    model._clip_shard = None
    for v in videos[:4]:
        v["object_mask_threshold"] = bench.calibrate_threshold(model, [v], args.candidates)
    for i, v in enumerate(videos[4:]):                                   # (the threshold only sets the candidate count)
        v["object_mask_threshold"] = videos[i % 4]["object_mask_threshold"]

    def run(world, owner, tb):
        model._clip_shard = EmulatedShard(world) if world > 1 else ClipShard()
        model.owner_rounds, model.tracker_batch = owner, tb
        per_round = world if owner else tb
        n = (args.clips // per_round) * per_round
        for _ in model.stream(iter(videos[:max(per_round * 2, 4)])):      # same round structure: every shape seen
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in model.stream(iter(videos[:n])):
            pass
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    base = None
    print(f"per-rank emulation, T = {T}, 720p, R50 offline, VPS, {args.candidates} candidates; ms per clip of ONE rank "
          f"= the N-rank job's time per clip with free collectives")
    print(f"{'ranks':>5} {'schedule':<34} {'ms/clip':>9} {'frames/s (job)':>15} {'vs N x 1-rank':>14}")
    for world in [int(w) for w in args.worlds.split(",")]:
        cases = [("north_star split, tracker_batch %d" % tb, False, tb) for tb in [int(b) for b in args.batches.split(",")]]
        if world > 1:
            cases.append(("tracker-owner rounds", True, 1))
        elif world == 1:
            cases = cases[:1]
        for name, owner, tb in cases:
            if args.clips < (world if owner else tb) * 2:
                continue
            s = run(world, owner, tb)
            fps = T / s
            if base is None:
                base = fps
            print(f"{world:>5} {name:<34} {s * 1e3:9.2f} {fps:15.1f} {fps / (world * base):14.3f}", flush=True)
    model._clip_shard = None


if __name__ == "__main__":
    main()
