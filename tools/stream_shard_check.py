"""Development check for stream() under frame sharding on ONE GPU box: N ranks on cuda:0 (collectives through gloo, or
RCCL when N == 1 with DVIS_FORCE_COLLECTIVES=1), a few distinct full-size clips through model.stream(); afterwards rank 0
re-runs every clip unsharded and compares the concatenated per-rank masks and the segment lists bit for bit.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29533 \
      tools/stream_shard_check.py --clips 4 --frames 6
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=4)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--out", default="gpurun_out/shard_check")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group(os.environ.get("DVIS_DIST_BACKEND", "gloo"))
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    torch.manual_seed(0)
    model = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.008).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    clips = []
    for c in range(args.clips):
        T = args.frames - (c % 2)                 # ragged lengths
        clips.append({"image": torch.randint(0, 256, (T, 3, args.height, args.width), generator=g).float().to(dev),
                      "height": args.height, "width": args.width})
    stash = []
    finish = model._finish_phase

    def spy(st, mask_embed, cls, aux):
        stash.append((mask_embed.clone(), cls.clone(), aux.clone()))
        return finish(st, mask_embed, cls, aux)
    model._finish_phase = spy

    def run():
        stash.clear()
        outs = [{"masks": o["pred_masks"].cpu(), "segs": o["segments_infos"], "ids": o["pred_ids"],
                 "fr": o["frame_ids"]} for o in model.stream(clips)]
        torch.cuda.synchronize()
        return outs, [tuple(t.cpu() for t in s_) for s_ in stash]

    def compare(x, y):
        """(largest |difference| of the refined mask embeddings / class logits, fraction of differing panoptic pixels)"""
        d = max(float((a - b).abs().max()) for sx, sy in zip(x[1], y[1]) for a, b in zip(sx, sy))
        # (the two schedules split a ragged clip differently — owner rounds rotate the split, one clip per round does
        # not — so a rank's pixel maps are comparable only where it holds the same frames in both)
        px = max([float((a["masks"] != b["masks"]).float().mean()) for a, b in zip(x[0], y[0])
                  if a["fr"] == b["fr"] and a["masks"].numel()] or [0.0])
        return d, px, all(a["segs"] == b["segs"] and a["ids"] == b["ids"] for a, b in zip(x[0], y[0]))
    run()                                          # warm-up: library algorithm choices settle on the first call
    own1, own2 = run(), run()                      # rounds of `world` clips, one tracker rank per clip
    model.owner_rounds = False
    repl = run()                                   # tracker replicated on every rank
    outs = own2[0]
    d0, p0, s0 = compare(own1, own2)
    d1, p1, s1 = compare(own2, repl)
    print(f"rank {rank}: owner rounds run-to-run: max|d|={d0:.2e} pixels={p0:.2e} segments_equal={s0}; "
          f"owner vs replicated tracker: max|d|={d1:.2e} pixels={p1:.2e} segments_equal={s1}", flush=True)
    for ci in range(len(clips)):
        f = lambda x, y: float((x[0][ci]["masks"] != y[0][ci]["masks"]).float().mean()) \
            if x[0][ci]["masks"].numel() and x[0][ci]["fr"] == y[0][ci]["fr"] else 0.
        print(f"  rank {rank} clip {ci} frames {own2[0][ci]['fr']}: pixels run-to-run {f(own1, own2):.2e}, "
              f"owner vs replicated {f(own2, repl):.2e}, ids {own2[0][ci]['ids']} / {repl[0][ci]['ids']}", flush=True)
    # run-to-run noise (library kernels with atomics) is 3e-6 .. 1e-4; the owner's and the replicated tracker are different
    # hipGraph captures (the library may pick other GEMM algorithms) and the tracker feeds its own output back 6 layers x T
    # frames: 1e-4 .. 1e-3 observed on embeddings of order 1.  Segment lists and ids must be EQUAL.
    same = s1 and d1 <= 5e-3
    flag = torch.tensor([0 if same else 1], device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(flag)
    os.makedirs(args.out, exist_ok=True)
    torch.save(outs, os.path.join(args.out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    model._clip_shard = None                       # forget the (forced) collectives of the destroyed group
    if rank == 0:
        parts = [torch.load(os.path.join(args.out, f"rank{r}.pt")) for r in range(world)]
        bad = int(flag.item())
        for ci, clip in enumerate(clips):
            single = model([clip])
            masks = torch.cat([p[ci]["masks"] for p in sorted(parts, key=lambda p: (p[ci]["fr"] or [1 << 30])[0])], 0)
            same = torch.equal(masks, single["pred_masks"].cpu())
            diff = (masks != single["pred_masks"].cpu()).float().mean().item()
            segs = all(p[ci]["segs"] == single["segments_infos"] and p[ci]["ids"] == single["pred_ids"] for p in parts)
            print(f"clip {ci}: T={len(clip['image'])} frames/rank={[len(p[ci]['fr']) for p in parts]} "
                  f"segments={len(single['segments_infos'])} masks_equal={same} (differing pixels {diff:.2e}) "
                  f"segments_equal={segs}")
            bad += (not segs) or diff > 3e-3   # other batch sizes -> other conv / GEMM algorithms -> a few tie pixels flip
        # ---- and against the ORACLE (not only against the unsharded product): the sharded ranks' concatenated panoptic
        # map of clip 0 vs the CPU oracle's windowed pipeline on the same frames and weights (from the backbone outputs on)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import pipeline_parity as PPar
        sd = PPar.cpu_state(model)
        clip0 = clips[0]
        ref, stages = PPar.run_oracle(model, sd, [f for f in clip0["image"].cpu()], offline=True, task="vps",
                                      object_mask_threshold=model.object_mask_threshold, out_hw=(args.height, args.width))
        masks0 = torch.cat([p[0]["masks"] for p in sorted(parts, key=lambda p: (p[0]["fr"] or [1 << 30])[0])], 0)
        sharded = {"pred_masks": masks0, "segments_infos": parts[0][0]["segs"], "pred_ids": parts[0][0]["ids"]}
        try:
            n = PPar.compare_vps(sharded, ref, stages, f"sharded stream() over {world} ranks, clip 0 vs ORACLE")
            print(f"clip 0 vs oracle: segment lists equal, {n} differing pixels, all within the 1e-3 logit allowance")
        except AssertionError as e:
            print("clip 0 vs oracle FAILED:", str(e)[:500])
            bad += 1
        print("SHARD_CHECK", "OK" if not bad else "FAILED", f"world={world}")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
