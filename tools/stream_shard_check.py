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
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import pipeline_parity as PPar
    torch.manual_seed(0)
    model = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0)
    # decisive masks (as trained ones are) and a deformable attention with something to do: with random-init mask heads every
    # sigmoid is 0.5 +- 0.005, every pixel of every comparison below would sit "within tolerance" and prove nothing
    PPar.perturb_msda(model.sem_seg_head.pixel_decoder)
    PPar.sharpen_masks(model, PPar.SHARPEN)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(1)
    clips = []
    for c in range(args.clips):
        T = args.frames - (c % 2)                 # ragged lengths
        # low-frequency pattern + noise (bench.synthetic_clip's recipe at this size): non-degenerate, spatially coherent masks
        yy, xx = torch.meshgrid(torch.linspace(0, 6.28, args.height), torch.linspace(0, 6.28, args.width), indexing="ij")
        noise = torch.randint(0, 256, (T, 3, args.height, args.width), generator=g).float()
        pat = torch.stack([(127 + 100 * torch.sin(xx * (1 + (t + c) % 3) + 0.2 * t) * torch.cos(yy * 2 + 0.1 * t)).clamp(0, 255)
                           for t in range(T)])
        img = (noise * 0.3 + pat[:, None] * 0.7).floor()
        clips.append({"image": img.to(dev), "height": args.height, "width": args.width})
    # per-clip score threshold that sends 8 queries to the panoptic stage (random-init class scores are near-uniform); rank
    # 0's value for everybody (the calibration pass itself runs sharded: same collectives on every rank)
    for clip in clips:
        thr = [bench.calibrate_threshold(model, [clip], 8, slack=3)]     # (at the widest score gap among 5 .. 11 candidates)
        dist.broadcast_object_list(thr, src=0)
        clip["object_mask_threshold"] = thr[0]
    model.allow_input_threshold = True          # (the per-clip thresholds above are honoured)
    model.overlap_threshold = 0.0               # every candidate that wins a pixel becomes a segment: whole arg-max map compared
    stash, gathered = [], []
    finish = model._finish_phase
    gather = model.clip_shard.all_gather_frames

    def spy(st, mask_embed, cls, aux):
        stash.append((mask_embed.clone(), cls.clone(), aux.clone()))
        return finish(st, mask_embed, cls, aux)

    def gather_spy(parts, T, **kw):
        out = gather(parts, T, **kw)
        gathered.append([t.clone() for t in out[:2]])
        return out
    model._finish_phase = spy
    model.clip_shard.all_gather_frames = gather_spy

    def run():
        stash.clear()
        gathered.clear()
        outs = [{"masks": o["pred_masks"].cpu(), "segs": o["segments_infos"], "ids": o["pred_ids"],
                 "fr": o["frame_ids"]} for o in model.stream(clips)]
        torch.cuda.synchronize()
        return outs, list(stash), list(gathered)

    def ids_match_oracle(segs, ids_a, ids_b):
        """Product vs ORACLE (other arithmetic, logits equal to ~1e-5): the query behind every THING segment must be the same.  A
        stuff segment merges every candidate of its class (inference_video_vps, dvis_Plus/meta_architecture.py:909-925) and reports
        the FIRST of them that wins a pixel — a candidate that wins two pixels or none decides that, so between two
        implementations that differ in the last bits it is not defined."""
        return len(ids_a) == len(ids_b) == len(segs) and all(x == y or not sg.get("isthing", True)
                                                             for sg, x, y in zip(segs, ids_a, ids_b))

    def segs_match(a, b, npix=0):
        """Two schedules' results for one clip: the SAME segment lists and query ids, exactly.  Phase A is frame-invariant
        (tests/test_phase_a_invariance_gpu.py: a frame's bits do not depend on its batch mates) and phase B deterministic, so
        another batch shape / another rank count may not move a single area or tie pixel.  (Round 4 compared areas with a
        tie-pixel allowance and skipped the representative query of merged STUFF segments: the segmenter's library GEMMs and the
        attention kernels' batch-sized key splits gave other batch shapes other last bits.)"""
        return a["segs"] == b["segs"] and list(a["ids"]) == list(b["ids"])

    def same_frames_equal(x, y):
        """Both schedules' panoptic maps, where a rank holds the same frames in both (owner rounds rotate the ragged split)."""
        return all(torch.equal(a["masks"], b["masks"]) for a, b in zip(x[0], y[0]) if a["fr"] == b["fr"])
    run()                                          # warm-up
    own1, own2 = run(), run()                      # rounds of `world` clips, one tracker rank per clip
    bad = 0
    # (1) run to run: no library kernel, no atomics, no run-dependent split anywhere in the pipeline — the gathered queries
    # (phase A), the tracker / refiner results (phase B) and every output must repeat bit for bit, with N processes on one GPU.
    in_eq = all(torch.equal(a, b) for gx, gy in zip(own1[2], own2[2]) for a, b in zip(gx, gy))
    out_eq = all(torch.equal(a, b) for sx, sy in zip(own1[1], own2[1]) for a, b in zip(sx, sy))
    r2r = in_eq and out_eq and all(segs_match(a, b) and torch.equal(a["masks"], b["masks"]) for a, b in zip(own1[0], own2[0]))
    print(f"rank {rank}: run-to-run: gathered queries (phase A) bit-identical={in_eq}, tracker / refiner results bit-identical={out_eq}",
          flush=True)
    # (2) the property north_star's split relies on (no broadcast of the replicated results): the tracker + refiner of clip
    # j, run by ITS OWNER RANK in the owner rounds and received here through the all-gather, against THIS rank's own
    # replicated run (_track_core) from the same gathered queries -> torch.equal, on every rank, for every clip.
    model.keep = False
    cross = True
    for ci, ((emb, emb_nn), (me_o, cls_o, aux_o)) in enumerate(zip(own2[2], own2[1])):
        with torch.no_grad():       # (under autograd the op front-ends take torch's library kernels: other bits)
            me, cls, aux = model._track_core(emb, emb_nn)
        T = me.shape[1]
        eq = torch.equal(me, me_o[:, :T]) and torch.equal(cls, cls_o) and torch.equal(aux, aux_o)
        if not eq:
            print(f"  rank {rank} clip {ci}: owner's results differ from this rank's replicated run: max|d| embeddings "
                  f"{float((me - me_o[:, :T]).abs().max()):.2e}, class logits {float((cls - cls_o).abs().max()):.2e}", flush=True)
        cross &= eq
    # (3) the replicated schedule end to end (one clip per round: other merged batch shapes in phase A): same segment lists
    model.owner_rounds = False
    repl = run()
    segs_eq = all(segs_match(a, b, max(1, a["masks"].numel())) for a, b in zip(own2[0], repl[0]))
    # ... and the SAME gathered per-frame queries: the ranks' frames went through other batch shapes (a merged round batch vs one
    # clip's shard) — phase A's frame invariance, across processes
    gq_eq = len(own2[2]) == len(repl[2]) and all(torch.equal(a, b) for gx, gy in zip(own2[2], repl[2]) for a, b in zip(gx, gy))
    segs_eq = segs_eq and gq_eq
    print(f"rank {rank}: gathered queries of the owner rounds == of the replicated schedule (other batch shapes) (torch.equal)={gq_eq}",
          flush=True)
    if not segs_eq:
        for ci, (a, b) in enumerate(zip(own2[0], repl[0])):
            if not segs_match(a, b, max(1, a["masks"].numel())):
                print(f"  rank {rank} clip {ci}: owner rounds ids {a['ids']} segs {a['segs']} | replicated ids {b['ids']} segs {b['segs']}",
                      flush=True)
    print(f"rank {rank}: owner rounds run-to-run consistent={r2r}; owner's tracker results == this rank's replicated "
          f"tracker from the same gathered queries (torch.equal)={cross}; owner vs replicated schedule segment lists equal="
          f"{segs_eq}, maps equal on shared frames={same_frames_equal(own2, repl)}", flush=True)
    outs = own2[0]
    maps_eq = same_frames_equal(own2, repl)
    same = r2r and cross and segs_eq and maps_eq
    flag = torch.tensor([0 if same else 1], device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(flag)
    os.makedirs(args.out, exist_ok=True)
    torch.save(outs, os.path.join(args.out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    model._finish_phase = finish
    model._clip_shard = None                       # forget the (forced) collectives of the destroyed group
    if rank == 0:
        parts = [torch.load(os.path.join(args.out, f"rank{r}.pt")) for r in range(world)]
        bad = int(flag.item())
        for ci, clip in enumerate(clips):
            single = model([clip])
            masks = torch.cat([p[ci]["masks"] for p in sorted(parts, key=lambda p: (p[ci]["fr"] or [1 << 30])[0])], 0)
            same = torch.equal(masks, single["pred_masks"].cpu())
            diff = (masks != single["pred_masks"].cpu()).float().mean().item()
            single_d = {"segs": single["segments_infos"], "ids": single["pred_ids"]}
            segs = all(segs_match(p[ci], single_d, masks.numel()) for p in parts)
            print(f"clip {ci}: T={len(clip['image'])} frames/rank={[len(p[ci]['fr']) for p in parts]} "
                  f"segments={len(single['segments_infos'])} masks_equal={same} (differing pixels {diff:.2e}) "
                  f"segments_equal={segs}")
            bad += (not segs) or not same      # sharded over N ranks == unsharded on one, bit for bit (frame-invariant phase A)
        # ---- and against the ORACLE (not only against the unsharded product): the sharded ranks' concatenated panoptic
        # map of clip 0 vs the CPU oracle's windowed pipeline on the same frames and weights (from the backbone outputs on),
        # with decisive masks: only pixels the MEASURED logit error can explain may differ, and they must be a small minority
        sd = PPar.cpu_state(model)
        clip0 = clips[0]
        ref, stages = PPar.run_oracle(model, sd, [f for f in clip0["image"].cpu()], offline=True, task="vps",
                                      object_mask_threshold=clip0["object_mask_threshold"], overlap_threshold=0.0,
                                      out_hw=(args.height, args.width))
        masks0 = torch.cat([p[0]["masks"] for p in sorted(parts, key=lambda p: (p[0]["fr"] or [1 << 30])[0])], 0)
        same_ids = ids_match_oracle(parts[0][0]["segs"], parts[0][0]["ids"], ref[2]) if len(ref[1]) == len(parts[0][0]["segs"]) else False
        sharded = {"pred_masks": masks0, "segments_infos": parts[0][0]["segs"],
                   "pred_ids": ref[2] if same_ids else parts[0][0]["ids"]}      # (stuff segments: see ids_match_oracle)
        tol = PPar.logit_tolerance(float(stages["masks"][stages["vps_query_ids"]].abs().max()))
        try:
            assert len(ref[1]) > 0, "degenerate check: the oracle keeps no segment"
            n = PPar.compare_vps(sharded, ref, stages, f"sharded stream() over {world} ranks, clip 0 vs ORACLE", tol_logit=tol,
                                 max_count=int(0.02 * masks0.numel()))
            print(f"clip 0 vs oracle: {len(ref[1])} segments, lists equal; {n} of {masks0.numel()} pixels differ, each within the "
                  f"logit allowance {tol:.1e} (|logit| up to {float(stages['masks'][stages['vps_query_ids']].abs().max()):.1f})")
        except AssertionError as e:
            print("clip 0 vs oracle FAILED:", str(e)[:500])
            print("   product ids", sharded["pred_ids"], "segments", sharded["segments_infos"], "\n   oracle ids", ref[2], "segments",
                  ref[1], "\n   oracle candidate queries", stages["vps_query_ids"].tolist(), "threshold", clip0["object_mask_threshold"])
            bad += 1
        print("SHARD_CHECK", "OK" if not bad else "FAILED", f"world={world}")
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
