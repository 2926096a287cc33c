"""The RCCL path on a ONE-GPU box: a world-size-1 process group on backend "nccl" (= RCCL) with DVIS_FORCE_COLLECTIVES=1, so every
collective of the sharded pipeline is really issued — all_gather_into_tensor of the packed per-frame queries (sync in stream() /
forward(), ASYNC + work.wait() on the tracker stream in the span-pipelined forward), the owner rounds' result all-gather, the VPS
area all-reduce — around the tracker / refiner hipGraphs and the side stream.  Every schedule's outputs must equal the plain
single-GPU run (no process group) bit for bit.  Run by tests/test_shard_gpu.py; prints RCCL_CHECK OK.

    DVIS_FORCE_COLLECTIVES=1 python tools/rccl_single_rank_check.py [--port 29561]
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--port", type=int, default=29561)
    ap.add_argument("--frames", type=int, default=5)
    args = ap.parse_args()
    os.environ["DVIS_FORCE_COLLECTIVES"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import pipeline_parity as PPar
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    m = build_dvis_plus_r50("offline", task="vps", num_classes=20, num_queries=100, n_things=10, enc_layers=2, dec_layers=4,
                            tracker_layers=2, refiner_layers=2, object_mask_threshold=0.06)
    PPar.perturb_msda(m.sem_seg_head.pixel_decoder)
    m = m.to(dev)
    clips = []
    for s in (3, 4, 5):
        g = torch.Generator().manual_seed(s)
        T = args.frames - (s % 2)
        clips.append({"image": [torch.randint(0, 256, (3, 120, 200), dtype=torch.uint8, generator=g).to(dev) for _ in range(T)],
                      "height": 120, "width": 200})

    def grab(o):
        return {"masks": o["pred_masks"].clone(), "segs": o["segments_infos"], "ids": list(o["pred_ids"])}

    # ---- the plain single-GPU run: no process group, no collective
    assert not dist.is_initialized() and not m.clip_shard.force
    want = [grab(m([c])) for c in clips]
    assert any(w["segs"] for w in want), "degenerate check: no segment anywhere"
    m.pipeline_rounds = 2                           # (the span-pipelined forward, still without a process group)
    want2 = [grab(m([c])) for c in clips]
    m.pipeline_rounds = 1

    # ---- world 1 on RCCL, every collective forced
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    m._clip_shard = None
    assert m.clip_shard.force and m.clip_shard.world == 1
    calls = {"all_gather": 0, "all_gather_async": 0, "all_reduce": 0}
    ag, ar = dist.all_gather_into_tensor, dist.all_reduce

    def ag_spy(out, inp, group=None, async_op=False):
        assert out.is_cuda and inp.is_cuda
        calls["all_gather_async" if async_op else "all_gather"] += 1
        return ag(out, inp, group=group, async_op=async_op)

    def ar_spy(t, *a, **k):
        calls["all_reduce"] += 1
        return ar(t, *a, **k)
    dist.all_gather_into_tensor, dist.all_reduce = ag_spy, ar_spy
    bad = 0

    def check(tag, outs, ref=None):
        nonlocal bad
        for i, (o, w) in enumerate(zip(outs, ref or want)):
            ok = torch.equal(o["masks"], w["masks"]) and o["segs"] == w["segs"] and o["ids"] == w["ids"]
            print(f"rank 0: {tag}: clip {i}: equal to the no-collective run = {ok}", flush=True)
            bad += not ok
    try:
        for owner in (True, False):
            m.owner_rounds = owner
            before = dict(calls)
            outs = [grab(o) for o in m.stream(clips)]
            torch.cuda.synchronize()
            check(f"stream(), tracker-owner rounds {'on' if owner else 'off'}", outs)
            n = calls["all_gather"] - before["all_gather"]
            # owner rounds: per round of `world` = 1 clip one query all-gather + one result all-gather; replicated: one per clip
            want_n = 2 * len(clips) if owner else len(clips)
            print(f"rank 0:   all_gather_into_tensor calls on RCCL: {n} (expected {want_n}), all_reduce {calls['all_reduce'] - before['all_reduce']}",
                  flush=True)
            bad += n != want_n
        m.owner_rounds = True
        check("forward() (one all-gather per clip)", [grab(m([c])) for c in clips])
        # the span-pipelined forward: ASYNC all-gather per span, work.wait() on the tracker stream, tracker hipGraph per span
        m.pipeline_rounds = 2
        before = calls["all_gather_async"]
        outs = [grab(m([c])) for c in clips]
        torch.cuda.synchronize()
        m.pipeline_rounds = 1
        n_async = calls["all_gather_async"] - before
        print(f"rank 0:   async all_gather_into_tensor calls on RCCL: {n_async} (2 spans x {len(clips)} clips)", flush=True)
        bad += n_async != 2 * len(clips)
        check("forward() with pipeline_rounds = 2 (async all-gather + stream wait) vs the same schedule without a group", outs, want2)
        for t in (torch.ones(3, device=dev),):
            dist.all_reduce(t)
            assert torch.equal(t, torch.ones(3, device=dev))
        # the range-guard decision is COLLECTIVE in a sharded run: the rank's guard word rides in the clip's all-gather and every
        # rank raises from the gathered tags (a rank raising alone would leave the others in the next collective)
        from dvis_plus_amd import functions as Fn
        snap_fn = m._guard_snapshot

        def flagged():
            snap = snap_fn()
            return None if snap is None else (*snap[:3], torch.full_like(snap[3], 1))      # tag 1 = the first packed weight
        m._guard_snapshot = flagged
        before = calls["all_gather"]
        try:
            m([clips[0]])
            print("rank 0: a flagged guard word did NOT raise in the sharded run", flush=True)
            bad += 1
        except Fn.X3RangeError as e:
            ok = calls["all_gather"] - before == 1            # ... raised AFTER the gather, from the gathered tags
            print(f"rank 0: flagged guard word -> X3RangeError after the clip's all-gather = {ok} ({str(e)[:60]}...)", flush=True)
            bad += not ok
        finally:
            m._guard_snapshot = snap_fn
        check("forward() after the guard error (state intact)", [grab(m([c])) for c in clips])
    finally:
        dist.all_gather_into_tensor, dist.all_reduce = ag, ar
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_CHECK", "OK" if not bad else "FAILED", f"backend=nccl world=1 collectives={calls}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
