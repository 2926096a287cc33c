"""GPU: the frame-sharded path with world > 1 on REAL hardware — N ranks on the one GPU of the box, collectives through
gloo (RCCL refuses two ranks on one device; the model code above the process group is the same), a few ragged clips
through model.stream() with tracker-owner rounds and with the replicated tracker; rank 0 then re-runs every clip
unsharded and compares the per-rank masks and the segment lists (tools/stream_shard_check.py prints the comparison).
Covers what the world_size-2 / -8 gloo tests on CPU cannot: hipGraph replay, the side stream and the HIP kernels under
sharding, a rank that gets no frame of a clip, ragged shards.  BASELINE config #4's structure at a size that fits a test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,frames,port", [(2, 5, 29541), (3, 4, 29542)])
def test_sharded_stream_on_one_gpu(world, frames, port, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DVIS_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "stream_shard_check.py"),
           "--clips", "3", "--frames", str(frames), "--out", str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    # on failure: the CHECKER's lines (rank … / clip … / SHARD_CHECK / [intcmp]), not torch-elastic's traceback
    own = [l for l in r.stdout.splitlines() if l.lstrip().startswith(("rank", "clip", "SHARD_CHECK", "[intcmp]", "product", "oracle"))]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Traceback" in l or "assert" in l][-10:]
    tail = "\n".join(own[-40:] + ["--- stderr (filtered) ---"] + err)
    assert r.returncode == 0 and f"SHARD_CHECK OK world={world}" in r.stdout, tail


def test_rccl_collectives_on_a_single_rank():
    """The RCCL path itself, in the driver's suite: backend "nccl" (= RCCL), world size 1, DVIS_FORCE_COLLECTIVES=1 — the packed
    query all-gather (sync and async + wait on the tracker stream), the owner rounds' result all-gather and the VPS all-reduce are
    really issued, around the tracker / refiner hipGraphs; stream() with owner rounds on and off, forward(), the span-pipelined
    forward: torch.equal to the run without a process group (tools/rccl_single_rank_check.py)."""
    env = dict(os.environ, DVIS_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_single_rank_check.py"), "--port", "29543"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    own = [l for l in r.stdout.splitlines() if l.startswith(("rank", "RCCL_CHECK"))]
    err = [l for l in r.stderr.splitlines() if "Error" in l or "error" in l or "Traceback" in l or "assert" in l][-10:]
    assert r.returncode == 0 and "RCCL_CHECK OK backend=nccl world=1" in r.stdout, "\n".join(own[-30:] + ["--- stderr ---"] + err)


def test_bench_forced_collectives_runs_on_rccl():
    """`bench.py --gpus 1` with DVIS_FORCE_COLLECTIVES=1: the process group the driver's N > 1 runs will use (backend nccl,
    device_id set, barrier + max-over-ranks all-reduce around the timed region, all_gather_object of the rank table) on one rank."""
    import json
    env = dict(os.environ, DVIS_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29544")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--frames", "6",
                        "--no-extra", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, tail
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dist"]["backend"] == "nccl" and d["dist"]["world_size"] == 1 and d["value"] > 0, tail


def test_bench_self_launches_two_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2` without a launcher around it starts its two ranks itself (bench.launch_command; the
    reference's analogue is detectron2's launch(main, args.num_gpus), train_net_video.py:322-329).  On a one-GPU box the ranks
    share the device over gloo and the line says so; what is asserted is everything the first RCCL run will need to get
    right above the transport: rc 0, ONE JSON line from rank 0, dist.world_size == 2, both measurements present (the
    headline = north_star's split, and `owner_rounds`), the metric and the workload of BASELINE.json."""
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0, tail
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, tail
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["unit"] == "frames/s" and d["value"] > 0
    assert d["metric"].startswith("frames/sec DVIS++ R50 offline, 720p T=30")
    assert d["dist"]["world_size"] == 2 and len(d["dist"]["ranks"]) == 2
    assert d["dist"]["ranks_share_one_gpu"] is True and d["dist"]["backend"] == "gloo"      # this box has one GPU
    assert d["owner_rounds"]["value"] > 0 and d["config"]["tracker_owner_rounds"] is False
    assert "frames sharded 2-way" in d["config"]["workload"]
