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
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0 and f"SHARD_CHECK OK world={world}" in r.stdout, tail
