"""CPU: bench.py's launcher logic.  `python bench.py --gpus N` with no launcher around it must start N ranks itself
(the reference's analogue: detectron2 launch(main, args.num_gpus, ...), train_net_video.py:322-329) instead of quietly
measuring one rank."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_is_one_rank_per_gpu_on_localhost():
    import bench
    cmd, env = bench.launch_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], n_devices=8, port=29777)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    assert cmd[-7].endswith("bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "DVIS_BENCH_ONE_DEVICE" not in env          # 8 GPUs for 8 ranks: RCCL, one device each
    # a development box with one GPU: the ranks share it over gloo (functional check, flagged in the JSON line)
    _, env1 = bench.launch_command(2, ["--gpus", "2"], n_devices=1, port=1)
    assert env1["DVIS_BENCH_ONE_DEVICE"] == "1" and env1["DVIS_DIST_BACKEND"] == "gloo"


def test_gpus_flag_without_launcher_reexecs(monkeypatch):
    import bench
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n: calls.append(n) or 7)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert calls == [4] and e.value.code == 7          # the launcher's exit code is bench.py's exit code


def test_self_launch_runs_the_launch_command(monkeypatch):
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    assert bench.self_launch(2) == 0
    assert "--nproc-per-node=2" in seen["cmd"] and seen["cmd"][-4:] == ["--gpus", "2", "--steps", "1"]
    port = int(seen["cmd"][seen["cmd"].index("--master-port") + 1])
    assert 1024 < port < 65536


def test_under_a_launcher_world_size_must_match_gpus(monkeypatch):
    """WORLD_SIZE set (torch.distributed.run started us): no re-exec; the first thing main() needs is a GPU."""
    import torch
    import bench
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(bench, "self_launch", lambda n: pytest.fail("must not re-exec under a launcher"))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    if not torch.cuda.is_available():
        import faulthandler
        try:
            with pytest.raises(AssertionError, match="needs a GPU"):
                bench.main()
        finally:
            faulthandler.cancel_dump_traceback_later()      # main() arms a watchdog first; not in the test process


def test_roofline_accounting_follows_the_survey_formulas():
    """mask contraction 2 Q C HW flops and 4 (Q C + C HW + Q HW) bytes per frame (3.015 GFLOP / 83.9 MB at
    Q = 100, C = 256, 184 x 320); masked cross-attention 4 Q HW_l C flops per frame and layer."""
    import bench
    a = bench._acct_mask_logits(None, None, 1, 100, 256, 58880, None, None)
    assert a["flops"] == 2 * 100 * 256 * 58880 == 3_014_656_000 and a["bytes"] == 4 * (100 * 256 + 256 * 58880 + 100 * 58880) == 83_947_520
    p = bench._acct_mask_pooled(None, None, 30, 100, 256, 23, 40, None, None, None)
    assert p["flops"] == 2.0 * 30 * 100 * 256 * 920 and p["form"] == 2
    import ctypes
    m = bench._acct_attention(None, None, None, None, None, None, None, None, ctypes.c_void_p(16), None, 30, 8, 100, 14720, 32, 0.1, None, None, 0)
    assert m["flops"] == 4.0 * 30 * 100 * 14720 * 256 and m["masked"] and m["long"]
    s = bench._acct_attention(None, None, None, None, None, None, None, None, None, None, 30, 8, 100, 100, 32, 0.1, None, None, 0)
    assert not s["masked"] and not s["long"]


def test_committed_pmc_traffic_files_are_readable():
    """bench.py fills `traffic` of the roofline entries from profiles/ (it cannot read PMCs itself)."""
    import bench
    t = bench.load_x3_traffic()
    for fam in ("conv1x1_x3_kernel", "x3_ffn_kernel"):
        assert fam in t and t[fam]["hbm_bytes_per_launch"] > 0 and "FETCH_SIZE" in t[fam]["source"]
