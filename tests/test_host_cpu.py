"""CPU: host-side behaviour that needs no GPU — loud failures, library symbols."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, make_msda_inputs


def test_ops_refuse_cpu_tensors():
    from dvis_plus_amd.functions import MSDeformAttnFunction
    value, s, lsi, loc, w = make_msda_inputs(1, 2, 4, [(3, 3)], 5, 2, torch.float32, 1)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        MSDeformAttnFunction.apply(value, s, lsi, loc, w, 128)


def test_library_exports_every_declared_symbol():
    """include/dvis_hip.h <-> libdvis_hip.so <-> native.SIGNATURES stay in sync (no compute calls here)."""
    from dvis_plus_amd import native
    hdr = open(os.path.join(ROOT, "include", "dvis_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dvis_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    if not os.path.exists(native.LIB_PATH):
        import dvis_plus_amd.build as b
        b.build()
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/dvis_hip.h but not exported"
    assert native.lib().dvis_version() >= 100


def test_import_name_shim():
    import MultiScaleDeformableAttention as MSDA
    assert callable(MSDA.ms_deform_attn_forward) and callable(MSDA.ms_deform_attn_backward)


def test_bench_warmup_replays_the_round_structure_of_the_timed_pass():
    """bench.py (multi-GPU, stream()): every round size that occurs in the timed pass also occurs in the warm-up, so no
    MIOpen solver search lands in the timed region."""
    import bench
    for world in (1, 2, 4, 8):
        for steps in (1, 3, 8, 10, 16, 20):
            for warmup in (0, 2, 3):
                w = bench.warmup_clip_count(warmup, steps, world, True, True)
                assert w >= warmup
                if world > 1:
                    timed, warm = set(bench.round_sizes(steps, world)), set(bench.round_sizes(w, world))
                    assert timed <= warm, (world, steps, warmup, timed, warm)
                else:
                    assert w == warmup
    assert bench.warmup_clip_count(2, 10, 8, False, True) == 2 and bench.warmup_clip_count(2, 10, 8, True, False) == 2
    # replicated tracker with two clips per round: a full round, plus the 1-clip round an odd step count ends with
    assert bench.warmup_clip_count(2, 10, 8, False, True, tracker_batch=2) == 2
    assert bench.warmup_clip_count(2, 5, 8, False, True, tracker_batch=2) == 3


def test_convolution_wrappers_keep_torch_semantics_on_cpu_tensors():
    """The own convolution kernels are GPU-only; the wrappers around them (functions.conv3x3_bias_act, conv3x3s2_bias_act,
    conv7x7s2_stem) give CPU tensors to torch's convolution + the same epilogue, and `own=True` refuses instead of falling back."""
    import pytest
    import torch
    import torch.nn.functional as F
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 16, 12, 14, generator=g)
    w = torch.randn(64, 16, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g)
    torch.testing.assert_close(Fn.conv3x3_bias_act(x, w, b, True), F.relu(F.conv2d(x, w, b, 1, 1)))
    torch.testing.assert_close(Fn.conv3x3s2_bias_act(x, w, b, True), F.relu(F.conv2d(x, w, b, 2, 1)))
    xs = torch.randn(1, 3, 16, 20, generator=g)
    ws = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    torch.testing.assert_close(Fn.conv7x7s2_stem(xs, ws), F.conv2d(xs, ws, None, 2, 3))
