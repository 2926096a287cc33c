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


def test_x3_split_arithmetic_on_the_cpu():
    """The arithmetic of csrc/gemm_x3.hip restated in numpy: an fp32 operand as two f16 terms (hi = rn16(v 2^e), lo = rn16(v 2^e
    - hi)), a product as a_lo w_hi + a_hi w_lo + a_hi w_hi.  With exact accumulation the 256-term dot product is within
    2^-22 / sqrt(K)-ish of sum|a||w| of the exact one — an order below the 1e-7 sum|a||w| an fp32 accumulation chain loses —
    and the weight exponent the host picks fills the f16 range without overflow."""
    import numpy as np
    from dvis_plus_amd import functions as Fn
    rng = np.random.default_rng(0)
    K = 256
    a = rng.standard_normal((64, K)).astype(np.float32)
    w = (rng.standard_normal((32, K)) * 0.06).astype(np.float32)
    ew = Fn._x3_exp(torch.from_numpy(w))
    assert 2.0 ** 13 <= float(np.abs(w).max()) * 2.0 ** ew < 2.0 ** 14
    assert Fn._x3_exp(torch.zeros(3, 4)) == 0

    def split(v, e):
        s = (v.astype(np.float64) * 2.0 ** e).astype(np.float32)
        hi = s.astype(np.float16)
        lo = (s - hi.astype(np.float32)).astype(np.float16)
        assert np.isfinite(hi).all()
        return hi.astype(np.float64), lo.astype(np.float64)
    ah, al = split(a, Fn.X3_XEXP)
    wh, wl = split(w, ew)
    got = (al @ wh.T + ah @ wl.T + ah @ wh.T) * 2.0 ** -(Fn.X3_XEXP + ew)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
    assert float((np.abs(got - ref) / scale).max()) < 5e-8
    # the same operands through an fp32 accumulation chain (numpy float32 matmul): the error an fp32 GEMM has anyway
    e32 = float((np.abs((a @ w.T).astype(np.float64) - ref) / scale).max())
    assert e32 > 2e-8


def test_x3_shape_tables():
    """Host-side tables of the split-f16 kernels (no GPU): which shapes are served, how large the packed weights are."""
    from dvis_plus_amd import native
    lib = native.lib()
    for N, K, ok in ((256, 256, 1), (288, 256, 1), (288, 512, 0), (768, 256, 1), (128, 64, 1), (192, 1024, 1), (3072, 1024, 1),
                     (100, 256, 0), (256, 100, 0), (320, 256, 0)):
        assert lib.dvis_x3_linear_supported(N, K, 0) == ok, (N, K)
        # hi + lo halves: 4 bytes per weight (N = 288: the item image carries a tenth, zero, block of 32 — csrc/gemm_x3.hip)
        assert (lib.dvis_x3_packed_bytes(N, K) == (320 if N == 288 else N) * K * 4) == bool(ok)
    assert lib.dvis_x3_linear_supported(256, 256, 1) == 1 and lib.dvis_x3_linear_supported(512, 256, 1) == 0
    assert lib.dvis_x3_ffn_packed_bytes(256, 1024, 256) == 2 * 256 * 1024 * 4 and lib.dvis_x3_ffn_packed_bytes(256, 1000, 256) < 0
    for C, K, ok in ((512, 128, 1), (64, 64, 1), (2048, 512, 1), (96, 128, 0), (256, 100, 0)):
        assert lib.dvis_conv1x1_x3_supported(C, K, 2, 920, 920) == ok
        assert (lib.dvis_conv3x3_x3_packed_bytes(C, K) == 9 * C * K * 4) == bool(ok)
    assert lib.dvis_conv1x1_x3_supported(256, 256, 30, 58880, 58880) == 1       # 1.81 GB: below the 2 GiB of 32-bit buffer offsets
    assert lib.dvis_conv1x1_x3_supported(256, 256, 64, 58880, 58880) == 0       # T = 64 in one call: served by the fp32 kernels
    # row images of the tiled GEMM: the fp32 tensor's size with the rows padded to the row tile of 128
    assert lib.dvis_x3_rows_image_bytes(110430, 1024) == 863 * 128 * 1024 * 4 and lib.dvis_x3_rows_image_bytes(128, 512) == 128 * 512 * 4
    assert lib.dvis_x3_rows_image_bytes(0, 1024) == 0 and lib.dvis_x3_rows_image_bytes(100, 1000) < 0


def test_every_1x1_layer_of_the_r50_at_the_benchmark_shape_is_served_by_the_split_f16_kernel():
    """conv1x1_bias_act hands a 1x1 layer to csrc/conv1x1_x3.hip first when dvis_conv1x1_x3_supported says so: at 30 frames of
    736 x 1280 that is every one of the R50's 36 (16 bottlenecks x (conv1, conv3) + 4 shortcuts, detectron2 BottleneckBlock with
    STRIDE_IN_1X1 False: conv1 reads the block's input map, the stride sits in the 3x3), and the pixel decoder's five projections."""
    from dvis_plus_amd import native
    lib = native.lib()
    layers, cin, hw = [], 64, (184, 320)                      # after the stem + max-pool: 64 channels at stride 4
    for stage, (blocks, mid, out, stride) in enumerate(((3, 64, 256, 1), (4, 128, 512, 2), (6, 256, 1024, 2), (3, 512, 2048, 2))):
        for b in range(blocks):
            s = stride if b == 0 else 1
            ohw = (hw[0] // s, hw[1] // s)
            layers.append((cin, mid, hw, hw))                 # conv1 (stride 1, on the input map)
            layers.append((mid, out, ohw, ohw))               # conv3 (after the strided 3x3)
            if b == 0:
                layers.append((cin, out, hw, ohw))            # shortcut (strided for res3 - res5)
            cin, hw = out, ohw
    assert len(layers) == 36 and hw == (23, 40)
    layers += [(2048, 256, (23, 40), (23, 40)), (1024, 256, (46, 80), (46, 80)), (512, 256, (92, 160), (92, 160)),    # input_proj
               (256, 256, (184, 320), (184, 320)), (256, 256, (184, 320), (184, 320))]                                 # lateral, mask_features
    for C, K, hin, hout in layers:
        assert lib.dvis_conv1x1_x3_supported(C, K, 30, hin[0] * hin[1], hout[0] * hout[1]) == 1, (C, K, hin, hout)


def test_gemm_config_families_share_their_k_split():
    """dvis_gemm_pick_config_nw: whatever the row count, the configuration comes from the family of `nw` waves (its K split) —
    the property that makes a frame's GEMM results independent of how many frames share the call (Fn.linear)."""
    from dvis_plus_amd import native
    lib = native.lib()
    for nw in (1, 4, 8):
        seen = set()
        for M in (1, 100, 300, 700, 3000, 6400, 20000, 579600):
            for N, K in ((256, 256), (2048, 256), (256, 2048), (125, 256), (1536, 512), (512, 512)):
                c = lib.dvis_gemm_pick_config_nw(M, N, K, 1, nw)
                assert 0 <= c < lib.dvis_gemm_num_configs() and lib.dvis_gemm_config_waves(c) == nw, (nw, M, N, K, c)
                seen.add(c)
        assert len(seen) >= 2, "a family should offer more than one tile size"
    assert lib.dvis_gemm_pick_config_nw(100, 256, 256, 1, 3) == -1 and lib.dvis_gemm_config_waves(99) == -1


def test_x3_stage_switches_and_pack_cache():
    from dvis_plus_amd import functions as Fn
    old_x3, old_off = Fn.X3, Fn.X3_OFF
    try:
        Fn.X3, Fn.X3_OFF = True, frozenset({"mask_path"})
        assert Fn.x3_on()
        with Fn.x3_stage("mask_path"):
            assert not Fn.x3_on()
            with Fn.x3_stage("encoder"):
                assert Fn.x3_on()
            assert not Fn.x3_on()
        with Fn.x3_disabled():
            assert not Fn.x3_on()
            with Fn.x3_disabled():
                assert not Fn.x3_on()
            assert not Fn.x3_on()
        assert Fn.x3_on()
        Fn.X3 = False
        assert not Fn.x3_on()
    finally:
        Fn.X3, Fn.X3_OFF = old_x3, old_off
    # least-recently-used eviction, one entry per (weight, kind), re-made when the version key changes
    cache = Fn._PackCache(cap=3)
    made = []
    import unittest.mock as mock
    ws = [torch.zeros(2, 2) for _ in range(5)]
    with mock.patch.object(Fn.X3_GUARD, "word", lambda dev: None), mock.patch.object(Fn.native, "lib") as lib:
        lib.return_value.dvis_x3_set_tag = lambda t: 1
        get = lambda w, kind, ver: cache.get(w, kind, ver, lambda: made.append((id(w), kind, ver)) or (id(w), kind))
        get(ws[0], "linear", 0), get(ws[0], "ffn", 0), get(ws[1], "linear", 0)
        assert len(cache) == 3 and len(made) == 3
        get(ws[0], "linear", 0)                       # hit: refreshed, nothing made
        assert len(made) == 3
        get(ws[2], "linear", 0)                       # evicts the least recently used: (ws[0], "ffn")
        assert len(cache) == 3 and len(made) == 4
        get(ws[0], "linear", 0)
        assert len(made) == 4                         # ... not the one just used
        get(ws[0], "ffn", 0)
        assert len(made) == 5                         # the evicted one is re-made
        get(ws[0], "ffn", 1)
        assert len(made) == 6                         # a new weight version re-packs in place
        tags = {e[3] for e in cache.d.values()}
        assert len(tags) == len(cache.d)              # every packed weight has its own range-guard tag


def test_range_guard_names_the_layer():
    from dvis_plus_amd import functions as Fn
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Sequential(torch.nn.Linear(4, 8)))
    g = Fn._X3RangeGuard()
    tag = g.new_tag(net[1][0].weight, "linear")
    assert g.describe(tag, net) == "1.0.weight (8, 4) [linear kernel]"
    assert "weight" in g.describe(tag, None) and "?" in g.describe(12345, net)
    assert g.snapshot(torch.device("cpu")) is None and g.verify(None) is None
