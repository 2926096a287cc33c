"""GPU parity: the HIP MSDA kernels (through the C ABI / MSDeformAttnFunction) vs the oracle and goldens.

Tolerances: fp64 1e-12 (the reference's own fp64 check is allclose defaults, ops/test.py:43);
fp32 atol 1e-5 + rtol 1e-4 — two orders inside the reference's own fp32 acceptance (rtol 1e-2, atol 1e-3,
ops/test.py:59) and inside the 1e-3 of BASELINE.json.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, Golden, make_msda_inputs, level_tensors
from oracle import msda as omsda

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g1_msda_*.npz")))
DEV = "cuda:0"


def _fn():
    from dvis_plus_amd.functions import MSDeformAttnFunction
    return MSDeformAttnFunction


def _tol(dtype):
    return dict(rtol=1e-9, atol=1e-12) if dtype == torch.float64 else dict(rtol=1e-4, atol=1e-5)


def _run(value, s, lsi, loc, w):
    return _fn().apply(value.to(DEV), s.to(DEV), lsi.to(DEV), loc.to(DEV), w.to(DEV), 128).cpu()


@pytest.mark.parametrize("name", CASES)
def test_forward_vs_golden(name):
    g = Golden(name)
    i, o = g.ins, g.outs
    out = _run(i["value"], i["shapes"], i["level_start"], i["loc"], i["w"])
    torch.testing.assert_close(out, o["out"], **_tol(out.dtype))


@pytest.mark.parametrize("name", [c for c in CASES if "reftest" not in c])
def test_backward_vs_golden(name):
    g = Golden(name)
    i, o = g.ins, g.outs
    value, loc, w = (i[k].to(DEV).requires_grad_(True) for k in ("value", "loc", "w"))
    out = _fn().apply(value, i["shapes"].to(DEV), i["level_start"].to(DEV), loc, w, 128)
    out.backward(i["grad_out"].to(DEV))
    tol = _tol(value.dtype) if value.dtype == torch.float64 else dict(rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(value.grad.cpu(), o["grad_value"], **tol)
    torch.testing.assert_close(loc.grad.cpu(), o["grad_loc"], **tol)
    torch.testing.assert_close(w.grad.cpu(), o["grad_w"], **tol)


# (N, M, D, shapes, Lq, P, dtype): tiled fast path (D 32/64, (L,P) in the tiled set), generic path, ragged Lq
SHAPES = [
    (3, 8, 32, [(5, 8), (10, 16), (20, 32)], None, 4, torch.float32),     # R50 layout, Lq = S = 1000 (ragged vs 64)
    (2, 8, 32, [(4, 5), (8, 10), (16, 20), (32, 40)], 130, 4, torch.float32),   # 4 levels (image Mask2Former)
    (2, 16, 64, [(20, 32)], 777, 4, torch.float32),                       # ViT-Adapter extractor layout
    (1, 8, 32, [(5, 8), (10, 16), (20, 32)], 63, 4, torch.float32),
    (1, 8, 32, [(5, 8), (10, 16), (20, 32)], 65, 4, torch.float32),
    (130, 2, 32, [(2, 3)], 3, 4, torch.float32),                          # N = 130: rejected by the reference (N % 128)
    (2, 4, 30, [(6, 4), (3, 2)], 50, 2, torch.float32),                   # generic: odd D
    (2, 2, 71, [(6, 4), (3, 2)], 17, 3, torch.float64),
    (1, 2, 1025, [(3, 2)], 5, 2, torch.float64),                          # the reference's large-D gradcheck sizes
    (1, 8, 32, [(5, 8), (10, 16), (20, 32)], 200, 4, torch.float64),
]


@pytest.mark.parametrize("case", SHAPES, ids=lambda c: f"N{c[0]}M{c[1]}D{c[2]}L{len(c[3])}P{c[5]}q{c[4]}{str(c[6])[-2:]}")
def test_forward_backward_vs_oracle(case):
    N, M, D, shapes, Lq, P, dt = case
    s0, _ = level_tensors(shapes)
    Lq = int(s0.prod(1).sum()) if Lq is None else Lq
    value, s, lsi, loc, w = make_msda_inputs(N, M, D, shapes, Lq, P, dt, seed=N * 7 + D)
    ref = torch.from_numpy(omsda.msda_forward(value, s, lsi, loc, w))
    v, l_, w_ = (t.to(DEV).requires_grad_(True) for t in (value, loc, w))
    out = _fn().apply(v, s.to(DEV), lsi.to(DEV), l_, w_, 128)
    torch.testing.assert_close(out.detach().cpu(), ref, **_tol(dt))
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dt)
    out.backward(go.to(DEV))
    gv, gl, gw = omsda.msda_backward(value, s, lsi, loc, w, go)
    tol = _tol(dt) if dt == torch.float64 else dict(rtol=2e-3, atol=5e-4)
    torch.testing.assert_close(v.grad.cpu(), torch.from_numpy(gv), **tol)
    torch.testing.assert_close(l_.grad.cpu(), torch.from_numpy(gl), **tol)
    torch.testing.assert_close(w_.grad.cpu(), torch.from_numpy(gw), **tol)


def test_gradcheck_like_reference():
    """ops/test.py:66-81: gradcheck in fp64 on the tiny config, one D per backward dispatch class."""
    shapes = [(6, 4), (3, 2)]
    for D in (30, 32, 64, 71):
        value, s, lsi, loc, w = make_msda_inputs(1, 2, D, shapes, 2, 2, torch.float64, seed=D, spread=1.0)
        value = (value * 0.01).to(DEV).requires_grad_(True)
        loc, w = loc.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        assert torch.autograd.gradcheck(_fn().apply, (value, s.to(DEV), lsi.to(DEV), loc, w, 2))


@pytest.mark.parametrize("D", [1025, 2048, 3096])
def test_gradcheck_large_head_dims_like_reference(D):
    """ops/test.py:88-89 also runs D = 1025, 2048, 3096 (the reference needs a dedicated multi-block backward kernel
    there, ms_deform_im2col_cuda.cuh:736-848; here one kernel serves every D).  gradcheck in fast mode (random
    projections of the Jacobian: the full one would be 185 k x 12 k doubles), plus the analytic gradients against the C
    oracle's backward."""
    shapes = [(6, 4), (3, 2)]
    value, s, lsi, loc, w = make_msda_inputs(1, 2, D, shapes, 2, 2, torch.float64, seed=D, spread=1.0)
    v = (value * 0.01).to(DEV).requires_grad_(True)
    l_, w_ = loc.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    # (grad_value is accumulated with atomics: two backward runs differ in the last fp64 bits)
    assert torch.autograd.gradcheck(_fn().apply, (v, s.to(DEV), lsi.to(DEV), l_, w_, 2), fast_mode=True, nondet_tol=1e-10)
    out = _fn().apply(v, s.to(DEV), lsi.to(DEV), l_, w_, 2)
    g = torch.randn(out.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(D))
    out.backward(g.to(DEV))
    gv, gl, gw = omsda.msda_backward(value * 0.01, s, lsi, loc, w, g.numpy())
    torch.testing.assert_close(v.grad.cpu(), torch.from_numpy(gv), rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(l_.grad.cpu(), torch.from_numpy(gl), rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(w_.grad.cpu(), torch.from_numpy(gw), rtol=1e-9, atol=1e-12)


def test_production_frame_720p_vs_oracle():
    """BASELINE config shapes: one 736x1280 frame, S = Lq = 19320, M=8, D=32, L=3, P=4 (fp32)."""
    shapes = [(23, 40), (46, 80), (92, 160)]
    value, s, lsi, loc, w = make_msda_inputs(1, 8, 32, shapes, 19320, 4, torch.float32, seed=5, spread=1.1)
    ref = torch.from_numpy(omsda.msda_forward(value, s, lsi, loc, w))
    out = _run(value, s, lsi, loc, w)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-5)


def test_fused_forward_matches_unfused_module_math():
    """softmax + (ref + off / (W,H)) inside the kernel == torch softmax / arithmetic + plain op."""
    from dvis_plus_amd.functions import msda_fused_forward
    shapes = [(5, 8), (10, 16), (20, 32)]
    N, M, D, L, P = 2, 8, 32, 3, 4
    s, lsi = level_tensors(shapes)
    S = Lq = int(s.prod(1).sum())
    g = torch.Generator().manual_seed(11)
    value = torch.randn(N, S, M, D, generator=g)
    proj = torch.randn(N * Lq, M * L * P * 3 + 4, generator=g) * 2     # one fused projection row: offsets | logits | pad
    offsets, logits = proj[:, :M * L * P * 2], proj[:, M * L * P * 2:M * L * P * 3]
    ref_pts = torch.rand(1, Lq, L, 2, generator=g)
    norm = torch.stack([s[:, 1], s[:, 0]], -1).float()
    loc = ref_pts[:, :, None, :, None, :] + offsets.reshape(N, Lq, M, L, P, 2) / norm[None, None, None, :, None, :]
    w = torch.softmax(logits.reshape(N, Lq, M, L * P), -1).reshape(N, Lq, M, L, P)
    ref = torch.from_numpy(omsda.msda_forward(value, s, lsi, loc.contiguous(), w.contiguous()))
    projd = proj.to(DEV)
    out = msda_fused_forward(value.to(DEV), s.to(DEV), lsi.to(DEV), ref_pts.to(DEV), projd[:, :M * L * P * 2],
                             projd[:, M * L * P * 2:M * L * P * 3], L, P).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-5)
    # per-frame reference points
    ref_n = ref_pts.expand(N, -1, -1, -1).contiguous()
    out2 = msda_fused_forward(value.to(DEV), s.to(DEV), lsi.to(DEV), ref_n.to(DEV), projd[:, :M * L * P * 2],
                              projd[:, M * L * P * 2:M * L * P * 3], L, P).cpu()
    assert torch.equal(out, out2)


def test_fused_forward_position_rows_equal_preadded_query():
    """linear(src + pos) = linear(src) + pos W^T: the kernel adds the per-query position rows to the raw rows."""
    from dvis_plus_amd.functions import msda_fused_forward
    shapes = [(5, 8), (10, 16), (20, 32)]
    N, M, D, L, P = 3, 8, 32, 3, 4
    s, lsi = level_tensors(shapes)
    S = Lq = int(s.prod(1).sum())
    g = torch.Generator().manual_seed(13)
    value = torch.randn(N, S, M, D, generator=g).to(DEV)
    n_off, n_all = M * L * P * 2, M * L * P * 3
    proj = (torch.randn(N * Lq, n_all, generator=g) * 2).to(DEV)
    pos = (torch.randn(Lq, n_all + 4, generator=g)).to(DEV)                   # row stride != width
    ref_pts = torch.rand(1, Lq, L, 2, generator=g).to(DEV)
    a = msda_fused_forward(value, s.to(DEV), lsi.to(DEV), ref_pts, proj[:, :n_off], proj[:, n_off:], L, P,
                           pos_offsets=pos[:, :n_off], pos_logits=pos[:, n_off:n_all])
    summed = (proj.view(N, Lq, n_all) + pos[None, :, :n_all]).view(N * Lq, n_all)
    b = msda_fused_forward(value, s.to(DEV), lsi.to(DEV), ref_pts, summed[:, :n_off], summed[:, n_off:], L, P)
    assert torch.equal(a, b)                                                  # same fp32 additions, same kernel
    with pytest.raises(RuntimeError):
        msda_fused_forward(value, s.to(DEV), lsi.to(DEV), ref_pts, proj[:, :n_off], proj[:, n_off:], L, P,
                           pos_offsets=pos[:, :n_off])


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_half_precision_forward(dt):
    """fp16/bf16 storage, fp32 accumulation (the reference only dispatches float/double)."""
    shapes = [(5, 8), (10, 16)]
    value, s, lsi, loc, w = make_msda_inputs(2, 4, 32, shapes, 90, 4, torch.float32, seed=3, spread=1.0)
    vq, lq, wq = value.to(dt), loc.to(dt), w.to(dt)
    ref = torch.from_numpy(omsda.msda_forward(vq.float(), s, lsi, lq.float(), wq.float()))
    out = _run(vq, s, lsi, lq, wq)
    assert out.dtype == dt
    torch.testing.assert_close(out.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dt,rel", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("M,D,shapes", [(8, 32, [(6, 10), (12, 20), (24, 40)]),      # pixel-decoder layout
                                        (16, 64, [(23, 40)]),                          # ViT-Adapter extractor layout
                                        (8, 64, [(5, 8), (10, 16), (20, 32), (40, 64)])])
def test_half_precision_tiled_kernel(dt, rel, M, D, shapes):
    """fp16 / bf16 storage on the TILED kernel (16 bytes = 8 channels per lane, fp32 accumulation): equal to the fp32
    oracle evaluated on the rounded inputs up to the rounding of the output (one half-precision ulp)."""
    value, s, lsi, loc, w = make_msda_inputs(2, M, D, shapes, 333, 4, torch.float32, seed=17, spread=1.2)
    vq, lq, wq = value.to(dt), loc.to(dt), w.to(dt)
    ref = torch.from_numpy(omsda.msda_forward(vq.float(), s, lsi, lq.float(), wq.float()))
    out = _run(vq, s, lsi, lq, wq)
    assert out.dtype == dt
    torch.testing.assert_close(out.float(), ref, rtol=rel, atol=rel * float(ref.abs().max()) * 0.25)


@pytest.mark.parametrize("dt,rel", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("M,D,shapes,Lq", [(8, 32, [(6, 10), (12, 20), (24, 40)], None),     # encoder self-attention (8 x 8 tiles)
                                           (16, 64, [(23, 40)], 777)])                         # ViT-Adapter extractor layout
def test_fused_forward_half_precision_storage(dt, rel, M, D, shapes, Lq):
    """The fused form (softmax + location arithmetic in the kernel) on fp16 / bf16 value / offsets / logits / output — what the
    module's projections produce under autocast — against the fp32 fused kernel run on the SAME rounded inputs: equal up to
    the rounding of the output (arithmetic is fp32 in both)."""
    from dvis_plus_amd.functions import msda_fused_forward
    N, L, P = 2, len(shapes), 4
    s, lsi = level_tensors(shapes)
    S = int(s.prod(1).sum())
    Lq = Lq or S
    g = torch.Generator().manual_seed(23)
    value = torch.randn(N, S, M, D, generator=g).to(dt)
    proj = (torch.randn(N * Lq, M * L * P * 3 + 8, generator=g) * 1.5).to(dt)          # row stride != width
    ref_pts = torch.rand(1, Lq, L, 2, generator=g)
    n_off = M * L * P * 2
    sh = [tuple(int(v) for v in hw) for hw in shapes] if Lq == S else None
    vd, pd_, rd = value.to(DEV), proj.to(DEV), ref_pts.to(DEV)
    got = msda_fused_forward(vd, s.to(DEV), lsi.to(DEV), rd, pd_[:, :n_off], pd_[:, n_off:n_off + M * L * P], L, P, shapes_host=sh)
    assert got.dtype == dt and got.shape == (N, Lq, M * D)
    pf = pd_.float()
    want = msda_fused_forward(vd.float(), s.to(DEV), lsi.to(DEV), rd, pf[:, :n_off], pf[:, n_off:n_off + M * L * P], L, P,
                              shapes_host=sh)
    torch.testing.assert_close(got.float(), want, rtol=rel, atol=rel * float(want.abs().max()) * 0.25)
    with pytest.raises(RuntimeError, match="dtype"):
        msda_fused_forward(vd, s.to(DEV), lsi.to(DEV), rd, pf[:, :n_off], pf[:, n_off:n_off + M * L * P], L, P)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_msdeformattn_module_under_autocast_takes_the_fused_half_kernel(dt):
    """How the reference evaluates (train_net_video.py:259: torch.autocast): fp32 parameters, projections in half precision.
    The module must take the fused kernel (spy on the C ABI), not a torch formulation, and agree with its fp32 result to
    half-precision accuracy."""
    from dvis_plus_amd import native
    from dvis_plus_amd.pixel_decoder import MSDeformAttn
    torch.manual_seed(5)
    shapes = [(6, 10), (12, 20), (24, 40)]
    s, lsi = level_tensors(shapes)
    S = int(s.prod(1).sum())
    attn = MSDeformAttn(d_model=256, n_levels=3, n_heads=8, n_points=4).to(DEV).eval()
    src = torch.randn(2, S, 256, device=DEV)
    ref = torch.rand(1, S, 3, 2, device=DEV)
    lib = native.lib()
    orig, calls = lib.dvis_msda_fused_forward_h, []
    try:
        lib.dvis_msda_fused_forward_h = lambda *a: (calls.append(a[0]), orig(*a))[1]
        with torch.no_grad():
            want = attn(src, ref, src, s.to(DEV), lsi.to(DEV), None, spatial_shapes_py=shapes)
            with torch.autocast("cuda", dtype=dt):
                got = attn(src, ref, src, s.to(DEV), lsi.to(DEV), None, spatial_shapes_py=shapes)
    finally:
        lib.dvis_msda_fused_forward_h = orig
    assert calls == [native.F16 if dt == torch.float16 else native.BF16] and got.dtype == dt
    tol = 2.0 ** (-7 if dt == torch.float16 else -4) * float(want.abs().max())
    assert float((got.float() - want).abs().max()) <= tol


@pytest.mark.parametrize("dt,rel", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
def test_half_precision_vs_reference_golden_vitl(dt, rel):
    """The reference's own outputs on the ViT-Adapter extractor layout (g1_msda_vitl: D = 64, L = 1, P = 4), inputs rounded
    to half precision: within the storage type's precision of the fp32 reference output."""
    g = Golden("g1_msda_vitl")
    i, o = g.ins, g.outs
    out = _run(i["value"].to(dt), i["shapes"], i["level_start"], i["loc"].to(dt), i["w"].to(dt))
    scale = float(o["out"].abs().max())
    assert float((out.float() - o["out"]).abs().max()) <= 4 * rel * scale      # value, location and weight rounding


def test_edge_cases():
    shapes = [(5, 8), (10, 16), (20, 32)]
    value, s, lsi, loc, w = make_msda_inputs(2, 8, 32, shapes, 100, 4, torch.float32, seed=9)
    # everything outside the maps -> exact zeros (zero padding), including exactly -1 / H boundaries
    far = loc.clone()
    far[:] = 3.0
    assert torch.count_nonzero(_run(value, s, lsi, far, w)) == 0
    # NaN locations contribute nothing (comparison false), like the reference's validity test
    nanloc = loc.clone()
    nanloc[0, 0, 0, 0, 0] = float("nan")
    ref = torch.from_numpy(omsda.msda_forward(value, s, lsi, nanloc, w))
    torch.testing.assert_close(_run(value, s, lsi, nanloc, w), ref, rtol=1e-4, atol=1e-5)
    # non-finite values next to the border must not leak into zero-padded corners
    v2 = value.clone()
    v2[0, 0] = float("inf")
    edge = loc.clone()
    edge[0, :, :, 0, :, :] = 0.999   # bottom-right corner of level 0: high corners are outside
    ref = torch.from_numpy(omsda.msda_forward(v2, s, lsi, edge, w))
    got = _run(v2, s, lsi, edge, w)
    assert torch.equal(torch.isfinite(got), torch.isfinite(ref))
    # empty batch / no queries
    assert _run(value[:0], s, lsi, loc[:0], w[:0]).shape == (0, 100, 256)
    assert _run(value, s, lsi, loc[:, :0], w[:, :0]).shape == (2, 0, 256)


def test_linearity_and_weight_scaling_full_size():
    """Size-independent properties at production size (30 frames): linear in value, linear in weights."""
    shapes = [(23, 40), (46, 80), (92, 160)]
    value, s, lsi, loc, w = make_msda_inputs(4, 8, 32, shapes, 19320, 4, torch.float32, seed=2, spread=1.05)
    f = lambda v, ww: _fn().apply(v, s.to(DEV), lsi.to(DEV), loc.to(DEV), ww, 128)
    v1, v2, wd = value.to(DEV), torch.randn_like(value).to(DEV), w.to(DEV)
    a, b, c = f(v1, wd), f(v2, wd), f(v1 + v2, wd)
    torch.testing.assert_close(c, a + b, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(f(v1, 2 * wd), 2 * a, rtol=1e-6, atol=1e-6)
    # constant value map + weights summing to 1 with all samples inside -> constant output
    ones = torch.ones_like(v1)
    inside = (loc * 0.5 + 0.25).to(DEV)
    out = _fn().apply(ones, s.to(DEV), lsi.to(DEV), inside, wd, 128)
    torch.testing.assert_close(out, torch.ones_like(out), rtol=1e-5, atol=1e-5)


def test_errors_are_loud():
    from dvis_plus_amd.functions import ms_deform_attn_forward
    value, s, lsi, loc, w = make_msda_inputs(1, 2, 4, [(3, 3)], 5, 2, torch.float32, 1)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ms_deform_attn_forward(value, s, lsi, loc, w, 128)                       # CPU tensors: no fallback
    with pytest.raises(RuntimeError, match="GPU tensor"):
        ms_deform_attn_forward(value.to(DEV), s, lsi.to(DEV), loc.to(DEV), w.to(DEV), 128)   # shapes on host
    with pytest.raises(RuntimeError, match="contiguous"):
        ms_deform_attn_forward(value.to(DEV).transpose(2, 3), s.to(DEV), lsi.to(DEV), loc.to(DEV), w.to(DEV), 128)
    with pytest.raises(RuntimeError, match="dtype"):
        ms_deform_attn_forward(value.to(DEV), s.to(DEV), lsi.to(DEV), loc.to(DEV).double(), w.to(DEV), 128)


def test_fused_forward_2d_query_tiling_is_only_a_schedule():
    """8x8 pixel-tile blocks (self-attention geometry, ragged map edges) give the same numbers as linear blocks."""
    from dvis_plus_amd.functions import msda_fused_forward
    for shapes in ([(23, 40), (46, 80), (92, 160)], [(5, 7), (9, 13), (17, 30)]):
        N, M, D, L, P = 2, 8, 32, 3, 4
        s, lsi = level_tensors(shapes)
        S = Lq = int(s.prod(1).sum())
        g = torch.Generator().manual_seed(12)
        value = torch.randn(N, S, M, D, generator=g).to(DEV)
        proj = (torch.randn(N * Lq, M * L * P * 3, generator=g) * 2).to(DEV)
        ref_pts = torch.rand(1, Lq, L, 2, generator=g).to(DEV)
        n_off = M * L * P * 2
        a = msda_fused_forward(value, s.to(DEV), lsi.to(DEV), ref_pts, proj[:, :n_off], proj[:, n_off:], L, P)
        b = msda_fused_forward(value, s.to(DEV), lsi.to(DEV), ref_pts, proj[:, :n_off], proj[:, n_off:], L, P,
                               shapes_host=shapes)
        assert torch.equal(a, b)
        # queries that are NOT the pixels: the hint is ignored (linear blocks)
        c = msda_fused_forward(value, s.to(DEV), lsi.to(DEV), ref_pts[:, :100].contiguous(), proj[:2 * 100, :n_off],
                               proj[:2 * 100, n_off:], L, P, shapes_host=shapes)
        assert c.shape == (N, 100, M * D)


def _centres(shapes):
    return torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w,
                                                 indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in shapes])


@pytest.mark.parametrize("shapes,spread,centred", [
    ([(23, 40), (46, 80), (92, 160)], 0.3, True),     # 720p maps, init-rule ring + small learned part
    ([(23, 40), (46, 80), (92, 160)], 40.0, True),    # most samples land outside the maps
    ([(5, 7), (9, 13), (17, 30)], 1.0, True),         # ragged maps: tiles stick out of every map
    ([(3, 4), (6, 8), (12, 16)], 0.5, True),          # maps smaller than a tile
    ([(23, 40), (46, 80), (92, 160)], 0.5, False),    # reference points that are NOT the pixel centres
])
def test_encoder_geometry_kernel_is_bit_identical_to_tile_kernel(shapes, spread, centred):
    """The encoder's self-attention geometry (queries = pixels, shapes passed on the host) runs 8 x 8 query tiles per
    workgroup instead of 64 consecutive queries: a schedule of the same arithmetic — torch.equal."""
    import math
    from dvis_plus_amd.functions import msda_fused_forward
    N, M, D, L, P = 2, 8, 32, 3, 4
    s, lsi = level_tensors(shapes)
    S = Lq = int(s.prod(1).sum())
    g = torch.Generator().manual_seed(int(spread * 10) + len(shapes))
    value = torch.randn(N, S, M, D, generator=g).to(DEV)
    ang = torch.arange(M) * (2 * math.pi / M)
    d = torch.stack([ang.cos(), ang.sin()], -1)
    d = d / d.abs().max(-1, keepdim=True)[0]
    ring = (d[:, None, None, :] * torch.arange(1, P + 1)[None, None, :, None]).expand(M, L, P, 2)
    off = (ring[None] + spread * torch.randn(N * Lq, M, L, P, 2, generator=g)).reshape(N * Lq, -1)
    lg = torch.randn(N * Lq, M * L * P, generator=g)
    proj = torch.cat([off, lg], 1).to(DEV)
    n_off = M * L * P * 2
    ref = _centres(shapes) if centred else torch.rand(Lq, 2, generator=g)
    ref = ref[None, :, None, :].expand(1, Lq, L, 2).contiguous().to(DEV)
    outs = [msda_fused_forward(value, s.to(DEV), lsi.to(DEV), ref, proj[:, :n_off], proj[:, n_off:], L, P, shapes_host=sh)
            for sh in (shapes, None)]
    assert torch.equal(outs[0], outs[1])
    # and against the oracle (plain op on locations / weights formed with torch)
    w = torch.softmax(lg.view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    norm = torch.tensor([[wd, h] for h, wd in shapes], dtype=torch.float32)
    loc = ref.cpu()[:, :, None, :, None, :] + off.view(N, Lq, M, L, P, 2) / norm[None, None, None, :, None, :]
    want = torch.from_numpy(omsda.msda_forward(value.cpu(), s, lsi, loc.contiguous(), w.contiguous()))
    torch.testing.assert_close(outs[0].cpu(), want, rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("case", [SHAPES[0], SHAPES[2], SHAPES[6]], ids=["r50", "vitl", "odd_d"])
def test_deterministic_backward_repeats_bit_for_bit_and_matches_the_oracle(case):
    """dvis_msda_backward_det: grad_value through 64-bit fixed-point integer atomics — two runs give the same
    bits (the float-atomic form, like the reference's col2im kernels, does not at these sizes), the values are the oracle's to fp32
    accuracy, grad_loc / grad_w equal the default entry point's bit for bit; heavy collisions (every query samples the same cell)
    and a large dynamic range included."""
    from dvis_plus_amd import functions as Fn
    N, M, D, shapes, Lq, P, dt = case
    s0, _ = level_tensors(shapes)
    Lq = int(s0.prod(1).sum()) if Lq is None else Lq
    value, s, lsi, loc, w = make_msda_inputs(N, M, D, shapes, Lq, P, dt, seed=N * 11 + D)
    loc[:, : Lq // 2] = 0.37                                        # half of the queries hit one cell per level: thousands of colliding atomics
    go = torch.randn(N, Lq, M * D, generator=torch.Generator().manual_seed(2), dtype=torch.float64).to(dt)
    go[0, 0] *= 1e4                                                 # dynamic range: the fixed-point scale follows max |grad_out|
    args = [t.to(DEV) for t in (value, s, lsi, loc, w, go)]
    a = Fn.ms_deform_attn_backward(*args, deterministic=True)
    b = Fn.ms_deform_attn_backward(*args, deterministic=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    d = Fn.ms_deform_attn_backward(*args, deterministic=False)
    assert torch.equal(a[1], d[1]) and torch.equal(a[2], d[2])
    # fp64 reference from the same (fp32-valued) inputs; error bound of fp32 contributions (three roundings each) summed exactly:
    # a few eps32 of the cell's sum of |contributions|
    v64, l64, w64, g64 = value.double(), loc.double(), w.double(), go.double()
    ref = torch.from_numpy(omsda.msda_backward(v64, s, lsi, l64, w64, g64)[0])
    bound = torch.from_numpy(omsda.msda_backward(v64, s, lsi, l64, w64.abs(), g64.abs())[0])
    err = (a[0].cpu().double() - ref.double()).abs()
    # + the fixed point's ABSOLUTE resolution: 2^-42 of max |grad_out| x max |w| per contribution
    floor = 1e-9 * float(go.abs().max()) * float(w.abs().max())
    assert bool((err <= 4e-7 * bound.double() + floor).all()), float((err - 4e-7 * bound.double()).max())
    assert float(err.max()) <= 2.0 * float((d[0].cpu().double() - ref.double()).abs().max()) + 1e-12
    # through autograd with torch's switch
    prev = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        v, l_, w_ = (t.to(DEV).requires_grad_(True) for t in (value, loc, w))
        out = _fn().apply(v, s.to(DEV), lsi.to(DEV), l_, w_, 128)
        out.backward(go.to(DEV).view_as(out))
        assert torch.equal(v.grad, a[0])
    finally:
        torch.use_deterministic_algorithms(prev)


def test_deterministic_backward_propagates_non_finite_gradients():
    """A NaN / Inf in grad_output must come back as non-finite grad_value from the fixed-point form too (a diverged
    training run may not be hidden by torch.use_deterministic_algorithms(True)); finite inputs stay finite."""
    from dvis_plus_amd import functions as Fn
    value, s, lsi, loc, w = make_msda_inputs(1, 8, 32, [(6, 10), (12, 20)], 300, 4, torch.float32, seed=3)
    go = torch.randn(1, 300, 8 * 32)
    args = [t.to(DEV) for t in (value, s, lsi, loc, w)]
    assert bool(torch.isfinite(Fn.ms_deform_attn_backward(*args, go.to(DEV), deterministic=True)[0]).all())
    for bad in (float("nan"), float("inf")):
        g = go.clone()
        g[0, 7, 3] = bad
        gv = Fn.ms_deform_attn_backward(*args, g.to(DEV), deterministic=True)[0]
        assert not bool(torch.isfinite(gv).any()), f"grad_output with {bad}: finite grad_value came back"
