"""csrc/gemm_x3.hip: the deformable encoder's dense layers (ops/modules/ms_deform_attn.py:96-117, msdeformattn.py:103-131)
with every fp32 operand as two f16 terms on the f16 matrix cores, three products per pair, fp32 accumulation.

The claim these tests pin: the results are fp32-GRADE — measured against fp64 the error of the split kernels is not larger
than the error of an fp32 GEMM of the same operands (whose own accumulation rounding dominates both) — and the layout is
exact: integer-valued operands come out bit for bit.  Plus the edges: ragged row counts, strided rows, rows of very
different magnitude, the f16 range, run-to-run bits, a token's result independent of what else is in the batch."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def _lin(K, N, seed, bias_std=1.0):
    torch.manual_seed(seed)
    lin = nn.Linear(K, N).to(DEV)
    nn.init.xavier_uniform_(lin.weight)
    nn.init.normal_(lin.bias, 0.0, bias_std)
    return lin


def _norm(C, seed):
    torch.manual_seed(seed)
    n = nn.LayerNorm(C).to(DEV)
    nn.init.normal_(n.weight, 1.0, 0.3)
    nn.init.normal_(n.bias, 0.0, 0.2)
    return n


def _rel(c, ref, scale):
    return float(((c.double() - ref).abs() / scale).max())


@pytest.mark.parametrize("N", [128, 192, 256, 288, 768])
@pytest.mark.parametrize("M", [1, 33, 127, 129, 5000])
def test_linear_error_is_that_of_an_fp32_gemm(N, M):
    from dvis_plus_amd import functions as Fn
    lin = _lin(256, N, N + M)
    x = torch.randn(M, 256, device=DEV)
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    scale = x.double().abs() @ lin.weight.double().abs().t() + lin.bias.double().abs()
    e_lib = _rel(F.linear(x, lin.weight, lin.bias), ref, scale)
    for relu in (False, True):
        got = Fn.x3_linear(x, lin.weight, lin.bias, relu=relu)
        assert got.shape == (M, N)
        e = _rel(got, ref.clamp_min(0) if relu else ref, scale)
        # fp32 GEMMs measure 2.5 - 3.5e-7 here; the split kernels 1.5 - 2.5e-7
        assert e <= max(1.25 * e_lib, 3e-7), (e, e_lib)


def test_layout_is_exact_on_integer_operands():
    """Small integers and a signed permutation-like weight are exactly representable in the hi term alone: every product and
    every partial sum is an integer below 2^24, so the result must equal the fp64 one bit for bit — any fragment / k-order /
    block mix-up shows as a wrong integer (asymmetric operands: G9 of the HIP guide)."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(5)
    for N in (256, 288):
        lin = nn.Linear(256, N).to(DEV)
        lin.weight.copy_(torch.randint(-3, 4, (N, 256), generator=g).float())
        lin.bias.copy_(torch.randint(-50, 50, (N,), generator=g).float())
        x = torch.randint(-40, 41, (777, 256), generator=g).float().to(DEV)
        ref = (x.double() @ lin.weight.double().t() + lin.bias.double()).float()
        assert torch.equal(Fn.x3_linear(x, lin.weight, lin.bias), ref)
    l1, l2, norm = nn.Linear(256, 1024).to(DEV), nn.Linear(1024, 256).to(DEV), nn.LayerNorm(256).to(DEV)
    l1.weight.copy_(torch.randint(-2, 3, (1024, 256), generator=g).float())
    l1.bias.copy_(torch.randint(-20, 20, (1024,), generator=g).float())
    l2.weight.copy_(torch.randint(-2, 3, (256, 1024), generator=g).float())
    l2.bias.copy_(torch.randint(-20, 20, (256,), generator=g).float())
    x = torch.randint(-3, 4, (300, 256), generator=g).float().to(DEV)
    pre = x.double() + F.relu(x.double() @ l1.weight.double().t() + l1.bias.double()) @ l2.weight.double().t() + l2.bias.double()
    assert float(pre.abs().max()) < 2 ** 24 and float(F.relu(x.double() @ l1.weight.double().t() + l1.bias.double()).max()) < 4000
    ref = F.layer_norm(pre, (256,), norm.weight.double(), norm.bias.double(), norm.eps)
    got = Fn.x3_ffn_ln(x, l1, l2, norm)
    assert float((got.double() - ref).abs().max()) < 2e-6          # exact pre-norm sums; only the LayerNorm rounds


def test_linear_strided_rows_and_magnitudes():
    from dvis_plus_amd import functions as Fn
    lin = _lin(256, 256, 3, bias_std=0.0)
    big = torch.randn(400, 320, device=DEV)
    x = big[:, 32:288]                                                   # row stride 320, 128-byte offset
    x = x * torch.logspace(-3, 2.5, 400, device=DEV)[:, None]            # |x| from 1e-3 to ~1300 per row
    ref = x.double() @ lin.weight.double().t()
    scale = x.double().abs() @ lin.weight.double().abs().t()
    assert _rel(Fn.x3_linear(x, lin.weight, lin.bias), ref, scale) <= 4e-7


def test_f16_range_is_loud():
    """|x * 2^xexp| beyond the f16 range (|x| >= 4094 at the default exponent) cannot be split: the result is not finite —
    never a silently clipped number.  A smaller exponent serves such data."""
    from dvis_plus_amd import functions as Fn
    lin = _lin(256, 256, 4)
    x = torch.randn(64, 256, device=DEV)
    x[3, 17] = 9000.0
    out = Fn.x3_linear(x, lin.weight, lin.bias)
    assert not torch.isfinite(out[3]).all() and torch.isfinite(out[:3]).all() and torch.isfinite(out[4:]).all()
    with pytest.raises(Fn.X3RangeError):          # ... and the kernels' range guard saw it (tests/test_x3_range_guard_gpu.py)
        Fn.X3_GUARD.check_now(x.device)
    out = Fn.x3_linear(x, lin.weight, lin.bias, xexp=0)
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    scale = x.double().abs() @ lin.weight.double().abs().t() + lin.bias.double().abs()
    # (at exponent 0 the low terms of |x| < 0.125 are f16 subnormals: an absolute floor of 2^-25 per element instead of
    # 2^-29 — measured 6e-7 of sum|a||w| on unit-normal rows, the price of the extra range)
    assert _rel(out, ref, scale) <= 2e-6


@pytest.mark.parametrize("M", [1, 129, 4001])
@pytest.mark.parametrize("with_pos", [False, True])
def test_linear_ln(M, with_pos):
    from dvis_plus_amd import functions as Fn
    lin, norm = _lin(256, 256, 11), _norm(256, 12)
    S = max(1, M // 3) if with_pos else M
    frames = M // S
    x = torch.randn(frames, S, 256, device=DEV)
    res = torch.randn(frames, S, 256, device=DEV)
    pos = torch.randn(1, S, 256, device=DEV) if with_pos else None
    ref = F.layer_norm(res.double() + x.double() @ lin.weight.double().t() + lin.bias.double(), (256,), norm.weight.double(),
                       norm.bias.double(), norm.eps)
    lib = norm(res + F.linear(x, lin.weight, lin.bias))
    e_lib = float((lib.double() - ref).abs().max())
    r = Fn.x3_linear_ln(x, lin.weight, lin.bias, res, norm, pos=pos)
    out = r[0] if with_pos else r
    assert float((out.double() - ref).abs().max()) <= max(1.25 * e_lib, 3e-6)
    if with_pos:
        assert float((r[1].double() - (ref + pos.double())).abs().max()) <= max(1.25 * e_lib, 4e-6)


@pytest.mark.parametrize("M", [1, 33, 300, 9000])
def test_ffn_ln(M):
    from dvis_plus_amd import functions as Fn
    l1, l2, norm = _lin(256, 1024, 21, 0.5), _lin(1024, 256, 22, 0.5), _norm(256, 23)
    x = torch.randn(M, 256, device=DEV)
    h = F.relu(x.double() @ l1.weight.double().t() + l1.bias.double())
    ref = F.layer_norm(x.double() + h @ l2.weight.double().t() + l2.bias.double(), (256,), norm.weight.double(), norm.bias.double(),
                       norm.eps)
    lib = norm(x + l2(F.relu(l1(x))))
    e_lib = float((lib.double() - ref).abs().max())
    out = Fn.x3_ffn_ln(x, l1, l2, norm)
    assert float((out.double() - ref).abs().max()) <= max(1.25 * e_lib, 4e-6)
    pos = torch.randn(1, M, 256, device=DEV)
    o1, o2 = Fn.x3_ffn_ln(x.view(1, M, 256), l1, l2, norm, pos=pos)
    assert torch.equal(o1.view(M, 256), out)
    assert float((o2.double() - (ref + pos.double())).abs().max()) <= max(1.25 * e_lib, 5e-6)


def test_a_tokens_result_does_not_depend_on_the_batch_and_bits_repeat():
    """What frame sharding relies on: a rank that holds 2 frames of a clip gets, for its tokens, the bits the unsharded run
    gets (the library GEMM picks other kernels for other row counts; these kernels have one summation order per token)."""
    from dvis_plus_amd import functions as Fn
    l1, l2, norm, lin = _lin(256, 1024, 31, 0.5), _lin(1024, 256, 32, 0.5), _norm(256, 33), _lin(256, 288, 34)
    x = torch.randn(6000, 256, device=DEV)
    full_f, full_l = Fn.x3_ffn_ln(x, l1, l2, norm), Fn.x3_linear(x, lin.weight, lin.bias)
    assert torch.equal(full_f, Fn.x3_ffn_ln(x, l1, l2, norm)) and torch.equal(full_l, Fn.x3_linear(x, lin.weight, lin.bias))
    for a, b in ((0, 1), (17, 1000), (4097, 6000), (5999, 6000)):
        part = x[a:b].contiguous()
        assert torch.equal(Fn.x3_ffn_ln(part, l1, l2, norm), full_f[a:b])
        assert torch.equal(Fn.x3_linear(part, lin.weight, lin.bias), full_l[a:b])


def test_encoder_layer_takes_the_split_kernels_and_matches_the_fp32_path():
    """The module-level switch: the same MSDeformAttnTransformerEncoderLayer with DVIS_X3 on and off (fp32 library GEMMs +
    add_layernorm kernel), at a 2-frame 96 x 160 pyramid — and a spy that the three entry points really ran."""
    from dvis_plus_amd import functions as Fn, native
    from dvis_plus_amd.pixel_decoder import MSDeformAttnTransformerEncoderLayer, MSDeformAttnTransformerEncoder
    torch.manual_seed(0)
    layer = MSDeformAttnTransformerEncoderLayer(256, 1024, 0.0, "relu", 3, 8, 4)
    for p in layer.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    layer.self_attn._reset_parameters()
    layer = layer.to(DEV).eval()
    with torch.no_grad():
        layer.self_attn.attention_weights.weight.normal_(0, 0.05)
        layer.self_attn.sampling_offsets.weight.normal_(0, 0.02)
    shapes = [(3, 5), (6, 10), (12, 20)]
    S = sum(h * w for h, w in shapes)
    src, pos = torch.randn(2, S, 256, device=DEV), torch.randn(1, S, 256, device=DEV)
    enc = MSDeformAttnTransformerEncoder(lambda: layer, 1)
    ref_pts = enc.reference_points_unpadded(shapes, DEV)
    ss = torch.as_tensor(shapes, dtype=torch.long, device=DEV)
    lsi = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    lib, calls = native.lib(), []
    orig = {n: getattr(lib, n) for n in ("dvis_x3_linear", "dvis_x3_linear_add", "dvis_x3_linear_ln", "dvis_x3_ffn_ln")}
    for n, f in orig.items():
        setattr(lib, n, (lambda n, f: lambda *a: (calls.append(n), f(*a))[1])(n, f))
    try:
        assert Fn.X3
        out_x3, q_x3 = layer(src, pos, ref_pts, ss, lsi, None, shapes_py=shapes, emit_next_query=True)
    finally:
        for n, f in orig.items():
            setattr(lib, n, f)
    # value_proj; offsets | logits with `src + pos` formed inside the kernel; output_proj + norm1; the FFN + norm2
    assert calls.count("dvis_x3_linear") == 1 and calls.count("dvis_x3_linear_add") == 1
    assert calls.count("dvis_x3_linear_ln") == 1 and calls.count("dvis_x3_ffn_ln") == 1
    assert q_x3 is None            # no layer needs the (N, S, C) tensor out + pos any more
    Fn.X3 = False
    try:
        out_f32, q_f32 = layer(src, pos, ref_pts, ss, lsi, None, shapes_py=shapes, emit_next_query=True)
    finally:
        Fn.X3 = True
    assert float((out_x3 - out_f32).abs().max()) < 2e-5 and float((q_f32 - (out_f32 + pos)).abs().max()) < 1e-6


@pytest.mark.parametrize("Ci,Co,H,W,stride,N,res,relu", [
    (512, 128, 23, 40, 1, 3, False, True),        # conv1 of a res3 block; 920 pixels per image: tiles straddle images
    (128, 512, 46, 80, 1, 1, True, True),         # conv3 + shortcut + ReLU
    (256, 1024, 12, 20, 1, 2, True, True),
    (2048, 512, 23, 40, 1, 2, False, True),       # 32 chunks of 64 input channels
    (512, 2048, 7, 9, 1, 3, True, False),         # 8 passes of 256 output channels, 63 pixels per image
    (1024, 2048, 23, 41, 2, 2, False, False),     # the stride-2 shortcut, odd width
    (256, 512, 45, 80, 2, 1, False, False),       # ... odd height
    (64, 256, 20, 31, 1, 2, True, True),          # one chunk
    (256, 64, 33, 47, 1, 2, False, True),         # 64 output channels (res2 conv1)
    (64, 64, 37, 50, 1, 2, False, True),          # res2.0 conv1: one chunk, one block pair
    (256, 128, 30, 44, 1, 2, False, True),        # res3.0 conv1 (reads the stride-4 map)
    (64, 256, 25, 36, 1, 3, False, False),        # res2.0 shortcut: no residual, no ReLU
])
def test_conv1x1_x3(Ci, Co, H, W, stride, N, res, relu):
    """csrc/conv1x1_x3.hip against fp64, next to the fp32 library convolution's error on the same operands."""
    from dvis_plus_amd import functions as Fn
    torch.manual_seed(Ci + Co + H)
    x = torch.randn(N, Ci, H, W, device=DEV)
    w = torch.randn(Co, Ci, 1, 1, device=DEV) * (2.0 / Ci) ** 0.5
    b = torch.randn(Co, device=DEV)
    OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
    r = torch.randn(N, Co, OH, OW, device=DEV) if res else None
    assert Fn.conv1x1_x3_ok(x, w, stride, r)
    xs = x[:, :, ::stride, ::stride]
    ref = F.conv2d(xs.double(), w.double(), b.double()) + (r.double() if res else 0)
    scale = F.conv2d(xs.double().abs(), w.double().abs(), b.double().abs()) + (r.double().abs() if res else 0)
    lib = F.conv2d(xs.contiguous(), w, b) + (r if res else 0)
    if relu:
        ref, lib = ref.clamp_min(0), lib.clamp_min(0)
    got = Fn.conv1x1_x3(x, w, b, r, relu, stride)
    assert got.shape == ref.shape
    e, e_lib = _rel(got, ref, scale), _rel(lib, ref, scale)
    assert e <= max(1.25 * e_lib, 3e-7), (e, e_lib)
    assert torch.equal(got, Fn.conv1x1_x3(x, w, b, r, relu, stride))


@pytest.mark.parametrize("C,C2,Co,H,W,s2,N", [
    (64, 64, 256, 46, 80, 1, 2),          # res2.0: conv3 64 -> 256 + shortcut 64 -> 256, same resolution
    (128, 256, 512, 23, 40, 2, 3),        # res3.0: shortcut reads the stride-4 map with stride 2
    (256, 512, 1024, 12, 20, 2, 2),       # res4.0
    (512, 1024, 2048, 6, 10, 2, 1),       # res5.0
    (128, 256, 512, 23, 39, 2, 2),        # odd input sizes: (45 x 77) -> (23 x 39)
])
def test_conv1x1_x3_dual_is_conv3_plus_shortcut(C, C2, Co, H, W, s2, N):
    """dvis_conv1x1_x3_dual: relu(conv3(a) + b3 + shortcut(x)[::s] + bs) as one accumulation, against fp64 next to the fp32
    library's error; integer operands: exact (the concatenated k order and the second source's stride-2 geometry)."""
    from dvis_plus_amd import functions as Fn
    torch.manual_seed(C + C2 + H)
    H2, W2 = (2 * H - (1 if W % 2 else 0), 2 * W - (1 if W % 2 else 0)) if s2 == 2 else (H, W)
    a = torch.randn(N, C, H, W, device=DEV).relu()
    x = torch.randn(N, C2, H2, W2, device=DEV).relu()
    w3 = torch.randn(Co, C, 1, 1, device=DEV) * (2.0 / C) ** 0.5
    ws = torch.randn(Co, C2, 1, 1, device=DEV) * (2.0 / C2) ** 0.5
    b3, bs = torch.randn(Co, device=DEV), torch.randn(Co, device=DEV)
    assert Fn.conv1x1_x3_dual_ok(a, w3, x, ws, s2)
    xs = x[:, :, ::s2, ::s2]
    ref = (F.conv2d(a.double(), w3.double(), b3.double()) + F.conv2d(xs.double(), ws.double(), bs.double())).clamp_min(0)
    scale = F.conv2d(a.double().abs(), w3.double().abs(), b3.double().abs()) + F.conv2d(xs.double().abs(), ws.double().abs(), bs.double().abs())
    lib = (F.conv2d(a, w3, b3) + F.conv2d(xs.contiguous(), ws, bs)).clamp_min(0)
    got = Fn.conv1x1_x3_dual(a, w3, b3, x, ws, bs, relu=True, stride2=s2)
    assert got.shape == ref.shape
    e, e_lib = _rel(got, ref, scale), _rel(lib, ref, scale)
    assert e <= max(1.25 * e_lib, 3e-7), (e, e_lib)
    assert torch.equal(got, Fn.conv1x1_x3_dual(a, w3, b3, x, ws, bs, relu=True, stride2=s2))
    # ... and a frame's result does not depend on its batch mates
    assert torch.equal(got[:1], Fn.conv1x1_x3_dual(a[:1].contiguous(), w3, b3, x[:1].contiguous(), ws, bs, relu=True, stride2=s2))
    g = torch.Generator().manual_seed(3)
    ai = torch.randint(-20, 21, (N, C, H, W), generator=g).float().to(DEV)
    xi = torch.randint(-20, 21, (N, C2, H2, W2), generator=g).float().to(DEV)
    w3i = torch.randint(-3, 4, (Co, C, 1, 1), generator=g).float().to(DEV)
    wsi = torch.randint(-3, 4, (Co, C2, 1, 1), generator=g).float().to(DEV)
    bi = torch.randint(-50, 50, (Co,), generator=g).float().to(DEV)
    want = (F.conv2d(ai.double(), w3i.double(), bi.double()) + F.conv2d(xi[:, :, ::s2, ::s2].double(), wsi.double())).float()
    assert torch.equal(Fn.conv1x1_x3_dual(ai, w3i, bi, xi, wsi, None, relu=False, stride2=s2), want)


def test_conv1x1_x3_layout_is_exact_on_integer_operands():
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(9)
    for Ci, Co, H, W, stride in ((192, 256, 9, 13, 1), (128, 128, 10, 7, 2), (320, 512, 5, 11, 1)):
        x = torch.randint(-30, 31, (3, Ci, H, W), generator=g).float().to(DEV)
        w = torch.randint(-3, 4, (Co, Ci, 1, 1), generator=g).float().to(DEV)
        b = torch.randint(-50, 50, (Co,), generator=g).float().to(DEV)
        OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
        r = torch.randint(-99, 99, (3, Co, OH, OW), generator=g).float().to(DEV)
        ref = (F.conv2d(x[:, :, ::stride, ::stride].double(), w.double(), b.double()) + r.double()).float()
        assert torch.equal(Fn.conv1x1_x3(x, w, b, r, False, stride), ref)


@pytest.mark.parametrize("K,N,M", [(64, 128, 300), (512, 256, 1000), (1024, 512, 777), (192, 192, 4097)])
def test_linear_streams_any_k(K, N, M):
    """The streaming projection kernel takes K in chunks of 64: any multiple (the ViT-Adapter's 1024-wide layers included)."""
    from dvis_plus_amd import functions as Fn
    lin = _lin(K, N, K + N)
    x = torch.randn(M, K, device=DEV)
    assert Fn.x3_ok(x, N, K)
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    scale = x.double().abs() @ lin.weight.double().abs().t() + lin.bias.double().abs()
    e_lib = _rel(F.linear(x, lin.weight, lin.bias), ref, scale)
    got = Fn.x3_linear(x, lin.weight, lin.bias)
    assert _rel(got, ref, scale) <= max(1.25 * e_lib, 3e-7)
    assert torch.equal(got, Fn.x3_linear(x, lin.weight, lin.bias))


@pytest.mark.parametrize("Ci,Co,H,W,stride,N,relu", [
    (64, 64, 17, 23, 1, 2, True),
    (128, 128, 23, 40, 2, 3, True),       # conv2 of the first res3 bottleneck (stride 2), tiles straddle images
    (256, 256, 12, 20, 1, 2, False),
    (512, 512, 9, 11, 2, 2, True),        # odd sizes
    (64, 256, 32, 32, 1, 1, False),
])
def test_conv3x3_x3(Ci, Co, H, W, stride, N, relu):
    """The nine-tap form of csrc/conv1x1_x3.hip (3x3, padding 1) against fp64, next to the library convolution's error."""
    from dvis_plus_amd import functions as Fn
    torch.manual_seed(Ci + H)
    x = torch.randn(N, Ci, H, W, device=DEV)
    w = torch.randn(Co, Ci, 3, 3, device=DEV) * (2.0 / (9 * Ci)) ** 0.5
    b = torch.randn(Co, device=DEV)
    assert Fn.conv3x3_x3_ok(x, w, stride)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride, 1)
    scale = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), stride, 1)
    lib = F.conv2d(x, w, b, stride, 1)
    if relu:
        ref, lib = ref.clamp_min(0), lib.clamp_min(0)
    got = Fn.conv3x3_x3(x, w, b, None, relu, stride)
    assert got.shape == ref.shape
    e, e_lib = _rel(got, ref, scale), _rel(lib, ref, scale)
    assert e <= max(1.25 * e_lib, 3e-7), (e, e_lib)
    xi = torch.randint(-9, 10, (N, Ci, H, W), device=DEV).float()
    wi = torch.randint(-2, 3, (Co, Ci, 3, 3), device=DEV).float()
    assert torch.equal(Fn.conv3x3_x3(xi, wi, None, None, False, stride), F.conv2d(xi.double(), wi.double(), None, stride, 1).float())


def test_reserved_cus_do_not_change_bits():
    """dvis_x3_set_reserve only changes how many persistent workgroups share the tiles: same bits for every reserve."""
    from dvis_plus_amd import functions as Fn, native
    l1, l2, norm = _lin(256, 1024, 41, 0.5), _lin(1024, 256, 42, 0.5), _norm(256, 43)
    x = torch.randn(40000, 256, device=DEV)
    xc = torch.randn(2, 256, 46, 80, device=DEV)
    w = torch.randn(256, 256, 3, 3, device=DEV) * 0.03
    lib = native.lib()
    prev = lib.dvis_x3_set_reserve(0)
    try:
        a, c = Fn.x3_ffn_ln(x, l1, l2, norm), Fn.conv3x3_x3(xc, w, None, None, True, 1)
        for r in (8, 32, 200, 1000):
            assert lib.dvis_x3_set_reserve(r) in (0, 8, 32, 200)
            assert torch.equal(Fn.x3_ffn_ln(x, l1, l2, norm), a) and torch.equal(Fn.conv3x3_x3(xc, w, None, None, True, 1), c)
    finally:
        lib.dvis_x3_set_reserve(prev)


@pytest.mark.parametrize("N,B,S", [(288, 3, 1000), (256, 2, 129), (128, 1, 77), (768, 2, 515)])
def test_linear_with_the_position_added_in_the_kernel(N, B, S):
    """dvis_x3_linear_add: (x + pos) W^T + b with pos (S, K) shared by the B frames (`with_pos_embed(src, pos)`,
    msdeformattn.py:99-101) — bit-identical to the kernel fed the materialised fp32 sum, and against fp64."""
    from dvis_plus_amd import functions as Fn
    lin = _lin(256, N, N + S)
    x, pos = torch.randn(B, S, 256, device=DEV), torch.randn(1, S, 256, device=DEV)
    got = Fn.x3_linear(x, lin.weight, lin.bias, xadd=pos)
    assert got.shape == (B, S, N) and torch.equal(got, Fn.x3_linear(x + pos, lin.weight, lin.bias))
    xs = (x + pos).double()
    ref = xs @ lin.weight.double().t() + lin.bias.double()
    scale = xs.abs() @ lin.weight.double().abs().t() + lin.bias.double().abs()
    assert _rel(got, ref, scale) <= 4e-7


@pytest.mark.parametrize("K,N,M,act,with_res", [(1024, 1024, 3000, None, True), (1024, 4096, 777, "gelu", False),
                                                (4096, 1024, 1500, None, True), (256, 256, 301, "gelu", True),
                                                (1024, 3072, 513, "relu", True)])
def test_linear_with_gelu_and_residual_in_the_epilogue(K, N, M, act, with_res):
    """dvis_x3_linear_res — the ViT blocks' `x + proj(...)`, `fc2(gelu(fc1(x)))` (backbones_vitAdapter) without separate GELU /
    add passes: against fp64, not worse than the fp32 composition; and through Fn.linear(tall=True, act=, residual=)."""
    from dvis_plus_amd import functions as Fn
    lin = _lin(K, N, K + N)
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV) if with_res else None
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    f32 = F.linear(x, lin.weight, lin.bias)
    if act == "gelu":
        ref, f32 = F.gelu(ref), F.gelu(f32)
    elif act == "relu":
        ref, f32 = torch.relu(ref), torch.relu(f32)
    if with_res:
        ref, f32 = ref + res.double(), f32 + res
    out = Fn.x3_linear(x, lin.weight, lin.bias, act=act, residual=res)
    scale = (x.double().abs() @ lin.weight.double().abs().t() + 1.0)
    e_x3, e_f32 = _rel(out, ref, scale), _rel(f32, ref, scale)
    assert e_x3 <= max(1.5 * e_f32, 2e-7), (e_x3, e_f32)
    # Fn.linear(tall=True): the tiled kernel from K = 512 on (csrc/gemm_x3_tile.hip), the streaming kernel below
    via = Fn.linear(x, lin.weight, lin.bias, tall=True, act=act, residual=res)
    e_via = _rel(via, ref, scale)
    assert e_via <= max(1.5 * e_f32, 2e-7), (e_via, e_f32)
    if K < Fn.X3_TILE_MIN_K:
        assert torch.equal(via, out)
    # a 3-D input with a residual view; a row's bits do not depend on the rows around it
    if with_res and M % 3 == 0:
        o3 = Fn.linear(x.view(3, M // 3, K), lin.weight, lin.bias, tall=True, act=act, residual=res.view(3, M // 3, N))
        assert torch.equal(o3.view(M, N), via)
    part = Fn.linear(x[:100], lin.weight, lin.bias, tall=True, act=act, residual=None if res is None else res[:100])
    assert torch.equal(part, via[:100])


@pytest.mark.parametrize("K,N,M", [(1024, 256, 1), (512, 512, 255), (1024, 1024, 257), (2048, 256, 1000)])
def test_tiled_kernel_layout_is_exact_on_integer_operands_and_strided_rows(K, N, M):
    """csrc/gemm_x3_tile.hip: integer-valued operands come out bit for bit (every fragment / chunk / tile index is right), rows
    with a stride (a slice of a wider tensor), ragged M."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(K + N + M)
    w = torch.randint(-8, 9, (N, K), generator=g).float().to(DEV)
    b = torch.randint(-8, 9, (N,), generator=g).float().to(DEV)
    wide = torch.randint(-8, 9, (M, K + 64), generator=g).float().to(DEV)
    x = wide[:, 32:32 + K]
    assert Fn.x3_tile_ok(x, N, K)
    out = Fn.x3_tile_linear(x, w, b)
    ref = (x.double() @ w.double().t() + b.double()).float()
    assert torch.equal(out, ref)
    res = torch.randint(-8, 9, (M, N), generator=g).float().to(DEV)
    out = Fn.x3_tile_linear(x, w, b, act="relu", residual=res)
    assert torch.equal(out, torch.relu(ref) + res)
