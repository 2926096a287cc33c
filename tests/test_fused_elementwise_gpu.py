"""GPU parity of the small fused passes (add+LayerNorm, upsample+add, GEMM+ReLU epilogue) vs the torch ops they replace."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,C", [(1000, 256), (100, 512), (3, 1024), (5, 32), (4097, 256)])
def test_add_layernorm(rows, C):
    from dvis_plus_amd.functions import add_layer_norm
    g = torch.Generator().manual_seed(rows + C)
    x, r = torch.randn(rows, C, generator=g) * 3, torch.randn(rows, C, generator=g)
    ln = torch.nn.LayerNorm(C)
    with torch.no_grad():
        ln.weight.normal_(1, 0.2, generator=g)
        ln.bias.normal_(0, 0.2, generator=g)
        ref = ln((x + r).double().float())
        ref64 = F.layer_norm((x.double() + r.double()), (C,), ln.weight.double(), ln.bias.double(), ln.eps)
        lnd = ln.to(DEV)
        out = add_layer_norm(x.to(DEV), r.to(DEV), lnd).cpu()
        out_nores = add_layer_norm(x.to(DEV), None, lnd).cpu()
    torch.testing.assert_close(out.double(), ref64, rtol=0, atol=5e-6)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(out_nores, ln.cpu()(x), rtol=1e-5, atol=5e-6)


def test_add_layernorm_3d_shapes_like_the_decoder():
    from dvis_plus_amd.functions import add_layer_norm
    x, r = torch.randn(100, 30, 256), torch.randn(100, 30, 256)
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ref = ln(x + r)
        out = add_layer_norm(x.to(DEV), r.to(DEV), ln.to(DEV)).cpu()
    assert out.shape == ref.shape
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=5e-6)


@pytest.mark.parametrize("N,C,h,w,H,W", [(2, 8, 5, 8, 10, 16), (1, 3, 7, 10, 23, 40), (2, 4, 46, 80, 92, 160),
                                           (2, 5, 5, 7, 10, 14), (3, 4, 3, 5, 7, 11)])     # W % 4 != 0: the scalar form
def test_upsample_add(N, C, h, w, H, W):
    from dvis_plus_amd.functions import upsample_add
    g = torch.Generator().manual_seed(h * w)
    lat, top = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, h, w, generator=g)
    with torch.no_grad():
        ref = lat + F.interpolate(top, size=(H, W), mode="bilinear", align_corners=False)
        out = upsample_add(lat.to(DEV), top.to(DEV)).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)


def test_linear_relu_epilogue():
    from dvis_plus_amd.functions import linear_relu
    lin = torch.nn.Linear(256, 1024)
    x = torch.randn(7, 100, 256)
    with torch.no_grad():
        ref = F.relu(lin(x))
        out = linear_relu(x.to(DEV), lin.to(DEV)).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 8, 12), (3, 256, 23, 40), (1, 7, 5, 4), (2, 64, 15, 27), (3, 5, 3, 7)])   # last two: HW % 4 != 0
def test_bias_act_inplace(N, C, H, W):
    from dvis_plus_amd.functions import bias_act_
    g = torch.Generator().manual_seed(C)
    x, r, b = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, H, W, generator=g), torch.randn(C, generator=g)
    for res, relu in ((None, True), (r, True), (r, False), (None, False)):
        ref = x + b.view(1, -1, 1, 1)
        if res is not None:
            ref = ref + res
        if relu:
            ref = ref.relu()
        xd = x.clone().to(DEV)
        with torch.no_grad():
            out = bias_act_(xd, b.to(DEV), None if res is None else res.to(DEV), relu)
        assert out.data_ptr() == xd.data_ptr()
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N,C,sizes", [(2, 256, [(5, 7), (9, 13), (17, 30)]), (3, 64, [(23, 40)]), (1, 100, [(3, 3), (64, 65)])])
def test_maps_to_tokens(N, C, sizes):
    """tiled transpose into the token matrix == flatten(2).transpose(1, 2) + cat (bit-exact: pure data movement)."""
    from dvis_plus_amd.functions import maps_to_tokens
    g = torch.Generator().manual_seed(C)
    maps = [torch.randn(N, C, h, w, generator=g).to("cuda:0") for h, w in sizes]
    want = torch.cat([m.flatten(2).transpose(1, 2) for m in maps], 1)
    assert torch.equal(maps_to_tokens(maps), want)


@pytest.mark.parametrize("N,C,H,W", [(2, 5, 8, 16), (1, 3, 46, 80), (2, 64, 92, 160)])
def test_bias_relu_maxpool_is_bit_identical_to_the_three_torch_ops(N, C, H, W):
    from dvis_plus_amd.functions import bias_relu_maxpool
    g = torch.Generator().manual_seed(H)
    x, b = torch.randn(N, C, H, W, generator=g), torch.randn(C, generator=g)
    with torch.no_grad():
        ref = F.max_pool2d(torch.relu(x + b.view(1, -1, 1, 1)), kernel_size=3, stride=2, padding=1)
        out = bias_relu_maxpool(x.to(DEV), b.to(DEV)).cpu()
        out_nob = bias_relu_maxpool(x.to(DEV)).cpu()
    assert torch.equal(out, ref)
    assert torch.equal(out_nob, F.max_pool2d(torch.relu(x), kernel_size=3, stride=2, padding=1))


@pytest.mark.parametrize("N,C,H,W,relu", [(2, 64, 6, 10, True), (3, 256, 23, 40, False), (1, 32, 92, 160, True),
                                              (2, 32, 15, 27, True), (3, 96, 5, 7, False)])      # groups not a multiple of 4 floats
def test_group_norm_affine_plus_apply_equals_torch_group_norm(N, C, H, W, relu):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * 3 + 5           # mean >> std: E[x^2] - E[x]^2 must not cancel
    gn = torch.nn.GroupNorm(32, C)
    with torch.no_grad():
        gn.weight.normal_(1, 0.2, generator=g)
        gn.bias.normal_(0, 0.2, generator=g)
        ref = gn(x)
        ref = torch.relu(ref) if relu else ref
        xd = x.to(DEV)
        scale, shift = Fn.group_norm_affine(xd, gn.to(DEV))
        out = Fn.scale_shift_act_(xd.clone(), scale, shift, relu=relu).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-5)


def test_upsample_add_with_lateral_group_norm():
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(3)
    lat, top = torch.randn(2, 64, 46, 80, generator=g) + 2, torch.randn(2, 64, 23, 40, generator=g)
    gn = torch.nn.GroupNorm(32, 64)
    with torch.no_grad():
        gn.weight.normal_(1, 0.2, generator=g)
        ref = gn(lat) + F.interpolate(top, size=(46, 80), mode="bilinear", align_corners=False)
        latd = lat.to(DEV)
        out = Fn.upsample_add(latd, top.to(DEV), Fn.group_norm_affine(latd, gn.to(DEV))).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-5)


def test_add_layernorm_second_output_with_position_embedding():
    from dvis_plus_amd.functions import add_layer_norm
    g = torch.Generator().manual_seed(11)
    x, r, pos = torch.randn(3, 50, 256, generator=g), torch.randn(3, 50, 256, generator=g), torch.randn(1, 50, 256, generator=g)
    ln = torch.nn.LayerNorm(256)
    with torch.no_grad():
        ref = ln(x + r)
        out, out_pos = add_layer_norm(x.to(DEV), r.to(DEV), ln.to(DEV), pos=pos.to(DEV))
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=5e-6)
    assert torch.equal(out_pos, out + pos.to(DEV))


def test_maps_to_tokens_with_group_norm_and_position_output():
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(5)
    maps = [torch.randn(2, 64, h, w, generator=g) + 1 for (h, w) in ((3, 5), (6, 10), (12, 20))]
    gns = [torch.nn.GroupNorm(32, 64) for _ in maps]
    S = sum(m.shape[2] * m.shape[3] for m in maps)
    pos = torch.randn(1, S, 64, generator=g)
    with torch.no_grad():
        for gn in gns:
            gn.weight.normal_(1, 0.3, generator=g)
            gn.bias.normal_(0, 0.3, generator=g)
        ref = torch.cat([gn(m).flatten(2).transpose(1, 2) for gn, m in zip(gns, maps)], 1)
        md = [m.to(DEV) for m in maps]
        aff = [Fn.group_norm_affine(m, gn.to(DEV)) for m, gn in zip(md, gns)]
        # level 0 (3 x 5: groups of 30 floats, not float4-sized) is served by the scalar form of the statistics kernel
        assert all(a is not None for a in aff)
        aff[1] = None                                              # mixed: level 1 normalised up front
        md[1] = gns[1](md[1])
        out, out_pos = Fn.maps_to_tokens(md, aff, pos=pos.to(DEV))
        plain = Fn.maps_to_tokens(md, aff)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=2e-5)
    assert torch.equal(out_pos, out + pos.to(DEV)) and torch.equal(plain, out)


def test_zero_size_batches_pass_through_every_glue_op():
    """A rank that holds no frame of a clip calls the ops with N = 0: no launch, correctly shaped empty outputs."""
    from dvis_plus_amd import functions as Fn
    z = lambda *s: torch.zeros(*s, device=DEV)
    with torch.no_grad():
        assert Fn.mask_logits(z(0, 10, 64), z(0, 64, 8, 12)).shape == (0, 10, 8, 12)
        m, a = Fn.attn_mask(z(0, 10, 64), z(0, 64, 8, 12), (4, 6))
        assert m.shape == (0, 10, 24) and a.shape == (0, 10)
        ln = torch.nn.LayerNorm(64).to(DEV)
        assert Fn.add_layer_norm(z(0, 5, 64), z(0, 5, 64), ln).shape == (0, 5, 64)
        assert Fn.bias_act_(z(0, 8, 4, 8), z(8)).shape == (0, 8, 4, 8)
        assert Fn.bias_relu_maxpool(z(0, 8, 4, 8), z(8)).shape == (0, 8, 2, 4)
        assert Fn.upsample_add(z(0, 8, 4, 8), z(0, 8, 2, 4)).shape == (0, 8, 4, 8)
        assert Fn.maps_to_tokens([z(0, 8, 2, 4), z(0, 8, 4, 8)]).shape == (0, 40, 8)
        gn = torch.nn.GroupNorm(4, 8).to(DEV)
        sc, sh = Fn.group_norm_affine(z(0, 8, 4, 8), gn)
        assert sc.numel() == 0 and sh.numel() == 0
        assert Fn.scale_shift_act_(z(0, 8, 4, 8), sc, sh, relu=True).shape == (0, 8, 4, 8)


@pytest.mark.parametrize("N,S,C,row0,h,w", [(2, 300, 256, 60, 12, 20), (1, 77, 96, 0, 7, 11), (3, 1000, 32, 500, 20, 25)])
def test_tokens_to_map_equals_the_strided_view(N, S, C, row0, h, w):
    """dvis_tokens_to_nchw: one level of the encoder's token matrix as a contiguous (N, C, h, w) map (msdeformattn.py:333-339)."""
    from dvis_plus_amd import functions as Fn
    tok = torch.randn(N, S, C, device="cuda")
    with torch.no_grad():
        got = Fn.tokens_to_map(tok, row0, h, w)
    assert got.is_contiguous() and torch.equal(got, tok[:, row0:row0 + h * w].transpose(1, 2).reshape(N, C, h, w))


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
@pytest.mark.parametrize("N,H,W,Hp,Wp", [(2, 37, 53, 64, 64), (3, 32, 64, 32, 64), (1, 45, 53, 45, 53), (2, 720, 1280, 736, 1280)])
def test_normalize_pad_has_the_bits_of_the_torch_expression(dtype, N, H, W, Hp, Wp):
    """dvis_normalize_pad: `(x - pixel_mean) / pixel_std` + zero padding to the size divisibility
    (dvis_Plus/meta_architecture.py:1310-1311) in one pass, with the same fp32 subtract / divide: torch.equal."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(H * W + N)
    x = torch.randint(0, 256, (N, 3, H, W), generator=g, dtype=torch.uint8)
    x = x.to(DEV) if dtype == torch.uint8 else (x.float() + torch.rand(N, 3, H, W, generator=g)).to(DEV)
    mean = torch.tensor([123.675, 116.280, 103.530], device=DEV).view(-1, 1, 1)
    std = torch.tensor([58.395, 57.120, 57.375], device=DEV).view(-1, 1, 1)
    with torch.no_grad():
        assert Fn.normalize_pad_ok(x, mean)
        got = Fn.normalize_pad(x, mean, std, Hp, Wp)
    ref = F.pad((x.float() - mean) / std, (0, Wp - W, 0, Hp - H))
    assert got.shape == ref.shape and torch.equal(got, ref)


def test_preprocess_of_the_meta_architecture_takes_the_fused_pass(monkeypatch):
    from dvis_plus_amd import functions as Fn
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    calls = []
    orig = Fn.normalize_pad
    monkeypatch.setattr(Fn, "normalize_pad", lambda *a: (calls.append(1), orig(*a))[1])
    m = build_dvis_plus_r50("offline", task="vps").to(DEV).eval()
    frames = [torch.randint(0, 256, (3, 50, 70), dtype=torch.uint8) for _ in range(2)]
    with torch.no_grad():
        x, size = m.preprocess(frames)
    ref = F.pad((torch.stack(frames).to(DEV).float() - m.pixel_mean) / m.pixel_std, (0, 96 - 70, 0, 64 - 50))
    assert calls and size == (50, 70) and torch.equal(x, ref)


@pytest.mark.parametrize("B,C,h8,w8,with_x1", [(2, 64, 6, 10, True), (1, 128, 4, 34, True), (3, 64, 2, 2, True), (2, 64, 8, 66, False)])
def test_adapter_res2_equals_transposed_conv_add_upsample_batchnorm(B, C, h8, w8, with_x1):
    """dvis_adapter_res2 + the transposed convolution as a GEMM (vit_adapter._res2_folded) against the reference's composition
    (adapter.py forward: c1 = up(c2) + c1; c1 += interpolate(x1, 4); norm1(c1)) in fp64."""
    import torch.nn.functional as F
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(B * 1000 + C + h8 + w8)
    up = torch.nn.ConvTranspose2d(C, C, 2, 2)
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        up.weight.copy_(torch.randn(C, C, 2, 2, generator=g) * 0.1)
        up.bias.copy_(torch.randn(C, generator=g))
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g))
        bn.running_mean.copy_(torch.randn(C, generator=g))
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    c2_tok = torch.randn(B, h8 * w8, C, generator=g)
    c1 = torch.randn(B, C, 2 * h8, 2 * w8, generator=g)
    x1_tok = torch.randn(B, (h8 // 2) * (w8 // 2), C, generator=g)
    with torch.no_grad():
        upd, bnd = up.double(), bn.double()
        ref = upd(c2_tok.double().transpose(1, 2).reshape(B, C, h8, w8)) + c1.double()
        if with_x1:
            ref = ref + F.interpolate(x1_tok.double().transpose(1, 2).reshape(B, C, h8 // 2, w8 // 2), scale_factor=4, mode="bilinear",
                                      align_corners=False)
        ref = bnd(ref)
        up, bn = up.float(), bn.float()
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = s * up.bias + bn.bias - bn.running_mean * s
        w_l = (up.weight * s.view(1, C, 1, 1)).permute(2, 3, 1, 0).reshape(4 * C, C).contiguous()
        gm = (c2_tok.reshape(-1, C).double() @ w_l.double().t()).float()
        out = Fn.adapter_res2(gm.to(DEV), c1.to(DEV), x1_tok.to(DEV) if with_x1 else None, s.to(DEV), shift.to(DEV), h8, w8)
    err = float((out.double().cpu() - ref).abs().max())
    assert err <= 2e-5 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("B,C,H,W,gelu", [(2, 64, 4, 6, True), (1, 256, 6, 10, True), (3, 8, 2, 2, False), (2, 260, 4, 14, True)])
def test_dwconv3x3_tokens_equals_the_transposed_grouped_convolution(B, C, H, W, gelu):
    """dvis_dwconv3x3_tokens on the three-level token tensor against adapter_modules.py's DWConv (+ ConvFFN's GELU): per level
    transpose to NCHW, grouped Conv2d, transpose back, concatenate."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(B + C + H + W)
    conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=True, groups=C)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(C, 1, 3, 3, generator=g) * 0.3)
        conv.bias.copy_(torch.randn(C, generator=g))
    levels = [(2 * H, 2 * W), (H, W), (H // 2, W // 2)]
    N = sum(h * w for h, w in levels)
    x = torch.randn(B, N, C, generator=g)
    outs, off = [], 0
    with torch.no_grad():
        cd = conv.double()
        for h, w in levels:
            m = x[:, off:off + h * w].double().transpose(1, 2).reshape(B, C, h, w)
            outs.append(cd(m).flatten(2).transpose(1, 2))
            off += h * w
        ref = torch.cat(outs, 1)
        if gelu:
            ref = F.gelu(ref)
        conv = conv.float()
        out = Fn.dwconv3x3_tokens(x.to(DEV), levels, conv.weight.to(DEV), conv.bias.to(DEV), gelu=gelu)
    assert float((out.double().cpu() - ref).abs().max()) <= 1e-5
