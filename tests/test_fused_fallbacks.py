"""The glue ops of dvis_plus_amd.functions on CPU tensors: their torch formulations (used for CPU tensors, autograd and
odd shapes — never for the named hot ops) must mean what the fused kernels are tested against on the GPU."""
import torch
import torch.nn.functional as F

from dvis_plus_amd import functions as Fn


def test_group_norm_affine_has_no_cpu_path_and_callers_fall_back():
    x = torch.randn(2, 64, 6, 10)
    assert Fn.group_norm_affine(x, torch.nn.GroupNorm(32, 64)) is None


def test_maps_to_tokens_cpu_formulation():
    g = torch.Generator().manual_seed(0)
    maps = [torch.randn(2, 8, h, w, generator=g) for (h, w) in ((2, 3), (4, 6))]
    want = torch.cat([m.flatten(2).transpose(1, 2) for m in maps], 1)
    assert torch.equal(Fn.maps_to_tokens(maps), want)
    pos = torch.randn(1, want.shape[1], 8, generator=g)
    scale, shift = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g)
    out, out_pos = Fn.maps_to_tokens(maps, [None, (scale, shift)], pos=pos)
    m1 = maps[1] * scale.view(2, 8, 1, 1) + shift.view(2, 8, 1, 1)
    want = torch.cat([maps[0].flatten(2).transpose(1, 2), m1.flatten(2).transpose(1, 2)], 1)
    assert torch.equal(out, want) and torch.equal(out_pos, want + pos)


def test_add_layer_norm_cpu_formulation_with_position_output():
    g = torch.Generator().manual_seed(1)
    x, r, pos = torch.randn(2, 5, 16, generator=g), torch.randn(2, 5, 16, generator=g), torch.randn(1, 5, 16, generator=g)
    ln = torch.nn.LayerNorm(16)
    with torch.no_grad():
        out, out_pos = Fn.add_layer_norm(x, r, ln, pos=pos)
        assert torch.equal(out, ln(x + r)) and torch.equal(out_pos, out + pos)
        assert torch.equal(Fn.add_layer_norm(x, None, ln), ln(x))


def test_bias_relu_maxpool_and_upsample_add_cpu_formulations():
    g = torch.Generator().manual_seed(2)
    x, b = torch.randn(2, 4, 8, 16, generator=g), torch.randn(4, generator=g)
    want = F.max_pool2d(torch.relu(x + b.view(1, -1, 1, 1)), kernel_size=3, stride=2, padding=1)
    assert torch.equal(Fn.bias_relu_maxpool(x, b), want)
    lat, top = torch.randn(2, 4, 8, 16, generator=g), torch.randn(2, 4, 4, 8, generator=g)
    up = F.interpolate(top, size=(8, 16), mode="bilinear", align_corners=False)
    assert torch.equal(Fn.upsample_add(lat, top), lat + up)
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g)
    want = lat * scale.view(2, 4, 1, 1) + shift.view(2, 4, 1, 1) + up
    torch.testing.assert_close(Fn.upsample_add(lat, top, (scale, shift)), want, rtol=0, atol=0)


def test_encoder_layer_query_plumbing_matches_plain_layers(monkeypatch):
    """MSDeformAttnTransformerEncoder: handing layer i+1 the query written by layer i's last add+LayerNorm gives the
    same result as every layer forming src + pos itself (CPU: torch formulations of the ops, oracle MSDA)."""
    import conftest as c
    monkeypatch.setattr(Fn, "msda_fused_forward", c._o_msda_fused)
    monkeypatch.setattr(Fn, "MSDeformAttnFunction", c._OMSDAFunction)
    from dvis_plus_amd.pixel_decoder import MSDeformAttnTransformerEncoderOnly
    torch.manual_seed(3)
    enc = MSDeformAttnTransformerEncoderOnly(d_model=32, nhead=2, num_encoder_layers=3, dim_feedforward=64,
                                             num_feature_levels=3, enc_n_points=4).eval()
    srcs = [torch.randn(2, 32, h, w) for (h, w) in ((2, 3), (4, 6), (8, 12))]
    pos = [torch.randn(1, 32, h, w) for (h, w) in ((2, 3), (4, 6), (8, 12))]
    with torch.no_grad():
        got = enc(srcs, pos)[0]
        # reference: the plain recurrence, each layer given src and pos only
        e = enc.encoder
        shapes_py = [(2, 3), (4, 6), (8, 12)]
        src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        lvl_pos = torch.cat([p.flatten(2).transpose(1, 2) + enc.level_embed[l].view(1, 1, -1) for l, p in enumerate(pos)], 1)
        ss, lsi = enc._shape_tensors(shapes_py, src.device)
        ref_pts = e.reference_points_unpadded(shapes_py, src.device)
        out = src
        for layer in e.layers:
            out = layer(out, lvl_pos, ref_pts, ss, lsi, None, shapes_py=shapes_py)
    torch.testing.assert_close(got, out, rtol=1e-6, atol=1e-6)
