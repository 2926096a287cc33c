"""torch.autocast around the product's entry points (host logic, CPU): the launcher evaluates inside ``with autocast():``
(train_net_video.py:259).  forward() is an fp32 island: same outputs as the plain call; the context itself is left as found.
The GPU counterpart (real kernels, fp16 and bf16) is tests/test_autocast_gpu.py."""
import pytest
import torch


def test_no_autocast_context_and_decorator_restore_the_callers_state():
    from dvis_plus_amd import functions as Fn
    seen = []

    @Fn.fp32_island
    def f(x):
        seen.append((torch.is_autocast_enabled("cpu"), torch.is_autocast_enabled()))
        return x @ x
    x = torch.ones(4, 4)
    assert f(x).dtype == torch.float32 and seen[-1] == (False, False)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        assert (x @ x).dtype == torch.bfloat16
        assert f(x).dtype == torch.float32 and seen[-1] == (False, False)
        assert torch.is_autocast_enabled("cpu") and (x @ x).dtype == torch.bfloat16     # restored
        with Fn.no_autocast():
            assert (x @ x).dtype == torch.float32
        assert (x @ x).dtype == torch.bfloat16
    assert Fn.f32(x.to(torch.bfloat16)).dtype == torch.float32 and Fn.f32(x) is x and Fn.f32(None) is None
    idx = torch.arange(3)
    assert Fn.f32(idx) is idx                                                           # integer tensors pass untouched


@pytest.mark.parametrize("mode", ["offline", "online"])
def test_g10_forward_inside_cpu_autocast_equals_the_reference_forward(oracle_ops, mode):
    """The g10 comparison (the reference's own forward on its fp32 CPU path) with the product called inside an autocast
    region: bit-equal panoptic maps, as in the plain call (tests/test_host_modules.py)."""
    import g10_model as G
    from test_host_modules import _g10_check_vps
    m, g, cfg, frames = G.build(mode, "vps")
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        out = m([G.video(frames, cfg)])
        assert torch.is_autocast_enabled("cpu")
    _g10_check_vps(out, g.outs, "off_vps" if mode == "offline" else "on_vps")


def test_stream_inside_cpu_autocast_keeps_the_consumers_context(oracle_ops):
    import g10_model as G
    m, g, cfg, frames = G.build("offline", "vps")
    with torch.no_grad():
        want = m([G.video(frames, cfg)])
        with torch.autocast("cpu", dtype=torch.bfloat16):
            outs = []
            for out in m.stream([G.video(frames, cfg), G.video(frames, cfg)]):
                assert torch.is_autocast_enabled("cpu")          # between two clips the consumer's context is in force
                outs.append(out)
    for out in outs:
        assert torch.equal(out["pred_masks"], want["pred_masks"]) and out["segments_infos"] == want["segments_infos"]


def test_linear_residual_promotes_like_the_reference_under_autocast():
    """Fn.linear(..., residual=r) outside the fused kernels must return ``r + y`` in r's dtype under autocast (the
    reference's ``x = x + ls1(attn(norm1(x)))`` promotes the half-precision branch to the fp32 residual stream) — not add the
    residual into the half-precision GEMM output in place."""
    from dvis_plus_amd import functions as Fn
    from dvis_plus_amd.vit_adapter import Block
    torch.manual_seed(0)
    x, w, b = torch.randn(5, 7, 16), torch.randn(16, 16), torch.randn(16)
    with torch.no_grad():
        plain = Fn.linear(x, w, b, residual=x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            y = Fn.linear(x, w, b, residual=x)
            g = Fn.linear(x, w, b, act="gelu", residual=x)
        assert plain.dtype == torch.float32 and y.dtype == torch.float32 and g.dtype == torch.float32
        torch.testing.assert_close(y, plain, rtol=2e-2, atol=2e-1)     # the branch is bf16, the stream fp32
        # fp32, no autocast, no autograd: still the in-place add (no extra pass)
        assert torch.equal(plain, x + torch.nn.functional.linear(x, w, b))


def test_vit_block_alone_under_autocast_keeps_an_fp32_residual_stream(oracle_ops):
    from dvis_plus_amd.vit_adapter import Block
    torch.manual_seed(0)
    blk = Block(32, 2, mlp_ratio=4, qkv_bias=True, init_values=0.5).eval()
    x = torch.randn(2, 9, 32)
    with torch.no_grad():
        ref = blk(x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = blk(x)
    assert out.dtype == torch.float32
    torch.testing.assert_close(out, ref, rtol=5e-2, atol=5e-2)
