"""GPU parity: mask-logit contraction / fused attention-mask kernels vs an fp64 oracle.

The oracle is the reference's own op sequence (einsum -> F.interpolate(bilinear, align_corners=False) ->
sigmoid < 0.5, dvis_Plus/video_mask2former_transformer_decoder.py:363-371) evaluated in fp64 on the CPU.
Logits: <= 1e-4 abs (BASELINE: 1e-3).  Mask bits: must equal the fp64 decision wherever the down-sized logit is
not within 1e-4 of the threshold (closer than that, fp32 summation order decides — in the reference too).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(B, Q, C, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, Q, C, generator=g), torch.randn(B, C, H, W, generator=g)


@pytest.mark.parametrize("B,Q,C,H,W", [(2, 100, 256, 24, 40), (1, 6, 16, 16, 24), (3, 200, 256, 8, 136),
                                        (1, 100, 256, 184, 320), (2, 17, 40, 5, 7), (1, 129, 64, 3, 50)])
def test_mask_logits_vs_fp64(B, Q, C, H, W):
    from dvis_plus_amd.functions import mask_logits
    e, f = _inputs(B, Q, C, H, W, seed=B * 100 + Q)
    ref = torch.einsum("bqc,bchw->bqhw", e.double(), f.double())
    out = mask_logits(e.to(DEV), f.to(DEV)).cpu()
    scale = (C ** 0.5)
    torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-6 * scale * 4)


def test_mask_logits_transpose_detecting():
    """asymmetric operands: identity-like embed picks single channels -> catches row/col or k-permutation mix-ups."""
    from dvis_plus_amd.functions import mask_logits
    B, Q, C, H, W = 1, 100, 256, 4, 48
    e = torch.zeros(B, Q, C)
    for q in range(Q):
        e[0, q, (q * 7 + 3) % C] = 1.0 + q
    f = torch.arange(C * H * W, dtype=torch.float32).reshape(1, C, H, W) / 100.0
    out = mask_logits(e.to(DEV), f.to(DEV)).cpu()
    ref = torch.einsum("bqc,bchw->bqhw", e, f)
    assert torch.equal(out, ref)     # one non-zero product per output: exact


@pytest.mark.parametrize("B,Q,C,H,W,s", [(2, 100, 256, 16, 32, 2), (2, 100, 256, 16, 32, 4), (1, 100, 256, 16, 32, 8),
                                          (1, 6, 16, 16, 24, 2), (1, 6, 16, 16, 24, 8), (2, 200, 256, 24, 40, 4),
                                          (1, 100, 256, 184, 320, 2), (1, 100, 256, 184, 320, 8)])
def test_attn_mask_vs_fp64(B, Q, C, H, W, s):
    from dvis_plus_amd.functions import attn_mask
    e, f = _inputs(B, Q, C, H, W, seed=7 + s)
    h, w = H // s, W // s
    logits = torch.einsum("bqc,bchw->bqhw", e.double(), f.double())
    small = F.interpolate(logits, size=(h, w), mode="bilinear", align_corners=False).flatten(2)
    ref = small.sigmoid() < 0.5
    mask, allowed = attn_mask(e.to(DEV), f.to(DEV), (h, w))
    mask, allowed = mask.cpu().bool(), allowed.cpu()
    decided = small.abs() > 1e-4
    assert torch.equal(mask[decided], ref[decided])
    assert (~decided).float().mean() < 1e-3
    assert torch.equal(allowed.long(), (~mask).sum(-1))


def test_attn_mask_fully_blocked_rows_are_reported():
    from dvis_plus_amd.functions import attn_mask
    B, Q, C, H, W = 2, 100, 256, 16, 32
    e, f = _inputs(B, Q, C, H, W, seed=3)
    f = f.abs() + 0.1
    e = e.abs()
    e[0, 5] = -e[0, 5]            # every logit of row (0, 5) negative -> blocked everywhere
    e[1, 99] = -e[1, 99]
    mask, allowed = attn_mask(e.to(DEV), f.to(DEV), (8, 16))
    allowed = allowed.cpu()
    assert allowed[0, 5] == 0 and allowed[1, 99] == 0
    assert (allowed > 0).sum() == B * Q - 2
    assert mask.cpu()[0, 5].all()


def test_mask_gemm_errors():
    from dvis_plus_amd.functions import attn_mask, mask_logits
    e, f = _inputs(1, 4, 8, 6, 6, 0)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        mask_logits(e.to(DEV), f)                   # mixed devices: never a silent copy or a torch formulation
    with pytest.raises(RuntimeError, match="GPU tensor"):
        mask_logits(e, f.to(DEV))
    # all-CPU tensors take the torch formulation of cpu_ops.py (BASELINE config #1), chosen by device alone
    torch.testing.assert_close(mask_logits(e, f), torch.einsum("bqc,bchw->bqhw", e, f))
    with pytest.raises(RuntimeError, match="even integer"):
        attn_mask(e.to(DEV), f.to(DEV), (2, 2))     # factor 3


@pytest.mark.parametrize("N,C,H,W", [(2, 256, 16, 32), (1, 7, 8, 8), (3, 64, 24, 40), (1, 256, 184, 320)])
def test_center_pool3_has_the_bits_of_the_torch_expression(N, C, H, W):
    """p_s = 0.25 * ((f_a + f_b) + (f_c + f_d)) over the four centre pixels of every s x s block: the taps and weights of
    F.interpolate(bilinear, align_corners=False) to 1 / s, added in the order dvis_attn_mask adds the four logits."""
    from dvis_plus_amd.functions import center_pool3
    f = torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(H + W)).to(DEV)
    with torch.no_grad():
        p8, p4, p2 = center_pool3(f)
    for s, p in ((8, p8), (4, p4), (2, p2)):
        o = s // 2 - 1
        a, b = f[:, :, o::s, o::s], f[:, :, o::s, o + 1::s]
        c, d = f[:, :, o + 1::s, o::s], f[:, :, o + 1::s, o + 1::s]
        assert torch.equal(p, ((a + b) + (c + d)) * 0.25), s
        # ... which is what the bilinear resize samples (fp64 check of the taps)
        ref = F.interpolate(f.double(), size=(H // s, W // s), mode="bilinear", align_corners=False)
        assert float((p.double() - ref).abs().max()) < 1e-6


@pytest.mark.parametrize("B,Q,C,H,W", [(2, 100, 256, 16, 32), (1, 6, 16, 16, 24), (2, 200, 256, 24, 40), (1, 100, 256, 184, 320),
                                        (3, 17, 40, 8, 8)])
def test_attn_mask_pooled_vs_fp64_and_vs_the_unpooled_kernel(B, Q, C, H, W):
    """The pyramid form against the reference's op sequence in fp64 (einsum -> interpolate -> sigmoid < 0.5): equal wherever the
    down-sized logit is not within 1e-4 of the threshold; the row counts are those of the emitted mask; against dvis_attn_mask
    (contract-then-average) only such near-threshold bits may differ."""
    from dvis_plus_amd.functions import attn_mask, attn_mask_pooled, center_pool3
    e, f = _inputs(B, Q, C, H, W, seed=11 + B)
    logits = torch.einsum("bqc,bchw->bqhw", e.double(), f.double())
    with torch.no_grad():
        pyr = center_pool3(f.to(DEV))
        assert pyr is not None
        for s, p in zip((8, 4, 2), pyr):
            h, w = H // s, W // s
            small = F.interpolate(logits, size=(h, w), mode="bilinear", align_corners=False).flatten(2)
            ref = small.sigmoid() < 0.5
            mask, allowed = attn_mask_pooled(e.to(DEV), p)
            old, old_allowed = attn_mask(e.to(DEV), f.to(DEV), (h, w))
            mask, allowed, old = mask.cpu().bool(), allowed.cpu(), old.cpu().bool()
            decided = small.abs() > 1e-4
            assert torch.equal(mask[decided], ref[decided]), s
            assert torch.equal(allowed.long(), (~mask).sum(-1)), s
            assert not (mask != old)[decided].any(), s


def test_attn_mask_pooled_fully_blocked_rows_and_odd_sizes():
    from dvis_plus_amd.functions import attn_mask_pooled, center_pool3
    B, Q, C = 2, 100, 256
    e, f = _inputs(B, Q, C, 16, 40, seed=3)
    f, e = f.abs() + 0.1, e.abs()
    e[0, 5], e[1, 99] = -e[0, 5], -e[1, 99]
    with torch.no_grad():
        p8, p4, p2 = center_pool3(f.to(DEV))                       # p8 is 2 x 5 = 10 pixels: not a multiple of 4 (byte stores)
        for p in (p8, p4, p2):
            mask, allowed = attn_mask_pooled(e.to(DEV), p)
            allowed = allowed.cpu()
            assert allowed[0, 5] == 0 and allowed[1, 99] == 0 and (allowed > 0).sum() == B * Q - 2
            assert mask.cpu()[0, 5].all() and not mask.cpu()[0, 6].any()
        assert center_pool3(torch.randn(1, 8, 12, 20, device=DEV)) is None        # H % 8 != 0: the caller keeps dvis_attn_mask
