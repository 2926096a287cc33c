"""GPU: size-independent properties of the hot-path kernels at BASELINE.json's full sizes (720p frame: 184x320 mask map,
14720 keys at the finest level, 100 / 200 queries) — linearity, permutation invariance, optimality — where an fp64 CPU
oracle would take too long.  Complements the oracle / golden parity tests at small sizes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_attention_full_size_linear_in_v_and_key_permutation_invariant():
    from dvis_plus_amd.functions import attention
    g = torch.Generator(device=DEV).manual_seed(0)
    Lq, Lk, B, H, d = 100, 14720, 2, 8, 32
    C = H * d
    q = torch.randn(Lq, B, C, device=DEV, generator=g)
    k = torch.randn(Lk, B, C, device=DEV, generator=g)
    v1 = torch.randn(Lk, B, C, device=DEV, generator=g)
    v2 = torch.randn(Lk, B, C, device=DEV, generator=g)
    mask = torch.rand(B, Lq, Lk, device=DEV, generator=g) < 0.6
    mask[:, :, 0] = False
    o1, o2 = attention(q, k, v1, H, mask), attention(q, k, v2, H, mask)
    o12 = attention(q, k, 0.7 * v1 - 1.3 * v2, H, mask)
    torch.testing.assert_close(o12, 0.7 * o1 - 1.3 * o2, rtol=1e-4, atol=2e-5)          # linear in V
    perm = torch.randperm(Lk, device=DEV, generator=g)
    op = attention(q, k[perm].contiguous(), v1[perm].contiguous(), H, mask[:, :, perm].contiguous())
    torch.testing.assert_close(op, o1, rtol=1e-4, atol=2e-5)                            # keys are a set
    # rows of probabilities sum to 1: with V = ones the output is ones
    ones = attention(q, k, torch.ones_like(v1), H, mask)
    torch.testing.assert_close(ones, torch.ones_like(ones), rtol=0, atol=1e-5)


def test_vit_sized_attention_matches_torch_sdpa():
    """3681 tokens x 16 heads x 64 (ViT-Adapter-L at 720p), strided views of a fused qkv projection."""
    from dvis_plus_amd.functions import attention
    g = torch.Generator(device=DEV).manual_seed(1)
    B, N, H, d = 2, 3681, 16, 64
    C = H * d
    qkv = torch.randn(B, N, 3 * C, device=DEV, generator=g)
    v = qkv.transpose(0, 1)
    out = torch.empty(B, N, C, device=DEV)
    attention(v[..., :C], v[..., C:2 * C], v[..., 2 * C:], H, out=out.transpose(0, 1))
    q4 = qkv.view(B, N, 3, H, d).permute(2, 0, 3, 1, 4).double()
    ref = F.scaled_dot_product_attention(q4[0], q4[1], q4[2]).transpose(1, 2).reshape(B, N, C)
    torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-5)


def test_mask_logits_full_size_linear_and_matches_einsum():
    from dvis_plus_amd.functions import mask_logits
    g = torch.Generator(device=DEV).manual_seed(2)
    B, Q, C, H, W = 2, 200, 256, 184, 320
    e1 = torch.randn(B, Q, C, device=DEV, generator=g)
    e2 = torch.randn(B, Q, C, device=DEV, generator=g)
    f = torch.randn(B, C, H, W, device=DEV, generator=g)
    a, b = mask_logits(e1, f), mask_logits(e2, f)
    torch.testing.assert_close(mask_logits(2.0 * e1 - 0.5 * e2, f), 2.0 * a - 0.5 * b, rtol=1e-4, atol=1e-3)
    ref = torch.einsum("bqc,bchw->bqhw", e1.double(), f.double())
    torch.testing.assert_close(a.double(), ref, rtol=0, atol=2e-4)                      # |logit| ~ 16, fp32 chain of 256


def test_attn_mask_full_size_equals_interpolated_threshold():
    from dvis_plus_amd.functions import attn_mask, mask_logits
    g = torch.Generator(device=DEV).manual_seed(3)
    B, Q, C, H, W = 2, 100, 256, 184, 320
    e = torch.randn(B, Q, C, device=DEV, generator=g)
    f = torch.randn(B, C, H, W, device=DEV, generator=g)
    full = mask_logits(e, f).double()
    for (h, w) in ((23, 40), (46, 80), (92, 160)):
        mask, allowed = attn_mask(e, f, (h, w))
        small = F.interpolate(full, size=(h, w), mode="bilinear", align_corners=False).flatten(2)
        want = small.sigmoid() < 0.5
        sure = small.abs() > 1e-3                                                       # away from the threshold
        got = mask.view(B, Q, h * w).bool()
        assert torch.equal(got[sure], want[sure])
        assert (~sure).float().mean().item() < 1e-3
        assert torch.equal(allowed.long(), (~got).sum(-1))


def test_lsap_200_queries_is_optimal_permutation():
    import ctypes
    from scipy.optimize import linear_sum_assignment
    from dvis_plus_amd import native
    rng = np.random.default_rng(0)
    lib = native.lib()
    for n in (100, 200):
        for _ in range(3):
            c = rng.random((n, n)).astype(np.float32).astype(np.float64)
            col = np.empty(n, dtype=np.int64)
            rc = lib.dvis_lsap_solve(c.ctypes.data_as(ctypes.c_void_p), n, n, col.ctypes.data_as(ctypes.c_void_p))
            assert rc == 0 and sorted(col.tolist()) == list(range(n))
            r, s = linear_sum_assignment(c)
            assert np.array_equal(col, s)                                               # same optimum, same tie-breaks
            assert c[np.arange(n), col].sum() <= c[np.arange(n), rng.permutation(n)].sum()


def test_msda_720p_weights_summing_to_one_reproduce_constant_maps():
    """A constant value map is a fixed point of bilinear sampling + convex weights wherever all corners are inside."""
    from dvis_plus_amd.functions import ms_deform_attn_forward
    shapes = torch.tensor([(23, 40), (46, 80), (92, 160)], device=DEV)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S, M, D, L, P = int(shapes.prod(1).sum()), 8, 32, 3, 4
    g = torch.Generator(device=DEV).manual_seed(4)
    value = torch.full((1, S, M, D), 3.25, device=DEV)
    loc = 0.1 + 0.8 * torch.rand(1, S, M, L, P, 2, device=DEV, generator=g)           # strictly inside every map
    w = torch.softmax(torch.randn(1, S, M, L * P, device=DEV, generator=g), -1).view(1, S, M, L, P)
    out = ms_deform_attn_forward(value, shapes, lsi, loc, w, 64)
    torch.testing.assert_close(out, torch.full_like(out, 3.25), rtol=0, atol=2e-6)
