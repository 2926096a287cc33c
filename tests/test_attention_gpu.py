"""GPU parity: the fp32-MFMA attention kernel vs torch's nn.MultiheadAttention math in fp64 on the CPU
(the op the reference calls: F.multi_head_attention_forward, need_weights=True).  Tolerance 2e-5 abs on
O(1) outputs (BASELINE: 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_attention(q, k, v, nheads, mask=None):
    """(L, B, C) tensors -> (Lq, B, C), fp64, explicit softmax; mask (B, Lq, Lk) True = blocked."""
    Lq, B, C = q.shape
    Lk, d = k.shape[0], C // nheads
    qh = q.double().view(Lq, B, nheads, d).permute(1, 2, 0, 3)
    kh = k.double().view(Lk, B, nheads, d).permute(1, 2, 0, 3)
    vh = v.double().view(Lk, B, nheads, d).permute(1, 2, 0, 3)
    s = (qh / d ** 0.5) @ kh.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(mask[:, None].bool(), float("-inf"))
    o = torch.softmax(s, -1) @ vh
    return o.permute(2, 0, 1, 3).reshape(Lq, B, C)


CASES = [  # (Lq, Lk, B, heads, d, masked)
    (100, 920, 2, 8, 32, True), (100, 3680, 2, 8, 32, True), (100, 14720, 1, 8, 32, True),
    (100, 100, 3, 8, 64, False), (30, 30, 100, 8, 64, False), (100, 100, 30, 8, 64, False),
    (6, 24, 2, 2, 32, True), (6, 6, 1, 2, 32, False), (200, 333, 2, 8, 32, True), (17, 70, 2, 4, 64, True),
    (5, 5, 6, 2, 32, False), (130, 257, 1, 2, 64, True),
    # few (batch, head, query-tile) workgroups and Lk <= 128: the key-split latency kernel (tracker shapes)
    (100, 100, 1, 8, 64, True), (100, 100, 1, 8, 32, True), (100, 128, 1, 8, 64, True), (33, 113, 2, 4, 32, True),
    (100, 1, 1, 8, 64, False), (1, 100, 1, 8, 64, False),
    # key-partitioned kernel (d = 32, > 64 queries, >= 512 keys): two query chunks, ragged last key tile, 65 queries
    # (one live row in the fifth tile), no mask with Lk % 4 != 0, many (batch, head) pairs (one split per wave)
    (200, 1000, 2, 8, 32, True), (65, 516, 1, 4, 32, True), (100, 777, 2, 8, 32, False), (112, 2000, 1, 2, 32, False),
    (100, 640, 40, 8, 32, True), (113, 530, 3, 8, 32, True),
]


@pytest.mark.parametrize("Lq,Lk,B,H,d,masked", CASES)
def test_attention_vs_fp64(Lq, Lk, B, H, d, masked):
    from dvis_plus_amd.functions import attention
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    C = H * d
    q, k, v = (torch.randn(L, B, C, generator=g) for L in (Lq, Lk, Lk))
    q = q * 2.0
    mask = None
    if masked:
        mask = torch.rand(B, Lq, Lk, generator=g) < 0.7
        mask[:, :, 0] = False                      # every row keeps at least one key
        mask[0, 0, 1:] = True                      # a row with exactly one live key
    ref = ref_attention(q, k, v, H, mask)
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV), H, None if mask is None else mask.to(DEV)).cpu()
    torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-5)


def test_attention_strided_views_of_fused_projection():
    """q/k/v as views into one (L, B, 3C) in-projection output, output written into a given buffer."""
    from dvis_plus_amd.functions import attention
    g = torch.Generator().manual_seed(1)
    L, B, H, d = 100, 4, 8, 64
    C = H * d
    qkv = torch.randn(L, B, 3 * C, generator=g)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    ref = ref_attention(q, k, v, H)
    dq = qkv.to(DEV)
    buf = torch.zeros(L, B, C + 64, device=DEV)                  # output into a strided buffer as well
    out = attention(dq[..., :C], dq[..., C:2 * C], dq[..., 2 * C:], H, out=buf[..., :C])
    assert out.data_ptr() == buf.data_ptr()
    torch.testing.assert_close(out.cpu().double(), ref, rtol=0, atol=2e-5)
    assert torch.count_nonzero(buf[..., C:]) == 0


def test_fully_blocked_rows_follow_the_reference_reset():
    """dvis_Plus/video_mask2former_transformer_decoder.py:297: a row blocked everywhere attends to everything."""
    from dvis_plus_amd.functions import attention
    g = torch.Generator().manual_seed(2)
    Lq, Lk, B, H, d = 100, 920, 2, 8, 32
    q, k, v = (torch.randn(L, B, H * d, generator=g) for L in (Lq, Lk, Lk))
    mask = torch.rand(B, Lq, Lk, generator=g) < 0.5
    mask[0, 3] = True
    mask[1, 99] = True
    allowed = (~mask).sum(-1).int()
    fixed = mask.clone()
    fixed[torch.where(fixed.sum(-1) == fixed.shape[-1])] = False
    ref = ref_attention(q, k, v, H, fixed)
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV), H, mask.to(DEV), allowed.to(DEV)).cpu()
    torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-5)


def test_fully_blocked_rows_short_sequences():
    """Same reset on the key-split latency kernel (Lk <= 128, few workgroups) and on the batched short path."""
    from dvis_plus_amd.functions import attention
    for B in (1, 40):
        g = torch.Generator().manual_seed(3 + B)
        Lq, Lk, H, d = 100, 100, 8, 64
        q, k, v = (torch.randn(L, B, H * d, generator=g) for L in (Lq, Lk, Lk))
        mask = torch.rand(B, Lq, Lk, generator=g) < 0.5
        mask[0, 3] = True
        mask[B - 1, 99] = True
        mask[0, 17, :96] = True            # only the last key tile (one wave's share) is live
        allowed = (~mask).sum(-1).int()
        fixed = mask.clone()
        fixed[torch.where(fixed.sum(-1) == fixed.shape[-1])] = False
        ref = ref_attention(q, k, v, H, fixed)
        out = attention(q.to(DEV), k.to(DEV), v.to(DEV), H, mask.to(DEV), allowed.to(DEV)).cpu()
        torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-5)


def test_attention_matches_torch_mha_module():
    """End to end against nn.MultiheadAttention itself (fp32 CPU), projections done with torch on the GPU."""
    from dvis_plus_amd.functions import attention
    torch.manual_seed(0)
    C, H, Lq, Lk, B = 256, 8, 100, 920, 2
    mha = torch.nn.MultiheadAttention(C, H).eval()
    q, k, v = torch.randn(Lq, B, C), torch.randn(Lk, B, C), torch.randn(Lk, B, C)
    mask = torch.rand(B, Lq, Lk) < 0.6
    mask[:, :, 0] = False
    with torch.no_grad():
        ref = mha(q, k, v, attn_mask=mask.repeat_interleave(H, 0))[0]
        W, b = mha.in_proj_weight.to(DEV), mha.in_proj_bias.to(DEV)
        qp = F.linear(q.to(DEV), W[:C], b[:C])
        kp = F.linear(k.to(DEV), W[C:2 * C], b[C:2 * C])
        vp = F.linear(v.to(DEV), W[2 * C:], b[2 * C:])
        o = attention(qp, kp, vp, H, mask.to(DEV))
        out = F.linear(o, mha.out_proj.weight.to(DEV), mha.out_proj.bias.to(DEV)).cpu()
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=2e-5)


def test_online_softmax_rescale_branch_is_exercised():
    """A key far above the rest late in the sequence forces the running-max rescale path (and only there)."""
    from dvis_plus_amd.functions import attention
    g = torch.Generator().manual_seed(4)
    Lq, Lk, B, H, d = 100, 512, 1, 2, 32
    q, k, v = (torch.randn(L, B, H * d, generator=g) for L in (Lq, Lk, Lk))
    k[400] = q[7] * 3.0            # spikes q[7]'s score at key 400
    k[130] = q[50] * 2.0
    ref = ref_attention(q, k, v, H)
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV), H).cpu()
    torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-5)


def test_lazy_row_maximum_between_raises():
    """The key-partitioned kernel raises its softmax reference only when a score exceeds it by more than 2^16: scores that
    climb steadily by less than that per tile (probabilities up to 2^16 relative to the stale reference), then a jump far
    above it, then a fall — against fp64."""
    from dvis_plus_amd.functions import attention
    g = torch.Generator().manual_seed(9)
    Lq, Lk, B, H, d = 100, 1024, 1, 2, 32
    q, k, v = (torch.randn(L, B, H * d, generator=g) for L in (Lq, Lk, Lk))
    ramp = torch.linspace(0.0, 1.0, Lk).view(Lk, 1, 1)
    k = k * 0.2 + q[11:12] * ramp * 0.35        # query 11: score grows ~ linearly to ~ +11 (log2 units) over the keys
    k[900] = q[11] * 4.0                        # ... then one key far above (a raise), followed by ordinary keys
    k[37] = q[60] * 1.0
    ref = ref_attention(q, k, v, H)
    out = attention(q.to(DEV), k.to(DEV), v.to(DEV), H).cpu()
    torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("L,B,H", [(3681, 1, 16), (1100, 2, 4), (1024, 1, 2)])
def test_long_self_attention_on_split_f16_products_vs_fp64(L, B, H, monkeypatch):
    """attn_x3_kernel (head dim 64, no mask, >= 1024 tokens: the ViT blocks of config #5): against fp64 at the fp32 kernel's
    tolerance, next to the fp32 kernel's own error on the same operands; ragged last key tile / query chunk (3681, 1100);
    operands of mixed magnitude; a batch entry's bits do not depend on its batch mates; same bits run to run."""
    from dvis_plus_amd import functions as Fn
    if not Fn.X3:
        pytest.skip("DVIS_X3=0")
    g = torch.Generator().manual_seed(L + B)
    C = H * 64
    q, k, v = (torch.randn(L, B, C, generator=g) for _ in range(3))
    q = q * 2.0
    k[:, :, ::7] *= 8.0                       # a few large channels, as LayerNorm'd ViT activations have
    v[:, :, 3::11] *= 0.01
    ref = ref_attention(q, k, v, H)
    dq, dk, dv = q.to(DEV), k.to(DEV), v.to(DEV)
    seen = []
    lib = Fn.native.lib()
    orig = lib.dvis_attention_forward_k
    monkeypatch.setattr(lib, "dvis_attention_forward_k", lambda *a: (seen.append(a[-1]), orig(*a))[1])
    out = Fn.attention(dq, dk, dv, H)
    assert seen == [2], "the long d = 64 self-attention must take the split-f16 kernel"
    with Fn.x3_disabled():
        exact = Fn.attention(dq, dk, dv, H)
    assert seen[-1] == 0
    e_x3 = float((out.cpu().double() - ref).abs().max())
    e_f32 = float((exact.cpu().double() - ref).abs().max())
    assert e_x3 <= max(2.0 * e_f32, 2e-5), (e_x3, e_f32)
    assert torch.equal(out, Fn.attention(dq, dk, dv, H))
    if B > 1:
        one = Fn.attention(dq[:, :1].contiguous(), dk[:, :1].contiguous(), dv[:, :1].contiguous(), H)
        assert torch.equal(one, out[:, :1])
    Fn.X3_GUARD.check_now(dq.device)
    dk[5, 0, 3] = 1e4                         # beyond the split's range: the guard names it instead of NaN output going on
    Fn.attention(dq, dk, dv, H)
    with pytest.raises(Fn.X3RangeError):
        Fn.X3_GUARD.check_now(dq.device)


@pytest.mark.parametrize("B,L,heads", [(2, 1100, 8), (3, 1024, 8), (1, 1283, 16)])
def test_qkv_projection_writing_the_attention_operands_equals_projection_then_attention(B, L, heads):
    """dvis_x3_tile_linear_qkv + dvis_attention_x3_packed (the ViT blocks' `attn(qkv(x))`, backbones_vitAdapter) against fp64, and
    against the unfused product path (Fn.linear -> Fn.attention): batch entries that straddle the GEMM's row tiles, ragged L."""
    from dvis_plus_amd import functions as Fn
    if not Fn.X3:
        pytest.skip("DVIS_X3=0")
    C = heads * 64
    g = torch.Generator().manual_seed(B * L + heads)
    x = torch.randn(B, L, C, generator=g).to(DEV)
    w = (torch.randn(3 * C, C, generator=g) * (C ** -0.5)).to(DEV)
    b = (torch.randn(3 * C, generator=g) * 0.1).to(DEV)
    with torch.no_grad():
        assert Fn.x3_qkv_attention_ok(x, w, heads)
        out = Fn.x3_qkv_attention(x, w, b, heads)
        qkv = (x.double() @ w.double().t() + b.double()).view(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)      # (3, B, h, L, 64)
        p = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) / 8.0, -1)
        ref = (p @ qkv[2]).permute(0, 2, 1, 3).reshape(B, L, C)
        q32 = Fn.linear(x, w, b, tall=True).transpose(0, 1)
        unfused = torch.empty_like(out)
        Fn.attention(q32[..., :C], q32[..., C:2 * C], q32[..., 2 * C:], heads, out=unfused.transpose(0, 1))
    err, err_u = float((out.double() - ref).abs().max()), float((unfused.double() - ref).abs().max())
    assert err <= 2e-5 and err <= 3 * err_u + 1e-6, (err, err_u)
