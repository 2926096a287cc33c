"""dvis_gemm_ln (csrc/gemm_ln.hip): a projection with the LayerNorm seam(s) of the post-norm blocks in its A-operand
prologue — `x = LN2(LN1(a) + add)`, `C = act(x W^T + bias + res)`, `a_out = x` — against fp64 math for every tile
configuration; the stacked (per-batch bias) GEMM; and the referring tracker's chain built from them against its
layer-by-layer form (dvis_Plus/tracker.py:277-318) at the production sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():          # the kernels are inference-only (under autograd the front-ends refuse)
        yield


def _ln64(x, norm):
    if norm is None:
        return x
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), norm.weight.double(), norm.bias.double(), norm.eps)


def _norm(K, seed):
    g = torch.Generator().manual_seed(seed)
    n = torch.nn.LayerNorm(K).to(DEV)
    with torch.no_grad():
        n.weight.copy_(1 + 0.3 * torch.randn(K, generator=g))
        n.bias.copy_(0.2 * torch.randn(K, generator=g))
    return n


def _configs(K):
    from dvis_plus_amd import native
    lib = native.lib()
    ok = []
    for c in range(lib.dvis_gemm_ln_num_configs()):
        ok.append(c)
    return ok


@pytest.mark.parametrize("M,N,K", [(100, 512, 512), (100, 1536, 512), (100, 2048, 512), (200, 512, 512), (37, 132, 512),
                                   (100, 512, 256), (100, 192, 64), (6, 64, 64), (100, 512, 496), (300, 128, 128),
                                   (100, 512, 2048), (50, 36, 1040)])
@pytest.mark.parametrize("mode", ["ln1", "add_ln2", "ln1_add_ln2", "plain"])
def test_every_config_vs_fp64(M, N, K, mode):
    if K > 512 and mode != "plain":
        pytest.skip("the norm forms hold K <= 512 (a row's fragments stay in registers)")
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a = (2.0 * torch.randn(M, K, generator=g) + 0.5).to(DEV)
    add = torch.randn(M, K, generator=g).to(DEV) if "add" in mode else None
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    n1 = _norm(K, 1) if "ln1" in mode else None
    n2 = _norm(K, 2) if "ln2" in mode else None
    x = _ln64(a.double(), n1)
    if add is not None:
        x = x + add.double()
    x = _ln64(x, n2)
    want = torch.relu(x @ w.double().t() + bias.double() + res.double())
    scale = float((x.abs() @ w.double().abs().t()).max()) + 1.0
    tried = 0
    for cfg in [-1] + _configs(K):
        try:
            got, xn = Fn.gemm_ln(a, w, bias, norm1=n1, add=add, norm2=n2, relu=True, res=res, config=cfg)
        except RuntimeError as e:
            assert "holds K <=" in str(e), e          # a configuration whose K split is too shallow for this K / plain-only
            continue
        tried += 1
        if mode == "plain":
            assert xn is None          # nothing was normalised: no second output
        else:
            assert float((xn.double() - x).abs().max()) <= 2e-5 * max(1.0, float(x.abs().max())), f"config {cfg}: normalised rows"
        err = float((got.double() - want).abs().max())
        assert err <= 2e-6 * scale, f"config {cfg}: max err {err:.3e} (scale {scale:.1f})"
    assert tried >= 2


def test_strided_operands_outputs_in_place_and_determinism():
    """Row-sliced views (a fused in_proj output, a slot of a preallocated (T, Q, B, C) buffer) work without copies; a_out /
    out land where the caller says; two calls give the same bits."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(5)
    M, N, K = 100, 512, 512
    big = torch.randn(M, 3 * K, generator=g).to(DEV)
    a = big[:, K:2 * K]
    add = torch.randn(M, 2 * K, generator=g).to(DEV)[:, :K]
    wbig = torch.randn(3 * N, K, generator=g).to(DEV)
    w = wbig[N:2 * N]
    n1, n2 = _norm(K, 3), _norm(K, 4)
    outs = torch.zeros(3, M, N, device=DEV)
    xs = torch.zeros(3, M, K, device=DEV)
    got, xn = Fn.gemm_ln(a, w, None, norm1=n1, add=add, norm2=n2, a_out=xs[1], out=outs[2])
    assert got.data_ptr() == outs[2].data_ptr() and xn.data_ptr() == xs[1].data_ptr()
    assert not outs[:2].any() and not xs[0].any() and not xs[2].any()
    x = _ln64(_ln64(a.double(), n1) + add.double(), n2)
    assert float((xn.double() - x).abs().max()) < 2e-5
    assert float((got.double() - x @ w.double().t()).abs().max()) < 2e-4
    again, xa = Fn.gemm_ln(a, w, None, norm1=n1, add=add, norm2=n2)
    assert torch.equal(again, got) and torch.equal(xa, xn)


def test_rows_do_not_depend_on_their_neighbours():
    """A row's bits depend on (N, K, configuration) only: a clip's Q rows alone == the same rows stacked with another
    clip's (what tracker_batch > 1 relies on), the configuration pinned through gemm_sizes_as."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(9)
    K, N = 512, 1536
    a = torch.randn(200, K, generator=g).to(DEV)
    w = torch.randn(N, K, generator=g).to(DEV)
    n1 = _norm(K, 6)
    with Fn.gemm_sizes_as(rows=100):
        alone, xa = Fn.gemm_ln(a[:100].contiguous(), w, None, norm1=n1)
        both, xb = Fn.gemm_ln(a, w, None, norm1=n1)
    assert torch.equal(both[:100], alone) and torch.equal(xb[:100], xa)


def test_refused_operands_raise():
    from dvis_plus_amd import functions as Fn
    a = torch.randn(10, 520, device=DEV)
    w = torch.randn(64, 520, device=DEV)
    n = torch.nn.LayerNorm(520).to(DEV)
    with pytest.raises(RuntimeError, match="not served"):
        Fn.gemm_ln(a, w, norm1=n)                           # K > 512 with a norm
    with pytest.raises(RuntimeError, match="not served"):
        Fn.gemm_ln(a[:, :40], w[:, :40])                    # K % 16 != 0
    with pytest.raises(RuntimeError, match="forms are"):
        Fn.gemm_ln(a[:, :64], w[:, :64], add=a[:, :64])     # add without norm2
    with pytest.raises(RuntimeError, match="not served"):
        Fn.gemm_ln(a[:, :64].half(), w[:, :64].half())


@pytest.mark.parametrize("L,M,N,K", [(6, 100, 512, 512), (3, 37, 64, 64), (2, 200, 132, 128)])
def test_stacked_projections_vs_fp64(L, M, N, K):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(L + M + N + K)
    a = torch.randn(M, L * K, generator=g).to(DEV)
    w = torch.randn(L, N, K, generator=g).to(DEV)
    b = torch.randn(L, N, generator=g).to(DEV)
    got = Fn.gemm_nt_stacked(a, w, b)
    for l in range(L):
        want = a[:, l * K:(l + 1) * K].double() @ w[l].double().t() + b[l].double()
        scale = float((a[:, l * K:(l + 1) * K].double().abs() @ w[l].double().abs().t()).max())
        assert float((got[l].double() - want).abs().max()) <= 4e-7 * scale, l
    assert torch.equal(Fn.gemm_nt_stacked(a, w, b), got)


@pytest.mark.parametrize("B", [1, 2])
@pytest.mark.parametrize("resume", [False, True])
def test_tracker_fused_chain_vs_layer_by_layer(B, resume):
    """Production sizes (C = 512, 8 heads, 6 layers, Q = 100): the fused chain (hoisted cross-attentions, LayerNorms in
    the consuming GEMMs' prologues, 35 launches per frame) against the layer-by-layer form, same weights, same inputs.
    Not bit-equal (different summation orders of the LayerNorm statistics) — agreement to fp32 rounding over a 12-frame
    recurrence of 6 layers; the assignment indices are computed before the chain and are identical."""
    from dvis_plus_amd.tracker import ReferringTracker_noiser
    torch.manual_seed(0)
    trk = ReferringTracker_noiser(hidden_channel=512, feedforward_channel=2048, num_head=8, decoder_layer_num=6,
                                  mask_dim=256, class_num=124).eval().to(DEV)
    g = torch.Generator().manual_seed(1)
    T, Q = 12, 100
    fe_nn = torch.randn(B, 512, 2 * T, Q, generator=g).to(DEV)
    fe = torch.nn.functional.layer_norm(fe_nn.permute(0, 2, 3, 1), (512,)).permute(0, 3, 1, 2).contiguous()
    outs = {}
    with torch.no_grad():
        for fused in (True, False):
            trk.fused_chain = fused
            if resume and B == 1:
                trk(fe[:, :, :T], None, resume=False, frame_embeds_no_norm=fe_nn[:, :, :T], need_masks=False)
                o, idx = trk(fe[:, :, T:], None, resume=True, frame_embeds_no_norm=fe_nn[:, :, T:], need_masks=False,
                             return_indices=True)
            else:
                o, idx = trk(fe[:, :, :T], None, resume=False, frame_embeds_no_norm=fe_nn[:, :, :T], need_masks=False,
                             return_indices=True)
            outs[fused] = (o, idx, trk.last_outputs.clone(), trk.last_reference.clone())
    (a, ia, la, ra), (b, ib, lb, rb) = outs[True], outs[False]
    assert all((x == y).all() for x, y in zip(ia, ib))
    for k in ("pred_logits", "pred_embds", "pred_references"):
        ref = float(b[k].abs().max())
        err = float((a[k] - b[k]).abs().max())
        assert err <= 2e-4 * max(1.0, ref), f"{k}: {err:.3e} vs scale {ref:.2f}"
    assert float((la - lb).abs().max()) <= 2e-4 * max(1.0, float(lb.abs().max()))
    assert float((ra - rb).abs().max()) <= 2e-4 * max(1.0, float(rb.abs().max()))


def test_tracker_chain_is_36_launches_per_frame():
    """The claim of DESIGN.md section 3.10, counted at the C ABI with the hipGraph off: a resumed clip of T frames issues one
    K / V GEMM, T x (3 reference-MLP GEMMs + the q projection + the 48-head attention + the batched out-projection + 6 layers x
    [QKV with the LayerNorm seam in its prologue, self-attention, out-projection, linear1 with its seam, linear2]) = T x 36
    kernels, and ONE stand-alone LayerNorm (the last frame's output); the layer-by-layer form needs ~65 per frame."""
    from dvis_plus_amd import native
    from dvis_plus_amd.tracker import ReferringTracker_noiser
    torch.manual_seed(0)
    trk = ReferringTracker_noiser(hidden_channel=512, feedforward_channel=2048, num_head=8, decoder_layer_num=6,
                                  mask_dim=256, class_num=124).eval().to(DEV)
    trk.use_graphs = False
    g = torch.Generator().manual_seed(2)
    T, Q = 3, 100
    fe_nn = torch.randn(1, 512, 2 * T, Q, generator=g).to(DEV)
    fe = torch.nn.functional.layer_norm(fe_nn.permute(0, 2, 3, 1), (512,)).permute(0, 3, 1, 2).contiguous()
    lib = native.lib()
    names = ("dvis_gemm_ln", "dvis_gemm_nt", "dvis_gemm_nt_hm", "dvis_gemm_nt_bb", "dvis_attention_forward_k", "dvis_add_layernorm")
    orig = {n: getattr(lib, n) for n in names}
    counts = {}

    def counting(n):
        def call(*a):
            counts[n] = counts.get(n, 0) + 1
            return orig[n](*a)
        return call
    per_frame = {}
    try:
        for fused in (True, False):
            trk.fused_chain = fused
            trk(fe[:, :, :T], None, resume=False, frame_embeds_no_norm=fe_nn[:, :, :T], need_masks=False)
            rec = trk._recurrence
            inside = {}

            def spied(*a, **k):
                counts.clear()
                out = rec(*a, **k)
                inside.update(counts)
                return out
            trk._recurrence = spied
            for n in names:
                setattr(lib, n, counting(n))
            trk(fe[:, :, T:], None, resume=True, frame_embeds_no_norm=fe_nn[:, :, T:], need_masks=False)
            for n in names:
                setattr(lib, n, orig[n])
            trk._recurrence = rec
            per_frame[fused] = (sum(inside.values()) - 1 - (1 if fused else 0)) / T       # minus the K / V GEMM (and the final LN)
            if fused:
                assert inside == {"dvis_gemm_nt_hm": 1 + T, "dvis_gemm_ln": T * (3 + 6 * 4), "dvis_gemm_nt_bb": T,
                                  "dvis_attention_forward_k": T * 7, "dvis_add_layernorm": 1}, inside
    finally:
        for n in names:
            setattr(lib, n, orig[n])
    assert per_frame[True] == 36 and per_frame[False] >= 60, per_frame
