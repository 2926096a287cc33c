"""dvis_gemm_nt (csrc/gemm.hip) — the deterministic exact-fp32 GEMM behind every projection of the tracker / refiner
(nn.MultiheadAttention in / out projections, FFN, MLP: dvis_Plus/tracker.py:293-318, refiner.py:104-139; the refiner's
nn.Conv1d pair as im2col GEMMs, refiner.py:42-54; the cosine matrices of noiser.py:43-56) — against fp64 math, for every
tile configuration, ragged sizes, strided operands, the epilogue variants and the batched form; bit-reproducibility."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(a, w, bias, res, relu):
    y = a.double() @ w.double().transpose(-1, -2)
    if bias is not None:
        y = y + bias.double()
    if res is not None:
        y = y + res.double()
    return torch.relu(y) if relu else y


def _n_configs():
    from dvis_plus_amd import native
    return native.lib().dvis_gemm_num_configs()


SHAPES = [(100, 512, 512), (100, 2048, 512), (100, 512, 2048), (3000, 1536, 512), (3000, 512, 2560), (200, 1536, 1024),
          (3000, 125, 1024), (3000, 1, 512), (7, 40, 36), (33, 17, 20), (1, 16, 4), (130, 72, 132)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_every_config_vs_fp64(M, N, K):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(DEV)
    w = torch.randn(N, K, generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    want = _ref(a, w, bias, None, False)
    scale = float((a.double().abs() @ w.double().abs().t()).max())
    for cfg in [-1] + list(range(_n_configs())):
        got = Fn.gemm_nt(a, w, bias, config=cfg)
        err = float((got.double() - want).abs().max())
        assert err <= 4e-7 * scale, f"config {cfg}: max err {err:.3e} (scale {scale:.1f})"


def test_layout_is_not_transposed():
    """Asymmetric exact case: A = rows of a permutation-like one-hot matrix -> C must be the selected rows of W^T exactly."""
    from dvis_plus_amd import functions as Fn
    M, N, K = 45, 70, 64
    w = torch.arange(N * K, dtype=torch.float32).view(N, K).to(DEV)
    sel = torch.randint(0, K, (M,), generator=torch.Generator().manual_seed(0))
    a = torch.zeros(M, K)
    a[torch.arange(M), sel] = 1.0
    for cfg in range(_n_configs()):
        got = Fn.gemm_nt(a.to(DEV), w, config=cfg).cpu()
        assert torch.equal(got, w.cpu()[:, sel].t()), f"config {cfg}"


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("with_res", [False, True])
def test_epilogue_and_strided_operands(relu, with_res):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(5)
    big = torch.randn(300, 3 * 128, generator=g).to(DEV)          # a = a column slice of a fused projection (row stride 384)
    a = big[:, 128:256]
    wfull = torch.randn(3 * 96, 128, generator=g).to(DEV)         # w = a row slice of an in_proj_weight
    w = wfull[96:192]
    bias = torch.randn(3 * 96, generator=g).to(DEV)[96:192]
    res = torch.randn(300, 96, generator=g).to(DEV) if with_res else None
    got = Fn.gemm_nt(a, w, bias, relu=relu, res=res)
    want = _ref(a, w, bias, res, relu)
    assert float((got.double() - want).abs().max()) < 2e-4
    # odd N: scalar store path (N = 125 class logits), 3-D input
    x = torch.randn(30, 100, 1024, generator=g).to(DEV)
    w2 = torch.randn(125, 1024, generator=g).to(DEV)
    b2 = torch.randn(125, generator=g).to(DEV)
    got = Fn.gemm_nt(x, w2, b2, relu=relu)
    assert got.shape == (30, 100, 125)
    assert float((got.double() - _ref(x, w2, b2, None, relu)).abs().max()) < 1e-3


def test_batched_and_linear_front_end():
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(9)
    a = torch.randn(30, 100, 512, generator=g).to(DEV)
    b = torch.randn(30, 100, 512, generator=g).to(DEV)
    got = Fn.bmm_nt(a, b)
    assert float((got.double() - a.double() @ b.double().transpose(1, 2)).abs().max()) < 2e-4
    lin = torch.nn.Linear(512, 2048).to(DEV)
    with torch.no_grad():
        y = Fn.linear(a, lin.weight, lin.bias, relu=True, own=True)
        assert float((y - torch.relu(lin(a))).abs().max()) < 1e-3
    # autograd keeps the torch path (the kernel is inference-only)
    y = Fn.linear(a, lin.weight, lin.bias, own=True)
    assert y.requires_grad


def test_refused_operands_raise():
    from dvis_plus_amd import functions as Fn
    a = torch.randn(8, 6, device=DEV)            # K % 4 != 0
    w = torch.randn(5, 6, device=DEV)
    with pytest.raises(RuntimeError):
        Fn.gemm_nt(a, w)
    with pytest.raises(RuntimeError):
        Fn.gemm_nt(a.cpu(), w.cpu())


def test_bit_reproducible_and_independent_of_neighbours():
    """Same arguments -> same bits (no atomics, fixed split), also while another stream keeps the GPU busy; and a row's
    result does not depend on which other rows are in the call when the configuration is pinned."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(11)
    a = torch.randn(3000, 512, generator=g).to(DEV)
    w = torch.randn(2048, 512, generator=g).to(DEV)
    first = Fn.gemm_nt(a, w)
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=DEV)
    for _ in range(5):
        with torch.cuda.stream(side):
            big @ big
        assert torch.equal(Fn.gemm_nt(a, w), first)
    torch.cuda.synchronize()
    part = Fn.gemm_nt(a[:100], w, config=3)
    assert torch.equal(part, Fn.gemm_nt(a, w, config=3)[:100])


@pytest.mark.parametrize("M,N,K", [(70001, 200, 256), (20000, 125, 1024)])
def test_tall_problems_persistent_tiles_vs_fp64(M, N, K):
    """A larger than the caches (M K 4 B > 64 MB): column-fastest XCD-aware tile order, and for the one-wave configurations
    the PERSISTENT kernel (a wave walks several tiles, the next tile's fragments requested before the current epilogue);
    ragged M and N, bias + ReLU epilogue, every configuration, checked on row samples incl. the first / last tiles."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(DEV)
    w = torch.randn(N, K, generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 300), torch.arange(M - 300, M),
                      torch.randint(0, M, (600,), generator=g)]).to(DEV)
    want = torch.relu(a[rows].double() @ w.double().t() + bias.double())
    scale = float((a[rows].double().abs() @ w.double().abs().t()).max())
    ref = None
    for cfg in [-1] + list(range(_n_configs())):
        got = Fn.gemm_nt(a, w, bias, relu=True, config=cfg)
        err = float((got[rows].double() - want).abs().max())
        assert err <= 4e-7 * scale, f"config {cfg}: max err {err:.3e}"
        assert torch.isfinite(got).all()
        if cfg == -1:
            ref = got
    assert torch.equal(Fn.gemm_nt(a, w, bias, relu=True), ref)          # and the same bits again


def test_head_major_output():
    """dvis_gemm_nt_hm: C written as (N / d, M, d) — the head-major value layout of MSDeformAttn — equals the row-major
    result re-laid, bit for bit (same tiles, same summation order; only the store address differs), small and tall."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(3)
    for M in (300, 70001):
        a = torch.randn(M, 256, generator=g).to(DEV)
        w = torch.randn(256, 256, generator=g).to(DEV)
        b = torch.randn(256, generator=g).to(DEV)
        plain = Fn.gemm_nt(a, w, b)
        hm = Fn.gemm_nt(a, w, b, head_major=32)
        assert hm.shape == (8, M, 32)
        assert torch.equal(hm, plain.view(M, 8, 32).permute(1, 0, 2))
