"""GPU parity: the fused 1x1 convolution (csrc/conv1x1.hip: contraction + FrozenBN shift + shortcut + ReLU in one pass)
against F.conv2d + bias + residual + relu evaluated in fp64 on the CPU — the op sequence of detectron2's BottleneckBlock
with the BN folded (backbone.py).  Tolerance: 2e-6 * sqrt(K) * max|.| (fp32 accumulation of K products)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def exact_fp32_dispatch(monkeypatch):
    """this file tests csrc/conv1x1.hip: with the split-f16 kernels on (the default) conv1x1_bias_act hands every layer with
    >= 64 input channels to csrc/conv1x1_x3.hip first (tests/test_gemm_x3_gpu.py)"""
    from dvis_plus_amd import functions as Fn
    monkeypatch.setattr(Fn, "X3", False)

# (N, K, M, H, W): the three kernel forms (<= 64, <= 128, passes of 256 rows), ragged pixel groups (H*W not a multiple of
# 64 / 32 / 512), row counts that are not a multiple of 16, several row passes, K that is not a multiple of 32
CASES = [(2, 256, 64, 24, 40), (1, 64, 64, 8, 12), (3, 64, 256, 8, 36), (2, 128, 512, 6, 10), (1, 256, 128, 16, 20),
         (2, 36, 20, 4, 5), (1, 100, 132, 2, 6), (2, 512, 64, 4, 8), (1, 64, 256, 92, 160)]


@pytest.mark.parametrize("N,K,M,H,W", CASES)
@pytest.mark.parametrize("with_res,relu", [(True, True), (False, True), (False, False)])
def test_fused_conv1x1_vs_fp64(N, K, M, H, W, with_res, relu):
    from dvis_plus_amd import native
    from dvis_plus_amd.functions import conv1x1_bias_act
    assert native.lib().dvis_conv1x1_supported(K, M, H * W) == 1
    g = torch.Generator().manual_seed(N * 1000 + K + M)
    x = torch.randn(N, K, H, W, generator=g)
    w = torch.randn(M, K, 1, 1, generator=g) / K ** 0.5
    b = torch.randn(M, generator=g)
    res = torch.randn(N, M, H, W, generator=g) if with_res else None
    ref = F.conv2d(x.double(), w.double(), b.double())
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = ref.relu()
    with torch.no_grad():
        out = conv1x1_bias_act(x.to(DEV), w.to(DEV), b.to(DEV), None if res is None else res.to(DEV), relu).cpu()
    assert out.shape == ref.shape
    torch.testing.assert_close(out.double(), ref, rtol=0, atol=2e-6 * K ** 0.5 * float(ref.abs().max() + 1))


def test_rows_are_not_mixed_up():
    """one non-zero weight per output channel: exact, catches row / k-permutation / pixel-order mistakes"""
    from dvis_plus_amd.functions import conv1x1_bias_act
    N, K, M, H, W = 2, 64, 256, 4, 24
    w = torch.zeros(M, K, 1, 1)
    for m in range(M):
        w[m, (m * 5 + 3) % K, 0, 0] = 1.0 + m
    x = torch.arange(N * K * H * W, dtype=torch.float32).reshape(N, K, H, W) / 64.0
    b = torch.arange(M, dtype=torch.float32)
    with torch.no_grad():
        out = conv1x1_bias_act(x.to(DEV), w.to(DEV), b.to(DEV), None, False).cpu()
    assert torch.equal(out, F.conv2d(x, w, b))


def test_unsupported_shapes_take_the_library_path(monkeypatch):
    from dvis_plus_amd import native
    from dvis_plus_amd.functions import conv1x1_bias_act
    monkeypatch.setenv("DVIS_STRICT", "0")
    assert native.lib().dvis_conv1x1_supported(1024, 256, 3680) == 0      # deep layers stay with the library
    assert native.lib().dvis_conv1x1_supported(256, 1024, 3680) == 0
    g = torch.Generator().manual_seed(5)
    x, w, b = torch.randn(1, 512, 4, 8, generator=g), torch.randn(256, 512, 1, 1, generator=g) / 22, torch.randn(256, generator=g)
    with torch.no_grad():
        out = conv1x1_bias_act(x.to(DEV), w.to(DEV), b.to(DEV), None, True).cpu()
    torch.testing.assert_close(out, F.conv2d(x, w, b).relu(), rtol=1e-4, atol=1e-4)


def test_repeated_launches_agree():
    """Regression: with the row offset in the SCALAR offset field of the 16-byte stores, dword 0 of lanes 12-15 of a row was
    stored with the next row's value in most — not all — launches of this shape (an unpadded store-data hazard, DESIGN.md
    section 3.6b).  50 launches, every one compared."""
    from dvis_plus_amd.functions import conv1x1_bias_act, mask_logits
    N, K, M, H, W = 2, 256, 64, 24, 40
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, K, H, W, generator=g).to(DEV)
    w = (torch.randn(M, K, 1, 1, generator=g) / 16).to(DEV)
    b = torch.randn(M, generator=g).to(DEV)
    with torch.no_grad():
        ref = F.conv2d(x.double(), w.double(), b.double()).float()
        for it in range(50):
            out = conv1x1_bias_act(x, w, b, None, False)
            bad = int(((out - ref).abs() > 1e-3).sum())
            assert bad == 0, f"launch {it}: {bad} elements off"
        # the mask contraction stores 16-byte pieces the same way
        e, f = torch.randn(2, 100, 256, device=DEV), torch.randn(2, 256, 24, 40, device=DEV)
        want = torch.einsum("bqc,bchw->bqhw", e.double(), f.double()).float()
        for it in range(20):
            assert int(((mask_logits(e, f) - want).abs() > 1e-3).sum()) == 0
