"""dvis_conv3x3_winograd (csrc/winograd_conv.hip) against the fp64 convolution — the 3x3 convolutions of the path: the pixel
decoder's FPN output convolution (msdeformattn.py:262-270, :343-349) and conv2 of the R50 bottlenecks."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _count_kernel_calls():
    """Counts dvis_conv3x3_winograd launches from here on (the wrapper must not fall back to the library quietly)."""
    from dvis_plus_amd import native
    lib = native.lib()
    real, n = lib.dvis_conv3x3_winograd, [0]

    class Spy:
        def __call__(self, *a):
            n[0] += 1
            return real(*a)
    lib.dvis_conv3x3_winograd = Spy()

    def done():
        lib.dvis_conv3x3_winograd = real
        return n[0]
    return done


def _case(N, C, K, H, W, bias, relu, seed=0, scale=1.0):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(N, C, H, W, device="cuda", generator=g) * scale
    w = torch.randn(K, C, 3, 3, device="cuda", generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, device="cuda", generator=g) if bias else None
    calls = _count_kernel_calls()
    got = Fn.conv3x3_bias_act(x, w, b, relu, winograd=True)
    assert calls() == 1, "the own kernel did not run"
    want = F.conv2d(x.double(), w.double(), None if b is None else b.double(), 1, 1)
    want = want.relu() if relu else want
    lib = F.conv2d(x, w, b, 1, 1)
    lib = lib.relu() if relu else lib
    mag = float(F.conv2d(x.double().abs(), w.double().abs(), None, 1, 1).max())     # accumulated magnitude
    return float((got.double() - want).abs().max()), float((lib.double() - want).abs().max()), mag, got


@pytest.mark.parametrize("N,C,K,H,W", [
    (2, 64, 64, 32, 48),          # one channel block, even sizes
    (3, 64, 128, 23, 40),         # odd height (stride-32 map of a 736 x 1280 frame): half-used last tile row
    (2, 128, 64, 17, 34),         # odd height, 17 tiles per row: the two tiles of a lane straddle tile rows (scalar stores)
    (5, 16, 64, 16, 18),          # 72 tiles per image: workgroups straddle two images; a single pair of stages
    (1, 256, 256, 46, 80),        # the stride-16 map
    (1, 512, 512, 15, 20),        # stride 32 of a 480 x 640 frame: 80 tiles, the second workgroup is mostly past the end
    (1, 256, 256, 120, 160),      # FPN output convolution of a 480 x 640 frame
])
@pytest.mark.parametrize("bias,relu", [(False, False), (True, True)])
def test_winograd_equals_fp64_convolution(N, C, K, H, W, bias, relu):
    from dvis_plus_amd import native
    assert native.lib().dvis_conv3x3_winograd_supported(C, K, H, W)
    err, lib_err, mag, _ = _case(N, C, K, H, W, bias, relu)
    # F(2x2, 3x3) in fp32: a few ulps of the accumulated magnitude (the direct library kernel sits at ~1 ulp of it)
    assert err <= 8 * 2.0 ** -24 * mag, (err, lib_err, mag)


def test_winograd_fpn_output_convolution_full_size():
    """256 -> 256 at 184 x 320 (stride 4 of a 720p frame), 4 frames: error against fp64 next to the library's."""
    err, lib_err, mag, _ = _case(4, 256, 256, 184, 320, False, False, seed=3)
    assert err <= 8 * 2.0 ** -24 * mag, (err, lib_err, mag)
    print(f"winograd max err {err:.3e}, library {lib_err:.3e}, accumulated magnitude {mag:.1f}")


def test_winograd_is_bit_reproducible_and_batch_independent():
    """Fixed accumulation order: the same bits run to run, and a frame's result does not depend on its batch mates."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(6, 128, 46, 80, device="cuda", generator=g)
    w = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03
    a = Fn.conv3x3_bias_act(x, w, None, False, winograd=True)
    b = Fn.conv3x3_bias_act(x, w, None, False, winograd=True)
    assert torch.equal(a, b)
    c = Fn.conv3x3_bias_act(x[2:5].contiguous(), w, None, False, winograd=True)
    assert torch.equal(a[2:5], c)


def test_winograd_weights_follow_the_parameter():
    """The packed weights are cached per parameter and refreshed when it changes (in place: captured graphs)."""
    from dvis_plus_amd import functions as Fn
    x = torch.randn(1, 64, 16, 16, device="cuda")
    w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device="cuda") * 0.05)
    with torch.no_grad():
        a = Fn.conv3x3_bias_act(x, w, None, False, winograd=True)
        w.mul_(2.0)
        b = Fn.conv3x3_bias_act(x, w, None, False, winograd=True)
    assert torch.allclose(b, 2 * a, rtol=1e-5, atol=1e-6)


def test_winograd_refuses_shapes_it_does_not_serve():
    from dvis_plus_amd import native
    lib = native.lib()
    assert not lib.dvis_conv3x3_winograd_supported(3, 64, 64, 64)        # stem: 3 input channels
    assert not lib.dvis_conv3x3_winograd_supported(64, 32, 64, 64)       # 32 output channels
    assert not lib.dvis_conv3x3_winograd_supported(64, 64, 8, 8)         # 16 tiles per image
    assert not lib.dvis_conv3x3_winograd_supported(64, 64, 64, 63)       # odd width (16-byte patch rows)
    x = torch.zeros(1, 3, 64, 64, device="cuda")
    uf = torch.zeros(16 * 64 * 3, device="cuda")
    y = torch.zeros(1, 64, 64, 64, device="cuda")
    rc = lib.dvis_conv3x3_winograd(native.dev_ptr(x, "x"), native.dev_ptr(uf, "uf"), None, native.dev_ptr(y, "y"), 1, 3, 64, 64, 64,
                                   0, native.stream_ptr(x.device))
    assert rc != 0 and "unsupported shape" in lib.dvis_last_error().decode()


# ---- the stride-2 sibling (csrc/conv3x3s2.hip): conv2 of the first bottleneck of res3 / res4 / res5
@pytest.mark.parametrize("N,C,K,H,W", [
    (2, 64, 64, 32, 48),          # even sizes, one channel block
    (3, 128, 128, 23, 40),        # odd height: the last output row reads one row past the image (zero)
    (2, 16, 64, 30, 34),          # 17 output pixels per row: a lane's two pixels straddle output rows; a single stage pair
    (1, 256, 256, 46, 80),
    (5, 32, 64, 16, 18),          # 72 pixels per image: workgroups straddle two images
])
@pytest.mark.parametrize("bias,relu", [(False, False), (True, True)])
def test_conv3x3_stride2_equals_fp64_convolution(N, C, K, H, W, bias, relu):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(N, C, H, W, device="cuda", generator=g)
    w = torch.randn(K, C, 3, 3, device="cuda", generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(K, device="cuda", generator=g) if bias else None
    got = Fn.conv3x3s2_bias_act(x, w, b, relu, own=True)
    want = F.conv2d(x.double(), w.double(), None if b is None else b.double(), 2, 1)
    want = want.relu() if relu else want
    assert got.shape == want.shape
    mag = float(F.conv2d(x.double().abs(), w.double().abs(), None, 2, 1).max())
    err = float((got.double() - want).abs().max())
    assert err <= 4 * 2.0 ** -24 * mag, (err, mag)        # a direct fp32 contraction: ~1 ulp of the accumulated magnitude
    assert torch.equal(got, Fn.conv3x3s2_bias_act(x, w, b, relu, own=True))          # fixed accumulation order
    if N > 1:
        assert torch.equal(got[1:2], Fn.conv3x3s2_bias_act(x[1:2].contiguous(), w, b, relu, own=True))


def test_conv3x3_stride2_res3_shape_and_refusals():
    from dvis_plus_amd import functions as Fn, native
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(4, 128, 184, 320, device="cuda", generator=g)
    w = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03
    got = Fn.conv3x3s2_bias_act(x, w, None, False, own=True)
    want = F.conv2d(x.double(), w.double(), None, 2, 1)
    mag = float(F.conv2d(x.double().abs(), w.double().abs(), None, 2, 1).max())
    assert float((got.double() - want).abs().max()) <= 4 * 2.0 ** -24 * mag
    lib = native.lib()
    assert not lib.dvis_conv3x3s2_supported(3, 64, 64, 64) and not lib.dvis_conv3x3s2_supported(64, 64, 64, 63)
    with pytest.raises(RuntimeError, match="not served"):
        Fn.conv3x3s2_bias_act(torch.zeros(1, 3, 64, 64, device="cuda"), torch.zeros(64, 3, 3, 3, device="cuda"), own=True)


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 96, 160), (3, 16, 32), (2, 736, 1280)])
@pytest.mark.parametrize("bias,relu", [(False, False), (True, True)])
def test_stem_conv7x7_stride2_equals_fp64_convolution(N, H, W, bias, relu):
    """csrc/conv7x7s2.hip (detectron2 BasicStem.conv1): all four borders (3 columns / rows of padding), workgroups that straddle
    images, the 720p frame."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn(N, 3, H, W, device="cuda", generator=g)
    w = torch.randn(64, 3, 7, 7, device="cuda", generator=g) * 0.1
    b = torch.randn(64, device="cuda", generator=g) if bias else None
    got = Fn.conv7x7s2_stem(x, w, b, relu, own=True)
    want = F.conv2d(x.double(), w.double(), None if b is None else b.double(), 2, 3)
    want = want.relu() if relu else want
    assert got.shape == want.shape
    mag = float(F.conv2d(x.double().abs(), w.double().abs(), None, 2, 3).max())
    err = float((got.double() - want).abs().max())
    assert err <= 4 * 2.0 ** -24 * mag, (err, mag)
    assert torch.equal(got, Fn.conv7x7s2_stem(x, w, b, relu, own=True))
    with pytest.raises(RuntimeError, match="not served"):
        Fn.conv7x7s2_stem(torch.zeros(1, 3, 33, 64, device="cuda"), w, own=True)


@pytest.mark.parametrize("N,C,K,H,W", [
    (2, 128, 64, 8, 8),           # one stage pair, 64 pixels per image
    (3, 256, 128, 23, 40),        # workgroups straddle images (920 pixels each)
    (2, 1024, 256, 46, 80),       # conv1 of a res4 bottleneck
    (1, 512, 2048, 23, 40),       # conv3 of a res5 bottleneck (with the shortcut)
])
@pytest.mark.parametrize("bias,res,relu", [(False, False, False), (True, True, True)])
def test_conv1x1_mfma_equals_fp64(N, C, K, H, W, bias, res, relu):
    """csrc/conv1x1_mfma.hip: the compute-bound 1x1 layers + folded-BN shift + shortcut + ReLU in one kernel."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randn(N, C, H, W, device="cuda", generator=g)
    w = torch.randn(K, C, 1, 1, device="cuda", generator=g) * (1.0 / C) ** 0.5
    b = torch.randn(K, device="cuda", generator=g) if bias else None
    r = torch.randn(N, K, H, W, device="cuda", generator=g) if res else None
    got = Fn.conv1x1_mfma(x, w, b, r, relu)
    want = F.conv2d(x.double(), w.double(), None if b is None else b.double())
    want = want if r is None else want + r.double()
    want = want.relu() if relu else want
    mag = float(F.conv2d(x.double().abs(), w.double().abs()).max())
    assert float((got.double() - want).abs().max()) <= 4 * 2.0 ** -24 * mag
    assert torch.equal(got, Fn.conv1x1_mfma(x, w, b, r, relu))
    # the dispatcher picks it for these shapes (conv1x1.hip does not serve them), and refuses what it cannot do
    from dvis_plus_amd import native
    assert native.lib().dvis_conv1x1_mfma_supported(C, K, H * W) and not native.lib().dvis_conv1x1_mfma_supported(64, 64, H * W)
    x3 = Fn.X3
    with torch.no_grad():
        Fn.X3 = False                     # the exact-fp32 dispatch
        try:
            assert torch.equal(Fn.conv1x1_bias_act(x, w, b, r, relu), got) or native.lib().dvis_conv1x1_supported(C, K, H * W)
        finally:
            Fn.X3 = x3
        if x3:                            # the default dispatch (split-f16 matrix-core kernel where it serves the shape): same bound
            assert float((Fn.conv1x1_bias_act(x, w, b, r, relu).double() - want).abs().max()) <= 4 * 2.0 ** -24 * mag


@pytest.mark.parametrize("N,C,K,H,W", [(2, 256, 512, 46, 80), (3, 128, 64, 17, 22), (1, 1024, 2048, 46, 80)])
def test_conv1x1_stride2_mfma_equals_fp64(N, C, K, H, W):
    """The down-sampling 1x1 shortcut (stride 2) read in place by csrc/conv1x1_mfma.hip."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(N, C, H, W, device="cuda", generator=g)
    w = torch.randn(K, C, 1, 1, device="cuda", generator=g) * (1.0 / C) ** 0.5
    b = torch.randn(K, device="cuda", generator=g)
    with torch.no_grad():
        if not Fn.conv1x1s2_supported(x, w):
            pytest.skip("odd output pixel count")
        got = Fn.conv1x1_mfma(x, w, b, None, False, stride=2)
    want = F.conv2d(x.double(), w.double(), b.double(), 2)
    assert got.shape == want.shape
    mag = float(F.conv2d(x.double().abs(), w.double().abs(), None, 2).max())
    assert float((got.double() - want).abs().max()) <= 4 * 2.0 ** -24 * mag
