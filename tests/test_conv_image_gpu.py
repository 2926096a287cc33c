"""GPU: operand images on either side of csrc/conv1x1_x3.hip (include/dvis_hip.h: dvis_conv_x3_image, dvis_upsample_add_image) —
the 64 .. 512-channel maps inside the R50's bottlenecks and the FPN output convolution's input as pre-split f16 fragments.

Pinned: the image layout both ways (integer operands bit for bit through producer and consumer: accumulator channel order,
chunks, groups, row ends that are not a multiple of 32 pixels, stride 2, zero padding of the nine taps), fp32-grade results
against fp64, equality with the fp32-map forms of the same kernels, a frame's bits independent of the batch, run-to-run bits,
and the top-down sum written as an image = the fp32 kernel's sum (to the 22 bits an image carries)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def decode(img):
    """OperandImage -> (N, C, H, W) fp32: hi + lo of every element, unscaled."""
    N, C, H, W = img.N, img.C, img.H, img.W
    XG, NCC = (W + 31) // 32, C // 64
    t = img.data.view(torch.float16).view(N, H, XG, NCC, 4, 2, 2, 32, 8).float()      # n y xg chunk S hl g x e
    v = (t[:, :, :, :, :, 0] + t[:, :, :, :, :, 1]) / (2.0 ** img.exp)               # n y xg chunk S g x e
    out = torch.zeros(N, C, H, XG * 32, device=img.data.device)
    for cc in range(NCC):
        for S in range(4):
            for g in range(2):
                for e in range(8):
                    c = 64 * cc + 32 * (S >> 1) + 16 * (S & 1) + 8 * (e >> 2) + 4 * g + (e & 3)
                    out[:, c] = v[:, :, :, cc, S, g, :, e].reshape(N, H, XG * 32)
    return out[:, :, :, :W]


def as_image(x):
    """An fp32 map as an operand image, through the product's own producer: top-down sum with a zero top map."""
    from dvis_plus_amd import functions as Fn
    N, C, H, W = x.shape
    return Fn.upsample_add_image(x.contiguous(), torch.zeros(N, C, max(1, H // 2), max(1, W // 2), device=x.device))


def _ints(shape, lo, hi, g, density=1.0):
    t = torch.randint(lo, hi, shape, generator=g).float()
    if density < 1.0:
        t = t * (torch.rand(shape, generator=g) < density).float()
    return t.to(DEV)


@pytest.mark.parametrize("C,K,H,W,stride,taps,N", [
    (256, 256, 9, 40, 1, 9, 2),          # the FPN output convolution's form; rows end inside a group
    (128, 128, 10, 64, 2, 9, 2),         # res3.0 conv2: stride 2
    (128, 128, 7, 33, 1, 9, 3),          # one pixel in the second group
    (512, 512, 5, 20, 1, 9, 1),          # two passes of 256 output channels, 8 chunks
    (128, 512, 6, 50, 1, 1, 2),          # conv3: 1x1 from an image, residual + ReLU below
    (256, 1024, 4, 17, 1, 1, 2),
])
def test_image_input_is_exact_on_integers_and_fp32_grade(C, K, H, W, stride, taps, N):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(C + K + H + W)
    k = 3 if taps == 9 else 1
    x = _ints((N, C, H, W), 0, 6, g, 0.5)
    w = _ints((K, C, k, k), -2, 3, g, 0.1)
    b = _ints((K,), -20, 20, g)
    OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
    r = _ints((N, K, OH, OW), -30, 30, g) if taps == 1 else None
    assert Fn.x3_images_ok(N, C, K, H, W, x.device, taps, stride)
    img = as_image(x)
    assert torch.equal(decode(img), x)
    want = F.conv2d(x.double(), w.double(), b.double(), stride, k // 2) + (r.double() if r is not None else 0)
    if taps == 1:
        want = want.clamp_min(0)
    got = Fn.conv_x3_image(img, w, b, r, relu=taps == 1, stride=stride)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    assert torch.equal(got, want.float())
    assert torch.equal(got, Fn.conv_x3_image(img, w, b, r, relu=taps == 1, stride=stride))
    # image out of an image in (conv2's form): the stored fragments decode to the same integers
    if taps == 9:
        out_img = Fn.conv_x3_image(img, w, b, None, relu=True, stride=stride, out_image=True)
        assert torch.equal(decode(out_img), want.clamp_min(0).float())
    # real operands: against fp64, next to the fp32 library convolution's error; equal to the fp32-map form of the same kernel
    # to the 22 bits the image carries
    xf = torch.randn(N, C, H, W, generator=g).relu().to(DEV)
    wf = (torch.randn(K, C, k, k, generator=g) * (2.0 / (C * taps)) ** 0.5).to(DEV)
    bf = torch.randn(K, generator=g).to(DEV)
    ref = F.conv2d(xf.double(), wf.double(), bf.double(), stride, k // 2)
    scale = F.conv2d(xf.double().abs(), wf.double().abs(), bf.double().abs(), stride, k // 2)
    lib = F.conv2d(xf, wf, bf, stride, k // 2)
    got = Fn.conv_x3_image(as_image(xf), wf, bf, None, stride=stride)
    e, e_lib = float(((got.double() - ref).abs() / scale).max()), float(((lib.double() - ref).abs() / scale).max())
    assert e <= max(1.5 * e_lib, 6e-7), (e, e_lib)
    # a frame alone = the frame in the batch
    assert torch.equal(got[:1], Fn.conv_x3_image(as_image(xf[:1].contiguous()), wf, bf, None, stride=stride))


@pytest.mark.parametrize("C,K,H,W,N", [(512, 128, 9, 40, 2), (1024, 256, 5, 33, 2), (256, 512, 4, 64, 1)])
def test_image_output_of_a_1x1_convolution(C, K, H, W, N):
    """conv1's form: fp32 map in, operand image out (+ ReLU)."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(C + K)
    x = _ints((N, C, H, W), 0, 4, g, 0.3)
    w = _ints((K, C, 1, 1), -2, 3, g, 0.05)
    b = _ints((K,), -10, 10, g)
    want = F.conv2d(x.double(), w.double(), b.double()).clamp_min(0).float()
    img = Fn.conv_x3_image(x, w, b, None, relu=True, out_image=True)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    assert (img.N, img.C, img.H, img.W) == (N, K, H, W)
    assert torch.equal(decode(img), want)
    xf = torch.randn(N, C, H, W, generator=g).relu().to(DEV)
    wf = (torch.randn(K, C, 1, 1, generator=g) * (2.0 / C) ** 0.5).to(DEV)
    ref = Fn.conv1x1_x3(xf, wf, None, None, True)
    got = decode(Fn.conv_x3_image(xf, wf, None, None, relu=True, out_image=True))
    assert float((got - ref).abs().max()) <= 2.0 ** -21 * float(ref.abs().max())


def test_a_bottleneck_through_operand_images():
    """conv1 (map -> image) -> conv2 (image -> image, 3x3) -> conv3 (image -> map, + shortcut, ReLU): the res3 - res5 form, against the
    same three layers through the fp32-map kernels and against fp64."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(11)
    N, C, M, H, W = 2, 512, 128, 23, 40
    x = torch.randn(N, C, H, W, generator=g).relu().to(DEV)
    w1 = (torch.randn(M, C, 1, 1, generator=g) * (2.0 / C) ** 0.5).to(DEV)
    w2 = (torch.randn(M, M, 3, 3, generator=g) * (2.0 / (9 * M)) ** 0.5).to(DEV)
    w3 = (torch.randn(C, M, 1, 1, generator=g) * (2.0 / M) ** 0.5).to(DEV)
    b1, b2, b3 = (torch.randn(n, generator=g).to(DEV) * 0.2 for n in (M, M, C))
    a1 = Fn.conv_x3_image(x, w1, b1, None, relu=True, out_image=True)
    a2 = Fn.conv_x3_image(a1, w2, b2, None, relu=True, out_image=True)
    y = Fn.conv_x3_image(a2, w3, b3, x, relu=True)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    r1 = Fn.conv1x1_x3(x, w1, b1, None, True)
    r2 = Fn.conv3x3_x3(r1, w2, b2, None, True)
    r3 = Fn.conv1x1_x3(r2, w3, b3, x, True)
    ref = (F.conv2d(F.relu(F.conv2d(F.relu(F.conv2d(x.double(), w1.double(), b1.double())), w2.double(), b2.double(), 1, 1)),
                    w3.double(), b3.double()) + x.double()).clamp_min(0)
    s = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * s
    assert float((y - r3).abs().max()) <= 2e-6 * s
    assert torch.equal(y, Fn.conv_x3_image(Fn.conv_x3_image(Fn.conv_x3_image(x, w1, b1, None, relu=True, out_image=True), w2, b2, None,
                                                            relu=True, out_image=True), w3, b3, x, relu=True))


@pytest.mark.parametrize("N,C,H,W", [(2, 256, 12, 40), (1, 64, 9, 33)])
def test_top_down_sum_as_an_image(N, C, H, W):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(H)
    lat = torch.randn(N, C, H, W, generator=g).to(DEV)
    top = torch.randn(N, C, H // 2, W // 2 if W % 2 == 0 else (W + 1) // 2, generator=g).to(DEV)
    aff = (torch.rand(N * C, generator=g).to(DEV) + 0.5, torch.randn(N * C, generator=g).to(DEV))
    want = Fn.upsample_add(lat, top, aff)
    got = decode(Fn.upsample_add_image(lat, top, aff))
    assert float((got - want).abs().max()) <= 2.0 ** -21 * float(want.abs().max())
    img = Fn.upsample_add_image(lat, top, aff)
    assert torch.equal(img.data, Fn.upsample_add_image(lat, top, aff).data)
