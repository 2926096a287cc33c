"""GPU: row images on either side of csrc/gemm_x3_tile.hip (include/dvis_hip.h, "ROW IMAGES") — the ViT blocks' GEMM inputs
(block.py:36-104 of the DINOv2 backbone: norm1 -> qkv -> attention -> proj, norm2 -> fc1 -> GELU -> fc2) written once, pre-split,
by the layer that produces them.

Pinned: the layout of every producer (integer rows come back bit for bit through producer and consumer, every k order, ragged
last row tile), fp32-grade results against fp64, agreement with the fp32-row forms of the same kernels to the 22 bits an image
carries, a row's bits independent of the rows around it, run-to-run bits, the range guard, and a whole block against the block
without images."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def slot_k(order, kt, c, e):
    """k of slot e of chunk c of k-tile kt (csrc/gemm_x3_tile.hip: x3_tile_k)."""
    if order == 1:
        return 16 * kt + 8 * (e >> 2) + 4 * c + (e & 3)
    if order == 2:
        return 64 * (kt >> 2) + 16 * (e & 3) + 2 * (2 * (kt & 3) + c) + (e >> 2)
    return 16 * kt + 8 * c + e


def decode(img):
    """RowImage -> (rows, K) fp32: hi + lo of every element, unscaled."""
    M, K = img.rows, img.shape[-1]
    TM, KT = (M + 127) // 128, K // 16
    t = img.data.view(torch.float16).view(TM, KT, 2, 2, 128, 8).float()          # tile kt hl chunk row e
    v = (t[:, :, 0] + t[:, :, 1]) / (2.0 ** img.exp)                             # tile kt chunk row e
    idx = torch.tensor([[[slot_k(img.order, kt, c, e) for e in range(8)] for c in range(2)] for kt in range(KT)], device=v.device)
    out = torch.zeros(TM * 128, K, device=v.device)
    out[:, idx.reshape(-1)] = v.permute(0, 3, 1, 2, 4).reshape(TM * 128, KT * 16)
    return out[:M], out[M:]


def _ints(shape, lo, hi, g, density=1.0):
    t = torch.randint(lo, hi, shape, generator=g).float()
    if density < 1.0:
        t = t * (torch.rand(shape, generator=g) < density).float()
    return t.to(DEV)


@pytest.mark.parametrize("M,K,N", [(300, 512, 256), (128, 1024, 512), (1000, 1024, 256), (77, 4096, 1024)])
def test_rows_image_and_its_consumer_are_exact_on_integers(M, K, N):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(M + K + N)
    x = _ints((M, K), -4, 5, g, 0.5)
    w = _ints((N, K), -2, 3, g, 0.1)
    b = _ints((N,), -20, 20, g)
    r = _ints((M, N), -30, 30, g)
    img = Fn.x3_rows_image(x)
    rows, tail = decode(img)
    assert torch.equal(rows, x) and float(tail.abs().max() if tail.numel() else 0.0) == 0.0
    want = x.double() @ w.double().T + b.double()
    got = Fn.x3_tile_linear(img, w, b)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    assert torch.equal(got, want.float())
    assert torch.equal(Fn.x3_tile_linear(img, w, b, act="relu", residual=r), (want.clamp_min(0) + r.double()).float())
    assert torch.equal(got, Fn.x3_tile_linear(x, w, b))                        # the fp32-row form of the kernel
    # real operands: against fp64 next to the library's fp32 GEMM; a row's bits do not depend on the rows around it
    xf = torch.randn(M, K, generator=g).to(DEV)
    wf = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    ref = xf.double() @ wf.double().T
    scale = xf.double().abs() @ wf.double().abs().T
    got = Fn.x3_tile_linear(Fn.x3_rows_image(xf), wf, None)
    e, e_lib = float(((got.double() - ref).abs() / scale).max()), float((((xf @ wf.T).double() - ref).abs() / scale).max())
    assert e <= max(1.5 * e_lib, 6e-7), (e, e_lib)
    assert torch.equal(got, Fn.x3_tile_linear(Fn.x3_rows_image(xf), wf, None))
    lo = M // 3
    assert torch.equal(got[lo:lo + 50], Fn.x3_tile_linear(Fn.x3_rows_image(xf[lo:lo + 50].contiguous()), wf, None))


@pytest.mark.parametrize("M,C", [(300, 1024), (131, 768), (64, 512)])
def test_layer_norm_writes_the_image_of_its_own_output(M, C):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(M, C, generator=g) * 3 + 0.5).to(DEV)
    norm = torch.nn.LayerNorm(C, eps=1e-6).to(DEV)
    norm.weight.data = torch.rand(C, generator=g).to(DEV) + 0.5
    norm.bias.data = torch.randn(C, generator=g).to(DEV) * 0.2
    assert Fn.layer_norm_rows_image_ok(x, norm)
    want = Fn.add_layer_norm(x, None, norm)
    img = Fn.layer_norm_rows_image(x, norm)
    rows, tail = decode(img)
    assert float((rows - want).abs().max()) <= 2.0 ** -21 * float(want.abs().max())
    assert float(tail.abs().max() if tail.numel() else 0.0) == 0.0
    assert float((want - F.layer_norm(x.double(), (C,), norm.weight.double(), norm.bias.double(), 1e-6)).abs().max()) < 1e-5
    assert torch.equal(img.data, Fn.layer_norm_rows_image(x, norm).data)


@pytest.mark.parametrize("M,K,H", [(300, 512, 1024), (1000, 1024, 4096)])
def test_gelu_form_writes_the_next_row_image(M, K, H):
    """fc1 -> GELU -> fc2: the hidden activation only exists as a row image (k order 1)."""
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(H)
    x = torch.randn(M, K, generator=g).to(DEV)
    w1 = (torch.randn(H, K, generator=g) / K ** 0.5).to(DEV)
    b1 = (torch.randn(H, generator=g) * 0.3).to(DEV)
    w2 = (torch.randn(K, H, generator=g) / H ** 0.5).to(DEV)
    b2 = (torch.randn(K, generator=g) * 0.3).to(DEV)
    hid = Fn.x3_tile_linear(Fn.x3_rows_image(x), w1, b1, act="gelu")
    Fn.X3_GUARD.check_now(torch.device(DEV))
    assert isinstance(hid, Fn.RowImage) and hid.order == 1 and hid.shape == (M, H)
    want = Fn.x3_tile_linear(x, w1, b1, act="gelu")                            # fp32 rows through the same products
    rows, _ = decode(hid)
    assert float((rows - want).abs().max()) <= 2.0 ** -21 * float(want.abs().max())
    ref = F.gelu(x.double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double() + x.double()
    got = Fn.x3_tile_linear(hid, w2, b2, residual=x)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    lib = F.gelu(x @ w1.T + b1) @ w2.T + b2 + x
    s = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= max(2.0 * float((lib.double() - ref).abs().max()), 2e-6 * s)
    # integers through the order-1 slots: an image built by hand in that order meets the weights packed for it
    xi = _ints((M, H), -3, 4, g, 0.3)
    wi = _ints((K, H), -2, 3, g, 0.1)
    nat = Fn.x3_rows_image(xi)
    perm = torch.tensor([slot_k(1, kt, c, e) for kt in range(H // 16) for c in range(2) for e in range(8)], device=DEV)
    src = torch.tensor([slot_k(0, kt, c, e) for kt in range(H // 16) for c in range(2) for e in range(8)], device=DEV)
    xp = torch.empty_like(xi)
    xp[:, src] = xi[:, perm]                                                   # natural slot s holds the value of order-1 slot s
    hand = Fn.RowImage(Fn.x3_rows_image(xp).data, (M, H), nat.exp, 1)
    assert torch.equal(decode(hand)[0], xi)
    assert torch.equal(Fn.x3_tile_linear(hand, wi, None), (xi.double() @ wi.double().T).float())


@pytest.mark.parametrize("B,L,heads", [(2, 1100, 8), (1, 1024, 16), (3, 1281, 8)])
def test_attention_writes_the_projections_row_image(B, L, heads):
    from dvis_plus_amd import functions as Fn
    C = heads * 64
    g = torch.Generator().manual_seed(L)
    x = torch.randn(B, L, C, generator=g).to(DEV)
    wq = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(DEV)
    bq = (torch.randn(3 * C, generator=g) * 0.1).to(DEV)
    wp = (torch.randn(C, C, generator=g) / C ** 0.5).to(DEV)
    bp = (torch.randn(C, generator=g) * 0.1).to(DEV)
    if not Fn.x3_qkv_attention_ok(x, wq, heads):
        pytest.skip("the fused qkv + attention form does not serve this shape")
    want = Fn.x3_qkv_attention(x, wq, bq, heads)                               # fp32 rows
    xin = Fn.x3_rows_image(x)
    assert Fn.x3_qkv_attention_ok(xin, wq, heads)
    assert torch.equal(Fn.x3_qkv_attention(xin, wq, bq, heads), want)          # row image in: the same products
    img = Fn.x3_qkv_attention(xin, wq, bq, heads, out_image=True)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    assert img.order == 2 and img.shape == (B, L, C)
    rows, tail = decode(img)
    assert float((rows - want.reshape(B * L, C)).abs().max()) <= 2.0 ** -21 * float(want.abs().max())
    assert float(tail.abs().max() if tail.numel() else 0.0) == 0.0
    assert torch.equal(img.data, Fn.x3_qkv_attention(xin, wq, bq, heads, out_image=True).data)
    # ... and the projection behind it
    ref = want.reshape(B * L, C).double() @ wp.double().T + bp.double()
    got = Fn.x3_tile_linear(img, wp, bp).reshape(B * L, C)
    s = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= 2e-6 * s
    # q / k / v as in torch
    q, k, v = (x @ wq.T + bq).reshape(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4).double()
    att = torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v
    assert float((want.double() - att.permute(0, 2, 1, 3).reshape(B, L, C)).abs().max()) <= 2e-5 * float(att.abs().max())


def test_range_guard_sees_an_image_value_out_of_range():
    from dvis_plus_amd import functions as Fn
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(200, 512, generator=g).to(DEV)
    w = (torch.randn(256, 512, generator=g) / 23).to(DEV)
    Fn.X3_GUARD.check_now(dev)
    Fn.x3_tile_linear(Fn.x3_rows_image(x), w, None)
    Fn.X3_GUARD.check_now(dev)
    x[17, 5] = 3.0e6                                                           # beyond 65504 / 2^xexp
    Fn.x3_tile_linear(Fn.x3_rows_image(x), w, None)
    with pytest.raises(Fn.X3RangeError):
        Fn.X3_GUARD.check_now(dev)
    Fn.X3_GUARD.check_now(dev)


def test_vit_block_through_row_images(monkeypatch):
    from dvis_plus_amd import functions as Fn
    from dvis_plus_amd.vit_adapter import Block
    torch.manual_seed(0)
    blk = Block(512, 8, qkv_bias=True, init_values=0.5).to(DEV).eval()
    for p in blk.parameters():
        if p.dim() == 1:
            p.data = p.data + torch.randn_like(p) * 0.1
    x = torch.randn(2, 1100, 512, device=DEV)
    assert blk.row_images_ok(x)
    calls = []
    orig = Fn.layer_norm_rows_image
    monkeypatch.setattr(Fn, "layer_norm_rows_image", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    got = blk(x)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    assert len(calls) == 2, "the block did not take the row-image forms"
    monkeypatch.setattr(Fn, "X3_ROW_IMAGES", False)
    calls.clear()
    ref = blk(x)
    assert not calls
    assert float((got - ref).abs().max()) <= 3e-6 * float(ref.abs().max())
    # ... and against torch in fp64
    d = blk.double()
    h = F.layer_norm(x.double(), (512,), d.norm1.weight, d.norm1.bias, d.norm1.eps)
    q, k, v = F.linear(h, d.attn.qkv.weight, d.attn.qkv.bias).reshape(2, 1100, 3, 8, 64).permute(2, 0, 3, 1, 4)
    a = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).permute(0, 2, 1, 3).reshape(2, 1100, 512)
    y = x.double() + d.ls1.gamma * F.linear(a, d.attn.proj.weight, d.attn.proj.bias)
    h = F.layer_norm(y, (512,), d.norm2.weight, d.norm2.bias, d.norm2.eps)
    y = y + d.ls2.gamma * F.linear(F.gelu(F.linear(h, d.mlp.fc1.weight, d.mlp.fc1.bias)), d.mlp.fc2.weight, d.mlp.fc2.bias)
    blk.float()
    assert float((got.double() - y).abs().max()) <= 1e-5 * float(y.abs().max())
