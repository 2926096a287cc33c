"""GPU: csrc/bneck_x3.hip — the res2 bottlenecks of the R50 (detectron2 BottleneckBlock) as a chain of one
launch per block (conv2 -> conv3 + shortcut -> the next block's conv1), the 64-channel maps between the launches as pre-split
operand images.

What is pinned here: the layout (integer operands come out bit for bit: accumulator channel order through two chained
contractions, the nine shifted reads of an operand image, zero padding, row / group boundaries, widths that are not a multiple
of the 32-pixel group, the projection-shortcut form, the block without a chained conv1), fp32-grade results against an fp64
evaluation, agreement with the layer-by-layer split-f16 path, a frame's bits independent of its batch mates, run-to-run bits,
the range guard."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def _blocks(nblocks, g, integer=False):
    """Folded weights / shifts of a res2-shaped stage (first block: projection shortcut)."""
    out = []
    for i in range(nblocks):
        cin = 64 if i == 0 else 256

        def w(co, ci, k, density):
            if integer:
                t = torch.randint(-1, 2, (co, ci, k, k), generator=g).float()
                return t * (torch.rand(co, ci, k, k, generator=g) < density).float()
            return torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5

        def b(co, lo, hi):
            return torch.randint(lo, hi, (co,), generator=g).float() if integer else torch.randn(co, generator=g) * 0.3
        # (integer form: sparse {-1, 0, 1} weights and small shifts keep about half of every map active and every activation
        # below 2000 after four blocks — inside the split-f16 range, where the whole chain is exact)
        blk = dict(w1=w(64, cin, 1, 0.2 if i == 0 else 0.015), b1=b(64, -2, 3), w2=w(64, 64, 3, 0.02), b2=b(64, -2, 3),
                   w3=w(256, 64, 1, 0.08), b3=b(256, -3, 2), ws=w(256, 64, 1, 0.08) if i == 0 else None,
                   bs=b(256, -3, 2) if i == 0 else None)
        out.append({k: (None if v is None else v.to(DEV)) for k, v in blk.items()})
    return out


def _reference(x, blocks, dtype=torch.float64):
    """relu(conv1) -> relu(conv2) -> relu(conv3 + shortcut), block by block; returns every block's output and the maximum
    magnitude any activation reached."""
    x = x.to(dtype)
    outs, amax = [], float(x.abs().max())
    for b in blocks:
        c = {k: (None if v is None else v.to(dtype)) for k, v in b.items()}
        a1 = F.relu(F.conv2d(x, c["w1"], c["b1"]))
        a2 = F.relu(F.conv2d(a1, c["w2"], c["b2"], padding=1))
        sc = x if c["ws"] is None else F.conv2d(x, c["ws"], c["bs"])
        x = F.relu(F.conv2d(a2, c["w3"], c["b3"]) + sc)
        amax = max(amax, float(a1.max()), float(a2.max()), float(x.max()))
        outs.append(x)
    return outs, amax


@pytest.mark.parametrize("N,H,W,nblocks", [
    (2, 9, 40, 3),          # two groups per row, the second one 8 pixels wide
    (3, 5, 64, 3),          # rows of exactly two groups; 3 * 5 * 2 = 30 groups: the last tile of 8 is ragged
    (1, 12, 33, 2),         # one pixel in the second group; two blocks (projection form feeding the last block directly)
    (2, 7, 17, 4),          # less than one group per row; four blocks (two identity blocks with a chained conv1)
    (1, 1, 1, 3),           # a single pixel
])
def test_layout_is_exact_on_integer_operands(N, H, W, nblocks):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
    blocks = _blocks(nblocks, g, integer=True)
    x = (torch.rand(N, 64, H, W, generator=g) < 0.3).float().to(DEV) * torch.randint(1, 4, (N, 64, H, W), generator=g).float().to(DEV)
    refs, amax = _reference(x, blocks)
    assert amax < 16000, f"test data leaves the split-f16 range ({amax})"      # (the comparison below would be meaningless)
    assert float(refs[-1].max()) > 0, "degenerate test data"
    assert Fn.bneck_stage_x3_ok(x, blocks)
    got = Fn.bneck_stage_x3(x, blocks)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    assert torch.equal(got, refs[-1].float())
    # every block boundary: the stage cut after k blocks ends in the form without a chained conv1
    for k in range(2, nblocks):
        assert torch.equal(Fn.bneck_stage_x3(x, blocks[:k]), refs[k - 1].float())


@pytest.mark.parametrize("N,H,W", [(2, 23, 40), (1, 46, 80), (3, 16, 50)])
def test_stage_is_fp32_grade_and_matches_the_layer_by_layer_path(N, H, W, monkeypatch):
    from dvis_plus_amd import functions as Fn
    g = torch.Generator().manual_seed(H + W)
    blocks = _blocks(3, g)
    x = torch.randn(N, 64, H, W, generator=g).relu().to(DEV)
    refs, _ = _reference(x, blocks)
    ref = refs[-1]
    got = Fn.bneck_stage_x3(x, blocks)
    Fn.X3_GUARD.check_now(torch.device(DEV))
    scale = float(ref.abs().max())
    e = float((got.double() - ref).abs().max()) / scale
    # the same stage in fp32 on the library (its own rounding against fp64 is the yardstick)
    lib, _ = _reference(x, blocks, torch.float32)
    e_lib = float((lib[-1].double() - ref).abs().max()) / scale
    assert e <= max(2.0 * e_lib, 1e-6), (e, e_lib)
    # ... and layer by layer through the split-f16 convolution kernels (csrc/conv1x1_x3.hip + the 3x3 kernels)
    y = x
    for b in blocks:
        a1 = Fn.conv1x1_x3(y, b["w1"], b["b1"], None, True)
        a2 = Fn.conv3x3_x3(a1, b["w2"], b["b2"], None, True)
        if b["ws"] is not None:
            y = Fn.conv1x1_x3_dual(a2, b["w3"], b["b3"], y, b["ws"], b["bs"], relu=True)
        else:
            y = Fn.conv1x1_x3(a2, b["w3"], b["b3"], y, True)
    assert float((got - y).abs().max()) / scale <= 2e-6
    # run-to-run bits; a frame alone = the frame in the batch
    assert torch.equal(got, Fn.bneck_stage_x3(x, blocks))
    for n in range(N):
        assert torch.equal(got[n:n + 1], Fn.bneck_stage_x3(x[n:n + 1].contiguous(), blocks))


def test_reserved_cus_and_launch_chunks_do_not_change_bits(monkeypatch):
    from dvis_plus_amd import functions as Fn, native
    g = torch.Generator().manual_seed(5)
    blocks = _blocks(3, g)
    x = torch.randn(5, 64, 20, 70, generator=g).relu().to(DEV)
    got = Fn.bneck_stage_x3(x, blocks)
    prev = native.lib().dvis_x3_set_reserve(64)
    try:
        assert torch.equal(got, Fn.bneck_stage_x3(x, blocks))
    finally:
        native.lib().dvis_x3_set_reserve(prev)


def test_range_guard_names_the_chain():
    from dvis_plus_amd import functions as Fn
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(7)
    blocks = _blocks(3, g)
    x = torch.randn(1, 64, 10, 40, generator=g).relu().to(DEV)
    Fn.X3_GUARD.check_now(dev)
    Fn.bneck_stage_x3(x, blocks)
    Fn.X3_GUARD.check_now(dev)
    # an activation beyond 65520 / 2^xexp inside the chain: a shift that pushes conv2's output out of the f16 range
    bad = [dict(b) for b in blocks]
    bad[1]["b2"] = blocks[1]["b2"] + 40000.0
    Fn.bneck_stage_x3(x, bad)
    with pytest.raises(Fn.X3RangeError, match=r"bottleneck chain"):
        Fn.X3_GUARD.check_now(dev)
    # ... and in the stage's input (the first conv1's operand)
    x2 = x.clone()
    x2[0, 3, 4, 5] = 30000.0
    Fn.bneck_stage_x3(x2, blocks)
    with pytest.raises(Fn.X3RangeError):
        Fn.X3_GUARD.check_now(dev)
    Fn.X3_GUARD.check_now(dev)


def test_the_r50_takes_the_chain(monkeypatch):
    """ResNet.forward routes res2 through the chain (and not when it is switched off): same outputs within the kernels' rounding."""
    from dvis_plus_amd import functions as Fn
    from dvis_plus_amd.backbone import build_resnet50
    torch.manual_seed(0)
    m = build_resnet50().to(DEV).eval()
    x = torch.randn(2, 3, 96, 160, device=DEV)          # (normalised-image scale: random-init stages multiply the activations' scale)
    calls = []
    orig = Fn.bneck_stage_x3
    monkeypatch.setattr(Fn, "bneck_stage_x3", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    out = m(x)
    assert calls, "res2 did not take csrc/bneck_x3.hip"
    monkeypatch.setattr(Fn, "X3_BNECK", False)
    calls.clear()
    ref = m(x)
    assert not calls
    for k in out:
        s = float(ref[k].abs().max())
        assert float((out[k] - ref[k]).abs().max()) <= 5e-6 * s, k
    Fn.X3_GUARD.check_now(torch.device(DEV))      # (and nothing left the split-f16 range on the way)
