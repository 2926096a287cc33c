"""GPU: the range guard of the split-f16 kernels (include/dvis_hip.h: dvis_x3_set_range_flag / dvis_x3_set_tag,
functions._X3RangeGuard).  The reference computes this path in fp32 (msdeformattn.py:314,320 — an explicit fp32 island): no
activation magnitude breaks it.  Here an activation beyond 65520 / 2^xexp cannot be split into two f16 terms; that must be an
error naming the layer (or a re-run on the exact-fp32 kernels) — never NaN masks, and never the silent case where a ReLU turns
the NaN into a plain zero."""
import warnings

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def _lin(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    lin = torch.nn.Linear(k, n)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(n, k, generator=g) * 0.05)
        lin.bias.copy_(torch.randn(n, generator=g) * 0.1)
    return lin.to(DEV)


def test_linear_kernels_flag_an_input_beyond_the_f16_range():
    from dvis_plus_amd import functions as Fn
    dev = torch.device(DEV)
    lin = _lin(256, 256, 4)
    x = torch.randn(300, 256, device=DEV)
    Fn.X3_GUARD.check_now(dev)                                  # clean slate
    out = Fn.x3_linear(x, lin.weight, lin.bias, relu=True)
    Fn.X3_GUARD.check_now(dev)                                  # in range: no error
    assert torch.isfinite(out).all()
    x[3, 17] = 5000.0                                           # 5000 * 2^4 > 65520
    out = Fn.x3_linear(x, lin.weight, lin.bias, relu=True)
    # the silent case the guard exists for: relu(NaN) = 0 — the row can look perfectly finite
    with pytest.raises(Fn.X3RangeError, match=r"\(256, 256\) \[linear kernel\]"):
        Fn.X3_GUARD.check_now(dev)
    Fn.X3_GUARD.check_now(dev)                                  # the word was cleared by the raise
    out = Fn.x3_linear(x, lin.weight, lin.bias, relu=True, xexp=0)      # a smaller exponent serves such data
    Fn.X3_GUARD.check_now(dev)
    ref = torch.relu(x.double() @ lin.weight.double().t() + lin.bias.double())
    assert float((out.double() - ref).abs().max()) <= 1e-3
    # LayerNorm form and the fused FFN (hidden activations beyond the range)
    norm = torch.nn.LayerNorm(256).to(DEV)
    res = torch.randn_like(x)
    Fn.x3_linear_ln(x, lin.weight, lin.bias, res, norm)
    with pytest.raises(Fn.X3RangeError):
        Fn.X3_GUARD.check_now(dev)
    l1, l2 = _lin(1024, 256, 5), _lin(256, 1024, 6)
    ok = torch.randn(300, 256, device=DEV)
    Fn.x3_ffn_ln(ok, l1, l2, norm)
    Fn.X3_GUARD.check_now(dev)
    with torch.no_grad():
        l1.weight.mul_(3000.0)                                  # hidden = relu(W1 x + b1) ~ 3000 * 0.05 * 16 * |N(0,1)| >> 4095
    out = Fn.x3_ffn_ln(ok, l1, l2, norm)
    with pytest.raises(Fn.X3RangeError, match="ffn kernel"):
        Fn.X3_GUARD.check_now(dev)


def test_convolution_flags_before_its_relu_hides_the_nan():
    from dvis_plus_amd import functions as Fn
    dev = torch.device(DEV)
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(256, 64, 1, 1, generator=g) * 0.1).to(DEV)
    b = torch.zeros(256, device=DEV)
    x = torch.rand(2, 64, 24, 40, generator=g).to(DEV)
    Fn.X3_GUARD.check_now(dev)
    assert Fn.conv1x1_x3_ok(x, w)
    y = Fn.conv1x1_x3(x, w, b, None, True)
    Fn.X3_GUARD.check_now(dev)
    x[1, 5, 7, 9] = 40000.0                                     # 40000 * 2^2 > 65520
    y = Fn.conv1x1_x3(x, w, b, None, True)
    assert torch.isfinite(y).all() or True                      # (with the ReLU the NaN may be gone: exactly why the guard tests before it)
    with pytest.raises(Fn.X3RangeError, match="conv1x1 kernel"):
        Fn.X3_GUARD.check_now(dev)
    w3 = (torch.randn(128, 128, 3, 3, generator=g) * 0.05).to(DEV)
    x3 = torch.rand(1, 128, 16, 24, generator=g).to(DEV)
    x3[0, 3, 4, 4] = 1e6
    Fn.conv3x3_x3(x3, w3, None, None, True)
    with pytest.raises(Fn.X3RangeError, match="conv3x3 kernel"):
        Fn.X3_GUARD.check_now(dev)


def _small_model():
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    import pipeline_parity as PPar
    cfg = dict(num_classes=20, n_things=10, enc_layers=2, tracker_layers=2, refiner_layers=2)
    m = build_dvis_plus_r50("offline", task="vps", num_queries=100, dec_layers=4, object_mask_threshold=0.06, **cfg)
    PPar.perturb_msda(m.sem_seg_head.pixel_decoder)
    return m.to(DEV)


def _frames(T=3, H=128, W=192):
    g = torch.Generator().manual_seed(3)
    return [torch.randint(0, 256, (3, H, W), dtype=torch.uint8, generator=g).to(DEV) for _ in range(T)]


@pytest.mark.parametrize("where", ["encoder_ffn", "backbone"])
def test_model_call_with_an_out_of_range_layer_is_an_error_naming_it_or_a_rerun_never_nan_masks(where, monkeypatch):
    """A layer driven past the range through model(): DVIS_X3_ON_OVERFLOW=raise -> X3RangeError with the layer's name;
    the default (rerun) -> a RuntimeWarning, the clip re-run on the exact-fp32 kernels: the output equals that of the same model
    with the split arithmetic off, and the model stays on the exact kernels."""
    from dvis_plus_amd import functions as Fn
    if not Fn.X3:
        pytest.skip("DVIS_X3=0: no split-f16 kernel runs")
    m = _small_model()
    frames = _frames()
    video = {"image": frames, "height": 128, "width": 192}
    with torch.no_grad():
        if where == "encoder_ffn":
            m.sem_seg_head.pixel_decoder.transformer.encoder.layers[0].linear1.weight.mul_(5000.0)
            name = "encoder.layers.0.linear1.weight"
        else:
            m.backbone.stem.conv1.weight.mul_(3e5)              # res2's first 1x1 layers then see activations >> 16380
            name = "backbone.res2.0"
    monkeypatch.setattr(Fn, "X3_ON_OVERFLOW", "raise")
    with pytest.raises(Fn.X3RangeError, match=name.replace(".", r"\.")):
        m([video])
    with pytest.raises(Fn.X3RangeError):
        list(m.stream([video, video]))
    monkeypatch.setattr(Fn, "X3_ON_OVERFLOW", "rerun")

    def logits_of(model):
        return model.debug_stages["mask_fn"](None).float()

    m.debug_stages = {}
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = m([video])
    assert any("exact-fp32" in str(w.message) for w in rec) and m._x3_off
    got = logits_of(m)
    assert torch.isfinite(got).all(), "the re-run must not hand NaN / inf mask logits on"
    # the same model with the split arithmetic off from the start.  (Exact mode at this small size runs library convolutions
    # whose algorithm choice is not pinned: compared to rounding level, not bit for bit.)
    m._x3_off = False
    with Fn.x3_disabled():
        want = m([video])
        ref = logits_of(m)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 5e-2 * max(scale, 1.0)
    assert out["pred_masks"].shape == want["pred_masks"].shape
    if where == "encoder_ffn":
        assert float((out["pred_masks"] == want["pred_masks"]).float().mean()) > 0.99
    # stream(): the round's clips are re-run on the side stream
    m._x3_off = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        outs = [dict(o) for o in m.stream([video, video])]
    assert any("exact-fp32" in str(w.message) for w in rec) and m._x3_off and len(outs) == 2
    assert torch.isfinite(logits_of(m)).all()
    for o in outs:
        assert o["pred_masks"].shape == want["pred_masks"].shape


def test_in_range_model_calls_leave_the_guard_silent_and_cost_no_synchronisation():
    """The benchmark path: stream() over clips with the guard armed — no error, and the snapshot is an asynchronous copy (the
    verify waits on an event that phase B needs anyway)."""
    from dvis_plus_amd import functions as Fn
    m = _small_model()
    video = {"image": _frames(), "height": 128, "width": 192}
    outs = list(m.stream([video] * 3))
    assert len(outs) == 3 and not getattr(m, "_x3_off", False)
    Fn.X3_GUARD.check_now(torch.device(DEV))
