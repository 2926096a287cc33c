"""The META_ARCH surface as the reference's launcher invokes it: ``with autocast(): inference_on_dataset(model, ...)``
(train_net_video.py:232, 259).  The product's forward() / stream() are fp32 islands (functions.no_autocast: the hand-written
fp32 / split-f16 kernels take fp32 tensors), so under the launcher's context the outputs are the plain call's, bit for bit —
and with that within BASELINE's 1e-3 of the fp32 CPU reference (the g10 comparison below, run INSIDE the context)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = [torch.float16, torch.bfloat16]


def _clip(T, seed, h=120, w=200):
    g = torch.Generator().manual_seed(seed)
    return {"image": [torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g).to(DEV) for _ in range(T)],
            "height": h, "width": w}


def _small(mode, task="vps"):
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    from pipeline_parity import perturb_msda
    m = build_dvis_plus_r50(mode, task=task, num_classes=20, num_queries=100, n_things=10, enc_layers=2, dec_layers=4,
                            tracker_layers=2, refiner_layers=2, object_mask_threshold=0.06)
    perturb_msda(m.sem_seg_head.pixel_decoder)
    return m.to(DEV)


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        x, y = a[k], b[k]
        if torch.is_tensor(x):
            assert x.dtype == y.dtype and torch.equal(x, y), k
        elif isinstance(x, (list, tuple)) and x and torch.is_tensor(x[0]):
            assert all(u.dtype == v.dtype and torch.equal(u, v) for u, v in zip(x, y)), k
        else:
            assert x == y, k


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mode,task", [("offline", "vps"), ("offline", "vis"), ("online", "vps"), ("online", "vss"),
                                       ("minvis", "vis")])
def test_video_meta_architectures_under_autocast_equal_the_plain_call(mode, task, dt):
    m = _small(mode, task)
    v = _clip(4, 3)
    plain = m([v])
    with torch.autocast("cuda", dtype=dt):
        assert torch.is_autocast_enabled()
        wrapped = m([v])
        assert torch.is_autocast_enabled()               # the island closes behind the call
    _same(plain, wrapped)
    if mode != "minvis" and task == "vps":
        assert len(plain["segments_infos"]) > 0, "degenerate test: no segment"


@pytest.mark.parametrize("dt", DTYPES)
def test_image_maskformer_under_autocast_equals_the_plain_call(dt):
    from dvis_plus_amd.meta_architecture import build_mask2former_r50
    from pipeline_parity import perturb_msda
    m = build_mask2former_r50(num_classes=19, num_queries=100, enc_layers=2, dec_layers=4, semantic_on=True, panoptic_on=True,
                              instance_on=True, object_mask_threshold=0.05, thing_ids=range(8))
    perturb_msda(m.sem_seg_head.pixel_decoder)
    m = m.to(DEV)
    img = torch.randint(0, 256, (3, 120, 160), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).to(DEV)
    plain = m([{"image": img, "height": 120, "width": 160}])[0]
    with torch.autocast("cuda", dtype=dt):
        wrapped = m([{"image": img, "height": 120, "width": 160}])[0]
    assert plain["sem_seg"].dtype == wrapped["sem_seg"].dtype == torch.float32
    assert torch.equal(plain["sem_seg"], wrapped["sem_seg"])
    assert torch.equal(plain["panoptic_seg"][0], wrapped["panoptic_seg"][0]) and plain["panoptic_seg"][1] == wrapped["panoptic_seg"][1]
    for k in ("pred_masks", "scores", "pred_classes"):
        assert torch.equal(plain["instances"][k], wrapped["instances"][k]), k


@pytest.mark.parametrize("dt", DTYPES)
def test_stream_under_autocast_equals_forward_and_leaves_the_consumers_context_alone(dt):
    m = _small("offline", "vps")
    clips = [_clip(4, s) for s in (3, 4, 5)]
    want = [m([c]) for c in clips]
    got = []
    with torch.autocast("cuda", dtype=dt):
        for out in m.stream(clips):
            # the consumer's code between two clips runs in ITS context: stream() must not leave autocast switched off
            assert torch.is_autocast_enabled()
            assert (torch.ones(4, 4, device=DEV) @ torch.ones(4, 4, device=DEV)).dtype == dt
            got.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items() if k != "ready_event"})
    for a, b in zip(got, want):
        assert torch.equal(a["pred_masks"], b["pred_masks"]) and a["segments_infos"] == b["segments_infos"]
        assert a["pred_ids"] == b["pred_ids"]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mode", ["offline", "online"])
def test_g10_reference_forward_comparison_inside_autocast(mode, dt):
    """The reference's own forward (golden g10_window_loop: its DVIS_Plus_offline / _online forward -> run_window_inference ->
    post_processing -> inference_video_vps on the fp32 CPU path) against the product called the launcher's way."""
    import g10_model as G
    from test_golden_gpu import _g10_lists_equal
    m, g, cfg, frames = G.build(mode, "vps", DEV)
    o = g.outs
    m.debug_stages = {}
    tag, key = ("off_vps", "off_refiner_masks") if mode == "offline" else ("on_vps", "on_masks")
    with torch.no_grad(), torch.autocast("cuda", dtype=dt):
        out = m([G.video(frames, cfg, device=DEV)])
        got_masks = m.debug_stages["mask_fn"](None)
    assert got_masks.dtype == torch.float32
    _g10_lists_equal(out, o, tag)
    err = float((got_masks.cpu() - o[key][0]).abs().max())
    assert err <= 1e-3, err                                                             # BASELINE.json's literal bound
    m.debug_stages = {}
    with torch.no_grad():
        plain = m([G.video(frames, cfg, device=DEV)])
    assert torch.equal(plain["pred_masks"], out["pred_masks"]) and plain["segments_infos"] == out["segments_infos"]


@pytest.mark.parametrize("dt", DTYPES)
def test_submodules_called_alone_under_autocast_stay_fp32(dt):
    """The registry surfaces one level down (a reference META_ARCH that builds OUR head / tracker / refiner and calls them inside
    its own autocast region): each is its own island, half inputs from the caller's region are taken as fp32."""
    m = _small("offline", "vps")
    v = _clip(3, 9)
    with torch.no_grad():
        images, _ = m.preprocess(v["image"])
        feats = m.backbone(images)
        plain = m.sem_seg_head(feats)
        with torch.autocast("cuda", dtype=dt):
            feats_ac = m.backbone(images)
            wrapped = m.sem_seg_head(feats_ac)
            half_in = m.sem_seg_head({k: f.to(dt) for k, f in feats.items()})      # a foreign half-precision backbone
        for k in ("res2", "res5"):
            assert feats_ac[k].dtype == torch.float32 and torch.equal(feats_ac[k], feats[k])
        for k in ("pred_logits", "pred_embds", "pred_embds_without_norm", "mask_features"):
            assert wrapped[k].dtype == torch.float32 and torch.equal(wrapped[k], plain[k]), k
            assert half_in[k].dtype == torch.float32
        to = lambda z: z
        trk_plain = m.tracker(plain["pred_embds"], None, frame_embeds_no_norm=plain["pred_embds_without_norm"], need_masks=False)
        ref_plain = m.refiner(trk_plain["pred_embds"], plain["pred_embds_without_norm"], None, need_masks=False)
        with torch.autocast("cuda", dtype=dt):
            trk = m.tracker(plain["pred_embds"], None, frame_embeds_no_norm=plain["pred_embds_without_norm"], need_masks=False)
            ref = m.refiner(trk["pred_embds"], plain["pred_embds_without_norm"], None, need_masks=False)
        for k in ("pred_logits", "pred_embds"):
            assert torch.equal(trk[k], trk_plain[k]) and torch.equal(ref[k], ref_plain[k]), k
