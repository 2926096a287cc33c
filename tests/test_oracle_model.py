"""CPU: the torch oracle (oracle/dvis_torch.py) against goldens captured from the imported reference modules
(tests/golden/g2..g6).  Same ops in the same order as the reference => tight tolerances / bit-exact integers."""
import numpy as np
import pytest
import torch

from conftest import Golden
from oracle import dvis_torch as O

TOL = dict(rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["g2_pixel_decoder", "g7_pixel_decoder_d32"])
def test_pixel_decoder_and_msdeformattn_module(name):
    g = Golden(name)
    sd, i, o = g.sd, g.ins, g.outs
    feats = {k[5:]: v for k, v in i.items() if k.startswith("feat_")}
    mf, out0, ms = O.pixel_decoder_forward(sd, feats, nheads=2, enc_layers=2)
    torch.testing.assert_close(mf, o["mask_features"], **TOL)
    torch.testing.assert_close(out0, o["out0"], **TOL)
    for a, b in zip(ms, (o["ms0"], o["ms1"], o["ms2"])):
        torch.testing.assert_close(a, b, **TOL)
    attn = O.ms_deform_attn_module(sd, "transformer.encoder.layers.0.self_attn", i["attn_query"], i["attn_ref"],
                                   i["attn_src"], [(2, 3), (4, 6), (8, 12)], 2, 4)
    torch.testing.assert_close(attn, o["attn_out"], **TOL)


@pytest.mark.parametrize("name", ["g3_decoder_dvisplus", "g3_decoder_dvisplus_d32"])
def test_decoder_dvisplus(name):
    g = Golden(name)
    sd, i, o = g.sd, g.ins, g.outs
    out = O.decoder_forward(sd, [i["x0"], i["x1"], i["x2"]], i["mask_features"], nheads=2, dec_layers=3)
    for k in ("pred_logits", "pred_masks", "pred_embds", "pred_embds_without_norm", "pred_reid_embed"):
        torch.testing.assert_close(out[k], o[k], **TOL)
    for n in range(3):
        torch.testing.assert_close(out["aux_logits"][n], o[f"aux{n}_logits"], **TOL)
        torch.testing.assert_close(out["aux_masks"][n], o[f"aux{n}_masks"], **TOL)


@pytest.mark.parametrize("name", ["g3_decoder_image", "g3_decoder_image_d32"])
def test_decoder_image_config1(name):
    g = Golden(name)
    sd, i, o = g.sd, g.ins, g.outs
    out = O.decoder_forward(sd, [i["x0"], i["x1"], i["x2"]], i["mask_features"], nheads=2, dec_layers=3,
                            dvis_plus=False)
    torch.testing.assert_close(out["pred_logits"], o["pred_logits"], **TOL)
    torch.testing.assert_close(out["pred_masks"], o["pred_masks"], **TOL)


def test_tracker_with_resume_and_indices():
    g = Golden("g4_tracker")
    sd, i, o, cfg = g.sd, g.ins, g.outs, g.meta["cfg"]
    trk = O.Tracker(sd, nheads=cfg["heads"], layers=cfg["layers"])
    T1 = cfg["T1"]
    fe, fn, mf = i["frame_embeds"], i["frame_embeds_no_norm"], i["mask_features"]
    a = trk.forward(fe[:, :, :T1], mf[:, :T1], resume=False, frame_embeds_no_norm=fn[:, :, :T1])
    b = trk.forward(fe[:, :, T1:], mf[:, T1:], resume=True, frame_embeds_no_norm=fn[:, :, T1:])
    for tag, r in (("a", a), ("b", b)):
        assert np.array_equal(r["indices"], o[f"{tag}_indices"].numpy())          # bit-exact assignment
        for k in ("pred_logits", "pred_masks", "pred_embds", "pred_references"):
            torch.testing.assert_close(r[k], o[f"{tag}_{k}"], **TOL)


def test_refiner():
    g = Golden("g4_refiner")
    sd, i, o, cfg = g.sd, g.ins, g.outs, g.meta["cfg"]
    r = O.refiner_forward(sd, i["instance_embeds"], i["frame_embeds"], i["mask_features"], cfg["heads"], cfg["layers"])
    for k in ("pred_logits", "pred_masks", "pred_embds"):
        torch.testing.assert_close(r[k], o[k], **TOL)


def test_match_embds_indices_bit_exact():
    g = Golden("g5_match")
    i, o = g.ins, g.outs
    for n in range(g.meta["ncases"]):
        idx = O.match_embds(i[f"ref{n}"], i[f"cur{n}"])
        assert np.array_equal(idx, o[f"idx{n}"].numpy()), n


def test_postprocessing_integer_outputs_bit_exact():
    g = Golden("g6_postprocess")
    i, o, cfg = g.ins, g.outs, g.meta["cfg"]
    logits, aux = O.post_processing(i["pred_logits"], i["aux_logits"])
    torch.testing.assert_close(logits[None], o["pp_logits"], rtol=0, atol=0)
    torch.testing.assert_close(aux, o["pp_aux"], rtol=0, atol=0)
    masks = i["pred_masks"][0]
    img, out_hw, first = cfg["img_size"], cfg["out_hw"], cfg["first_resize"]
    s, l, ids, m = O.inference_video_vis(logits, masks, img, out_hw, first, cfg["K"], cfg["max_num"], aux)
    torch.testing.assert_close(s, o["vis_scores"], rtol=0, atol=0)
    assert torch.equal(l, o["vis_labels"]) and torch.equal(ids, o["vis_ids"]) and torch.equal(m, o["vis_masks"])
    pan, segs, out_ids = O.inference_video_vps(logits.clone(), masks, img, out_hw, first, cfg["K"], cfg["n_things"],
                                               cfg["object_mask_threshold"], cfg["overlap_threshold"], aux)
    assert torch.equal(pan, o["vps_masks"])
    assert [x["id"] for x in segs] == o["vps_seg_id"].tolist()
    assert [x["category_id"] for x in segs] == o["vps_seg_cat"].tolist()
    assert [x["isthing"] for x in segs] == o["vps_seg_isthing"].tolist()
    assert out_ids == o["vps_ids"].tolist()
    assert len(segs) >= 2                                   # the fixture exercises thing + merged stuff segments
    sem = O.inference_video_vss(logits, masks, img, out_hw, first, aux)
    assert torch.equal(sem, o["vss_masks"])
    assert torch.equal(O.get_instance_labels(i["pred_logits"]), o["instance_labels"])


def test_minvis_alignment_and_top10():
    g = Golden("g9_minvis")
    i, o, cfg = g.ins, g.outs, g.meta["cfg"]
    logits, masks, perms = O.minvis_post_processing(i["pred_logits"], i["pred_masks"], i["pred_embds"])
    torch.testing.assert_close(logits, o["pp_logits"], **TOL)
    assert torch.equal(masks, o["pp_masks"])                                    # pure re-ordering: bit-exact
    assert sorted(perms[-1].tolist()) == list(range(cfg["Q"]))
    s, l, m, _ = O.minvis_inference_video(logits[0], masks[0], cfg["img_size"], cfg["out_hw"], cfg["first_resize"],
                                          cfg["K"], cfg["topk"])
    order_ref, order = np.argsort(-o["scores"].numpy(), kind="stable"), np.argsort(-s.numpy(), kind="stable")
    np.testing.assert_allclose(s.numpy()[order], o["scores"].numpy()[order_ref], rtol=1e-6)
    assert np.array_equal(l.numpy()[order], o["labels"].numpy()[order_ref])
    assert torch.equal(m[order], o["masks"][order_ref])


# ---------------------------------------------------------------------------------------------------------------
# g10: the a12 COMPOSITION (window loop, state hand-off, which embeddings go where, mean logits, `keep`) — the reference's
# own DVIS_Plus_offline / DVIS_Plus_online forward + run_window_inference + post_processing + inference_video_*, driven
# with a stub `self` in the build container (tests/golden/gen_golden.py::g10_window_loop).
# ---------------------------------------------------------------------------------------------------------------
def _g10():
    from toy_backbone import ToyBackbone
    g = Golden("g10_window_loop")
    cfg, sd = g.meta["cfg"], g.sd
    bb = ToyBackbone().eval()
    bb.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
    frames = [f for f in g.ins["frames"]]
    kw = dict(nheads=(cfg["nheads"], cfg["trk_heads"]), enc_layers=cfg["enc_layers"], dec_layers=cfg["dec_layers"],
              tracker_layers=cfg["tracker_layers"], refiner_layers=cfg["refiner_layers"], window_size=cfg["window"],
              num_classes=cfg["K"], n_things=cfg["n_things"], max_num=cfg["max_num"],
              object_mask_threshold=cfg["object_mask_threshold"], overlap_threshold=cfg["overlap_threshold"],
              out_hw=tuple(cfg["out_hw"]))
    return g, cfg, sd, bb, frames, kw


def _check_vps(got, o, tag):
    pan, segs, ids = got
    assert [s["id"] for s in segs] == o[f"{tag}_seg_id"].tolist(), tag
    assert [s["category_id"] for s in segs] == o[f"{tag}_seg_cat"].tolist(), tag
    assert [s["isthing"] for s in segs] == o[f"{tag}_seg_isthing"].tolist(), tag
    assert list(ids) == o[f"{tag}_ids"].tolist(), tag
    assert torch.equal(pan.to(torch.int64), o[f"{tag}_masks"].to(torch.int64)), tag


def test_oracle_composition_offline_equals_reference_forward():
    from oracle import dvis_torch as O
    g, cfg, sd, bb, frames, kw = _g10()
    o = g.outs
    with torch.no_grad():
        st = {}
        got = O.dvis_plus_forward(sd, bb, frames, offline=True, task="vps", stages=st, **kw)
        torch.testing.assert_close(st["refiner_logits"], o["off_refiner_logits"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(st["masks"][None], o["off_refiner_masks"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(st["refiner_embds"], o["off_refiner_embds"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(st["online_logits"], o["off_online_logits"], rtol=1e-4, atol=1e-5)
        assert len(got[1]) >= 2                                            # non-degenerate: several segments survive
        _check_vps(got, o, "off_vps")
        scores, labels, qidx, masks = O.dvis_plus_forward(sd, bb, frames, offline=True, task="vis", **kw)
        torch.testing.assert_close(scores, o["off_vis_scores"], rtol=1e-4, atol=1e-6)
        assert torch.equal(labels, o["off_vis_labels"]) and torch.equal(qidx, o["off_vis_ids"])
        assert torch.equal(masks, o["off_vis_masks"])
        sem = O.dvis_plus_forward(sd, bb, frames, offline=True, task="vss", **kw)
        assert torch.equal(sem, o["off_vss_masks"].to(sem.dtype))
        # offline + keep: the window loop ignores it (meta_architecture.py:1479-1486)
        st = {}
        O.dvis_plus_forward(sd, bb, frames[:4], offline=True, task="vps", stages=st, **kw)
        got = O.dvis_plus_forward(sd, bb, frames[4:], offline=True, task="vps", keep=True, tracker=st["tracker"], **kw)
        _check_vps(got, o, "off_keep_vps")


def test_oracle_composition_online_equals_reference_forward():
    from oracle import dvis_torch as O
    g, cfg, sd, bb, frames, kw = _g10()
    o = g.outs
    with torch.no_grad():
        st = {}
        got = O.dvis_plus_forward(sd, bb, frames, offline=False, task="vps", stages=st, **kw)
        torch.testing.assert_close(st["online_logits"], o["on_logits"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(st["masks"][None], o["on_masks"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(st["instance_embds"], o["on_embds"], rtol=1e-4, atol=1e-5)
        _check_vps(got, o, "on_vps")
        scores, labels, qidx, masks = O.dvis_plus_forward(sd, bb, frames, offline=False, task="vis", **kw)
        torch.testing.assert_close(scores, o["on_vis_scores"], rtol=1e-4, atol=1e-6)
        assert torch.equal(labels, o["on_vis_labels"]) and torch.equal(qidx, o["on_vis_ids"])
        assert torch.equal(masks, o["on_vis_masks"])
        # online + keep: frames 0..3, then 4..6 resuming the tracker (meta_architecture.py:793)
        st = {}
        _check_vps(O.dvis_plus_forward(sd, bb, frames[:4], offline=False, task="vps", stages=st, **kw), o, "on_keep_a_vps")
        got = O.dvis_plus_forward(sd, bb, frames[4:], offline=False, task="vps", keep=True, tracker=st["tracker"], **kw)
        _check_vps(got, o, "on_keep_b_vps")
