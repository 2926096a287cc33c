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
