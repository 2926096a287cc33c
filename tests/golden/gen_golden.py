"""Generate golden vectors from the IMPORTED reference (build container only).

    cd tests/golden && python gen_golden.py

Runs the reference's own Python (fp32/fp64 CPU path, the parity target named by
BASELINE.json) on small seeded inputs and writes ``*.npz`` fixtures next to
this script.  Fixtures hold data only: inputs, the module ``state_dict`` and the
reference outputs.  Nothing here is needed (or importable) at test time.

Versions are recorded in each fixture (torch / scipy / numpy): the third-party
arithmetic on the path (MultiheadAttention, grid_sample, interpolate,
linear_sum_assignment) is defined by these goldens.
"""
import os
import sys
import types

import numpy as np
import scipy
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _ref_import as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
META = dict(torch=torch.__version__, scipy=scipy.__version__, numpy=np.__version__)


def _np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def _small_int(t):
    """Integer maps (segment ids, class indices) as int16: values are tiny, fixtures stay small."""
    a = _np(t)
    assert a.min() >= -32768 and a.max() <= 32767
    return a.astype(np.int16)


def save(name, ins=None, outs=None, sd=None, **meta):
    d = {}
    for k, v in (ins or {}).items():
        d["in/" + k] = _np(v)
    for k, v in (outs or {}).items():
        d["out/" + k] = _np(v)
    for k, v in (sd or {}).items():
        d["sd/" + k] = _np(v)
    m = dict(META)
    m.update(meta)
    d["meta"] = np.array(repr(m))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def level_tensors(shapes):
    shapes_t = torch.as_tensor(shapes, dtype=torch.long)
    lsi = torch.cat((shapes_t.new_zeros((1,)), shapes_t.prod(1).cumsum(0)[:-1]))
    return shapes_t, lsi


# --------------------------------------------------------------------------- G1
def g1_msda():
    F_ = R.ref("mask2former.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    core = F_.ms_deform_attn_core_pytorch

    # (1) the reference's own test recipe (ops/test.py:19-39): seed 3, tiny shapes,
    # fp64 and fp32.  Its vectors come from the CUDA RNG and cannot be regenerated
    # on CPU; the recipe is replayed with the CPU RNG instead.
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes, lsi = level_tensors([(6, 4), (3, 2)])
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        w = torch.rand(N, Lq, M, L, P) + 1e-5
        w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        out = core(value.to(dt), shapes, loc.to(dt), w.to(dt))
        save(f"g1_msda_reftest_{tag}",
             ins=dict(value=value.to(dt), shapes=shapes, level_start=lsi, loc=loc.to(dt), w=w.to(dt)),
             outs=dict(out=out), seed=3, recipe="ops/test.py")

    # (2) realistic layouts incl. out-of-range locations (zero padding, border corners)
    cases = [
        ("r50", 2, 8, 32, 3, 4, [(2, 3), (4, 6), (8, 12)], None, torch.float32),
        ("r50_f64", 1, 8, 32, 3, 4, [(3, 5), (6, 10), (12, 20)], 40, torch.float64),
        ("vitl", 1, 16, 64, 1, 4, [(5, 7)], 50, torch.float32),
        ("d30", 1, 2, 30, 2, 2, [(4, 5), (2, 3)], 7, torch.float64),
        ("d71", 1, 2, 71, 2, 3, [(4, 5), (2, 3)], 5, torch.float64),
        ("d64", 2, 3, 64, 2, 4, [(4, 6), (2, 3)], 9, torch.float32),
        ("l4", 1, 4, 32, 4, 4, [(2, 3), (4, 6), (8, 12), (16, 24)], 33, torch.float32),
    ]
    for i, (tag, N, M, D, L, P, shp, Lq, dt) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + i)
        shapes, lsi = level_tensors(shp)
        S = int(shapes.prod(1).sum())
        Lq = S if Lq is None else Lq
        value = torch.randn(N, S, M, D, generator=g, dtype=torch.float64).to(dt)
        loc = (torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64) * 1.4 - 0.2).to(dt)
        # a few exact-border / far-outside locations
        loc[0, 0, 0, 0, 0] = torch.tensor([0.0, 0.0], dtype=dt)
        loc[0, 0, 0, 0, 1] = torch.tensor([1.0, 1.0], dtype=dt)
        loc[0, -1, -1, -1, 0] = torch.tensor([-3.0, 0.5], dtype=dt)
        loc[0, -1, -1, -1, 1] = torch.tensor([0.5, 7.0], dtype=dt)
        w = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64)
        w = (w / w.flatten(-2).sum(-1)[..., None, None]).to(dt)
        value.requires_grad_(True), loc.requires_grad_(True), w.requires_grad_(True)
        out = core(value, shapes, loc, w)
        go = torch.randn(out.shape, generator=g, dtype=torch.float64).to(dt)
        gv, gl, gw = torch.autograd.grad(out, (value, loc, w), go)
        save(f"g1_msda_{tag}",
             ins=dict(value=value, shapes=shapes, level_start=lsi, loc=loc, w=w, grad_out=go),
             outs=dict(out=out, grad_value=gv, grad_loc=gl, grad_w=gw), seed=100 + i)


# --------------------------------------------------------------------------- G2
def g2_pixel_decoder(conv_dim=32, name="g2_pixel_decoder", seed=20):
    m = R.ref("mask2former.modeling.pixel_decoder.msdeformattn")
    SS = sys.modules["detectron2.layers"].ShapeSpec
    torch.manual_seed(seed)
    chans = dict(res2=8, res3=12, res4=16, res5=20)
    strides = dict(res2=4, res3=8, res4=16, res5=32)
    inp = {k: SS(channels=chans[k], stride=strides[k]) for k in chans}
    pd = m.MSDeformAttnPixelDecoder(
        inp, transformer_dropout=0.0, transformer_nheads=2, transformer_dim_feedforward=64,
        transformer_enc_layers=2, conv_dim=conv_dim, mask_dim=16, norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    # non-trivial offsets/weights (the reference initialises their weights to 0)
    with torch.no_grad():
        for layer in pd.transformer.encoder.layers:
            layer.self_attn.sampling_offsets.weight.normal_(0, 0.3)
            layer.self_attn.attention_weights.weight.normal_(0, 0.5)
            layer.self_attn.attention_weights.bias.normal_(0, 0.5)
    H, W = 64, 96
    feats = {k: torch.randn(2, chans[k], H // strides[k], W // strides[k]) for k in chans}
    with torch.no_grad():
        mf, out0, ms = pd.forward_features(feats)
        # the MSDeformAttn module alone (first encoder layer) on its own inputs
        attn = pd.transformer.encoder.layers[0].self_attn
        shapes, lsi = level_tensors([(2, 3), (4, 6), (8, 12)])
        S = int(shapes.prod(1).sum())
        q = torch.randn(2, S, conv_dim)
        src = torch.randn(2, S, conv_dim)
        vr = torch.ones(2, 3, 2)
        refp = m.MSDeformAttnTransformerEncoder.get_reference_points(shapes, vr, "cpu")
        attn_out = attn(q, refp, src, shapes, lsi, None)
    save(name,
         ins=dict(**{f"feat_{k}": v for k, v in feats.items()}, attn_query=q, attn_src=src, attn_ref=refp),
         outs=dict(mask_features=mf, out0=out0, ms0=ms[0], ms1=ms[1], ms2=ms[2], attn_out=attn_out),
         sd=pd.state_dict(), seed=seed,
         cfg=dict(conv_dim=conv_dim, mask_dim=16, nheads=2, ffn=64, enc_layers=2, chans=chans))


# --------------------------------------------------------------------------- G3
def g3_decoder(hid=32, suffix="", seed=30):
    d = R.ref("dvis_Plus.video_mask2former_transformer_decoder")
    torch.manual_seed(seed)
    K = 7
    dec = d.VideoMultiScaleMaskedTransformerDecoder_dvisPlus(
        hid, True, num_classes=K, hidden_dim=hid, num_queries=6, nheads=2, dim_feedforward=64,
        dec_layers=3, pre_norm=False, mask_dim=16, enforce_input_project=False, num_frames=2,
        num_reid_head_layers=3, reid_hidden_dim=hid).eval()
    x = [torch.randn(2, hid, 2, 3), torch.randn(2, hid, 4, 6), torch.randn(2, hid, 8, 12)]
    mf = torch.randn(2, 16, 16, 24)
    with torch.no_grad():
        out = dec(x, mf)
    outs = {k: out[k] for k in ("pred_logits", "pred_masks", "pred_embds", "pred_embds_without_norm",
                                "pred_reid_embed")}
    for i, a in enumerate(out["aux_outputs"]):
        outs[f"aux{i}_logits"] = a["pred_logits"]
        outs[f"aux{i}_masks"] = a["pred_masks"]
    save("g3_decoder_dvisplus" + suffix, ins=dict(x0=x[0], x1=x[1], x2=x[2], mask_features=mf), outs=outs,
         sd=dec.state_dict(), seed=seed,
         cfg=dict(num_classes=K, hidden=hid, Q=6, nheads=2, ffn=64, dec_layers=3, mask_dim=16))

    # image decoder of BASELINE config #1 (mask2former_transformer_decoder.py:363-448)
    im = R.ref("mask2former.modeling.transformer_decoder.mask2former_transformer_decoder")
    torch.manual_seed(seed + 1)
    dec2 = im.MultiScaleMaskedTransformerDecoder(
        hid, True, num_classes=K, hidden_dim=hid, num_queries=6, nheads=2, dim_feedforward=64,
        dec_layers=3, pre_norm=False, mask_dim=16, enforce_input_project=False).eval()
    with torch.no_grad():
        out2 = dec2(x, mf)
    save("g3_decoder_image" + suffix, ins=dict(x0=x[0], x1=x[1], x2=x[2], mask_features=mf),
         outs=dict(pred_logits=out2["pred_logits"], pred_masks=out2["pred_masks"]),
         sd=dec2.state_dict(), seed=seed + 1, cfg=dict(num_classes=K, hidden=hid, Q=6, nheads=2, ffn=64, dec_layers=3))


# --------------------------------------------------------------------------- G4
def g4_tracker_refiner():
    t = R.ref("dvis_Plus.tracker")
    r = R.ref("dvis_Plus.refiner")
    torch.manual_seed(40)
    K, C, Q, MD = 7, 64, 6, 16
    trk = t.ReferringTracker_noiser(hidden_channel=C, feedforward_channel=128, num_head=2,
                                    decoder_layer_num=3, noise_mode="wa", mask_dim=MD, class_num=K).eval()
    T1, T2 = 3, 2
    fe = torch.randn(1, C, T1 + T2, Q)
    # make consecutive frames correlated but permuted so the Hungarian step is non-trivial
    perm = torch.stack([torch.randperm(Q) for _ in range(T1 + T2)])
    base = torch.randn(C, Q)
    for i in range(T1 + T2):
        fe[0, :, i] = base[:, perm[i]] + 0.3 * torch.randn(C, Q)
    fe_nn = fe * 1.7 + 0.1 * torch.randn_like(fe)
    mf = torch.randn(1, T1 + T2, MD, 8, 12)
    with torch.no_grad():
        o1, idx1 = trk(fe[:, :, :T1], mf[:, :T1], resume=False, return_indices=True,
                       frame_embeds_no_norm=fe_nn[:, :, :T1])
        o2, idx2 = trk(fe[:, :, T1:], mf[:, T1:], resume=True, return_indices=True,
                       frame_embeds_no_norm=fe_nn[:, :, T1:])
    outs = {}
    for tag, o, idx in (("a", o1, idx1), ("b", o2, idx2)):
        for k in ("pred_logits", "pred_masks", "pred_embds", "pred_references"):
            outs[f"{tag}_{k}"] = o[k]
        outs[f"{tag}_indices"] = np.stack([np.asarray(x) for x in idx]).astype(np.int64)
    save("g4_tracker", ins=dict(frame_embeds=fe, frame_embeds_no_norm=fe_nn, mask_features=mf),
         outs=outs, sd=trk.state_dict(), seed=40,
         cfg=dict(K=K, C=C, Q=Q, mask_dim=MD, heads=2, ffn=128, layers=3, T1=T1, T2=T2))

    torch.manual_seed(41)
    ref = r.TemporalRefiner(hidden_channel=C, feedforward_channel=128, num_head=2, decoder_layer_num=2,
                            mask_dim=MD, class_num=K, windows=2).eval()
    T = T1 + T2
    inst = torch.cat([o1["pred_embds"], o2["pred_embds"]], dim=2)
    with torch.no_grad():
        ro = ref(inst, fe_nn, mf)
    save("g4_refiner", ins=dict(instance_embeds=inst, frame_embeds=fe_nn, mask_features=mf),
         outs=dict(pred_logits=ro["pred_logits"], pred_masks=ro["pred_masks"], pred_embds=ro["pred_embds"]),
         sd=ref.state_dict(), seed=41,
         cfg=dict(K=K, C=C, Q=Q, mask_dim=MD, heads=2, ffn=128, layers=2, T=T, windows=2))


# --------------------------------------------------------------------------- G5
def g5_match():
    n = R.ref("dvis_Plus.noiser")
    noiser = n.Noiser(noise_ratio=0.5, mode="wa")
    g = torch.Generator().manual_seed(50)
    ins, outs = {}, {}
    for i, (Q, C, kind) in enumerate([(6, 16, "rand"), (25, 64, "rand"), (100, 512, "perm"),
                                      (100, 512, "rand"), (12, 8, "dup"), (9, 4, "nan")]):
        ref = torch.randn(Q, 1, C, generator=g)
        if kind == "perm":
            p = torch.randperm(Q, generator=g)
            cur = ref[p] + 0.2 * torch.randn(Q, 1, C, generator=g)
        elif kind == "dup":  # duplicated rows -> tied costs
            cur = torch.randn(Q, 1, C, generator=g)
            cur[1] = cur[0]
            cur[5] = cur[4]
            ref[3] = ref[2]
        elif kind == "nan":  # zero vectors: 0/(0+1e-6) -> 0 similarity, cost 1 everywhere for that row
            cur = torch.randn(Q, 1, C, generator=g)
            cur[2] = 0
            ref[7] = 0
        else:
            cur = torch.randn(Q, 1, C, generator=g)
        idx = noiser.match_embds(ref, cur)
        ins[f"ref{i}"], ins[f"cur{i}"] = ref, cur
        outs[f"idx{i}"] = np.asarray(idx).astype(np.int64)
    save("g5_match", ins=ins, outs=outs, seed=50, ncases=6)


# --------------------------------------------------------------------------- G6
def g6_postprocess():
    """inference_video_{vis,vps,vss} + post_processing on hand-made decoder outputs.  Two fixtures: the original small
    one (every resize has output height + width <= 128: torch's "channels-last" bilinear kernel, and a crop in x that
    leaves scalar-exp tail columns in the sigmoid) and a larger one whose resizes take torch's generic separable kernel
    like every real frame size does."""
    _g6("g6_postprocess", 60, T=3, hw=(10, 14), img_size=(37, 53), out_hw=(30, 45), first=(40, 56))
    _g6("g6_postprocess_large", 61, T=2, hw=(16, 24), img_size=(61, 90), out_hw=(75, 110), first=(64, 96))


def _g6(name, seed, T, hw, img_size, out_hw, first):
    ma = R.ref_meta()
    cls = ma.DVIS_Plus_offline
    K, Q = 5, 8
    h, w = hw
    g = torch.Generator().manual_seed(seed)
    stub = types.SimpleNamespace(
        sem_seg_head=types.SimpleNamespace(num_classes=K), num_queries=Q, max_num=4, device="cpu",
        object_mask_threshold=0.8, overlap_threshold=0.8,
        metadata=types.SimpleNamespace(thing_dataset_id_to_contiguous_id={0: 0, 1: 1, 2: 2}))
    pred_logits = torch.randn(1, T, Q, K + 1, generator=g) * 2
    pred_logits[0, :, 0, 1] += 9
    pred_logits[0, :, 1, 4] += 9  # stuff class
    pred_logits[0, :, 2, 4] += 9  # same stuff class -> merged
    pred_logits[0, :, 3, 0] += 9
    pred_logits[0, :, 5, K] += 9  # void
    aux_logits = torch.randn(1, T, Q, K + 1, generator=g) * 2
    masks = torch.randn(1, Q, T, h, w, generator=g) * 3
    # structured masks for the confident queries so VPS keeps real segments (thing, thing, merged stuff)
    masks[0, :4] = -6 + torch.randn(4, T, h, w, generator=g)
    masks[0, 0, :, 0:h // 2, 0:w // 2] += 12            # thing, class 1
    masks[0, 1, :, 0:h // 2, w // 2:w] += 12            # stuff class 4
    masks[0, 2, :, h // 2:h, 0:w // 2 - 1] += 12        # stuff class 4 again -> merged into the same segment id
    masks[0, 3, :, h // 2:h, w // 2 - 1:w] += 12        # thing, class 0
    masks[0, 3, 1, 0:3, 0:3] += 12                      # overlaps query 0 in frame 1
    outputs = dict(pred_logits=pred_logits.clone(), pred_masks=masks.clone())
    outputs, aux = cls.post_processing(stub, outputs, aux_logits=aux_logits.clone())
    mask_cls, mask_pred, pid = outputs["pred_logits"][0], outputs["pred_masks"][0], outputs["ids"][0]
    outs = dict(pp_logits=outputs["pred_logits"], pp_aux=aux)
    vis = cls.inference_video_vis(stub, mask_cls.clone(), mask_pred.clone(), img_size, *out_hw, first, pid,
                                  aux_pred_cls=aux.clone())
    outs.update(vis_scores=np.array(vis["pred_scores"], dtype=np.float32),
                vis_labels=np.array(vis["pred_labels"], dtype=np.int64),
                vis_ids=np.array(vis["pred_ids"], dtype=np.int64),
                vis_masks=torch.stack(vis["pred_masks"]))
    vps = cls.inference_video_vps(stub, mask_cls.clone(), mask_pred.clone(), img_size, *out_hw, first, pid,
                                  aux_pred_cls=aux.clone())
    outs.update(vps_masks=vps["pred_masks"],
                vps_ids=np.array([int(x) for x in vps["pred_ids"]], dtype=np.int64),
                vps_seg_id=np.array([s["id"] for s in vps["segments_infos"]], dtype=np.int64),
                vps_seg_isthing=np.array([s["isthing"] for s in vps["segments_infos"]], dtype=np.bool_),
                vps_seg_cat=np.array([s["category_id"] for s in vps["segments_infos"]], dtype=np.int64))
    vss = cls.inference_video_vss(stub, mask_cls.clone(), mask_pred.clone(), img_size, *out_hw, first, pid,
                                  aux_pred_cls=aux.clone())
    outs.update(vss_masks=vss["pred_masks"])
    labels = cls._get_instance_labels(stub, pred_logits.clone())
    outs.update(instance_labels=labels)
    save(name, ins=dict(pred_logits=pred_logits, aux_logits=aux_logits, pred_masks=masks),
         outs=outs, seed=seed,
         cfg=dict(K=K, Q=Q, T=T, max_num=4, object_mask_threshold=0.8, overlap_threshold=0.8, n_things=3,
                  img_size=img_size, out_hw=out_hw, first_resize=first))


def g9_minvis():
    """MinVIS.post_processing (frame-by-frame Hungarian alignment, meta_architecture.py:255-301) + inference_video
    (top-10, :362-407) on random decoder outputs with permuted, noisy embeddings."""
    ma = R.ref_meta()
    cls = ma.MinVIS
    K, Q, T, C = 6, 14, 5, 16
    g = torch.Generator().manual_seed(90)
    stub = types.SimpleNamespace(sem_seg_head=types.SimpleNamespace(num_classes=K), num_queries=Q, device="cpu")
    stub.match_from_embds = types.MethodType(cls.match_from_embds, stub)
    base = torch.randn(C, Q, generator=g)
    embds = torch.stack([base[:, torch.randperm(Q, generator=g)] + 0.25 * torch.randn(C, Q, generator=g)
                         for _ in range(T)], 1)[None]                                   # (1, C, T, Q)
    logits = torch.randn(1, T, Q, K + 1, generator=g) * 2
    masks = torch.randn(1, Q, T, 10, 14, generator=g) * 3
    out = cls.post_processing(stub, dict(pred_logits=logits.clone(), pred_masks=masks.clone(), pred_embds=embds.clone()))
    img_size, out_hw, first = (37, 53), (30, 45), (40, 56)
    vid = cls.inference_video(stub, out["pred_logits"][0].clone(), out["pred_masks"][0].clone(), img_size, *out_hw, first)
    save("g9_minvis", ins=dict(pred_logits=logits, pred_masks=masks, pred_embds=embds),
         outs=dict(pp_logits=out["pred_logits"], pp_masks=out["pred_masks"],
                   scores=np.array(vid["pred_scores"], dtype=np.float32),
                   labels=np.array(vid["pred_labels"], dtype=np.int64), masks=torch.stack(vid["pred_masks"])),
         seed=90, cfg=dict(K=K, Q=Q, T=T, C=C, img_size=img_size, out_hw=out_hw, first_resize=first, topk=10))


def g7_head_dim_32():
    """g2 / g3 at conv_dim = hidden = 64 with 2 heads (head dim 32): the narrowest width the HIP kernels serve, so the
    GPU tests can compare the product against the reference's outputs directly (g4 already has head dim 32)."""
    g2_pixel_decoder(conv_dim=64, name="g7_pixel_decoder_d32", seed=70)
    g3_decoder(hid=64, suffix="_d32", seed=71)


def g8_vit_adapter():
    """DINOv2 ViT + ViT-Adapter (backbones_vitAdapter/adapter.py:422-586, backbones.py:36-260) at a tiny width with head
    dim 32: embed 64, 2 heads, depth 4 (one block per interaction stage), deformable heads 2 (D = 32), input 64x96."""
    from functools import partial
    A = R.ref_vit_adapter()
    B = sys.modules["mask2former.modeling.backbones_vitAdapter.backbones"]
    L = sys.modules["mask2former.modeling.backbones_vitAdapter.layers"]
    torch.manual_seed(80)
    vit = B.DinoVisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=4, num_heads=2, mlp_ratio=4,
                                  block_fn=partial(L.NestedTensorBlock, attn_class=L.MemEffAttention),
                                  init_values=0.5, ffn_layer="mlp", block_chunks=0, qkv_bias=True, proj_bias=True,
                                  ffn_bias=True)
    ad = A.DinoV2ViTAdapter(vit_module=vit, pretrain_size=64, conv_inplane=8, n_points=4, deform_num_heads=2,
                            init_values=1e-6, interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]], with_cffn=True,
                            cffn_ratio=0.25, deform_ratio=0.5, add_vit_feature=True, use_extra_extractor=True).eval()
    with torch.no_grad():
        for m in ad.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.normal_(1, 0.2)
                m.bias.normal_(0, 0.2)
            if type(m).__name__ == "LayerScale":
                m.gamma.normal_(0.5, 0.2)
            if type(m).__name__ == "MSDeformAttn":
                m.sampling_offsets.weight.normal_(0, 0.3)
                m.attention_weights.weight.normal_(0, 0.5)
                m.attention_weights.bias.normal_(0, 0.5)
            if isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.normal_(0, 0.1)
        vit.cls_token.normal_(0, 0.5)
        vit.pos_embed.normal_(0, 0.5)
    x = torch.randn(2, 3, 64, 96)
    with torch.no_grad():
        f1, f2, f3, f4 = ad(x)
        tok, H, W = vit.prepare_tokens_with_masks(x, masks=None, return_HW=True)
        blk0 = vit.blocks[0](tok)
    save("g8_vit_adapter", ins=dict(x=x), outs=dict(f1=f1, f2=f2, f3=f3, f4=f4, tokens=tok, block0=blk0),
         sd=ad.state_dict(), seed=80,
         cfg=dict(embed=64, heads=2, depth=4, patch=16, img_size=64, conv_inplane=8, deform_heads=2, n_points=4,
                  interaction_indexes=[[0, 0], [1, 1], [2, 2], [3, 3]], cffn_ratio=0.25, HW=[int(H), int(W)]))


# --------------------------------------------------------------------------- G10
class _ImageList:
    """detectron2.structures.ImageList.from_tensors — un-vendored third-party semantics the reference relies on at
    meta_architecture.py:1311: stack to the max H, W rounded UP to a multiple of size_divisibility,
    zero padding bottom / right, image_sizes = the un-padded sizes."""

    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    @staticmethod
    def from_tensors(tensors, size_divisibility=0):
        sizes = [tuple(t.shape[-2:]) for t in tensors]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            H, W = (H + d - 1) // d * d, (W + d - 1) // d * d
        out = tensors[0].new_zeros((len(tensors), tensors[0].shape[0], H, W))
        for i, t in enumerate(tensors):
            out[i, :, :t.shape[-2], :t.shape[-1]] = t
        return _ImageList(out, sizes)


def g10_window_loop():
    """The a12 COMPOSITION through the reference's own methods, called unbound on a stub `self` that holds reference
    sub-modules (MaskFormerHead with MSDeformAttnPixelDecoder + dvisPlus decoder, ReferringTracker_noiser,
    TemporalRefiner) and a toy backbone:
      DVIS_Plus_offline.forward eval branch (meta_architecture.py:1301-1317, 1376-1396) -> run_window_inference
        (:1446-1500) -> post_processing (:758-772) -> inference_video_{vps,vis,vss};
      DVIS_Plus_online.forward eval branch (:629-642, 687-706) -> run_window_inference (:774-816), incl. a second call
        with `keep` (the demo's long-video hand-off; the OFFLINE window loop ignores `keep`: pinned too).
    T = 7 frames of 70 x 100 (padded to 96 x 128), window 3 (ragged last window 3 + 3 + 1), output size 105 x 150."""
    import importlib
    sys.path.insert(0, os.path.dirname(OUT))
    from toy_backbone import ToyBackbone, CHANS, STRIDES
    ma = R.ref_meta()
    sys.modules["detectron2.layers"].DeformConv = None            # un-vendored name imported by pixel_decoder/fpn.py:14
    if "mask2former.modeling.meta_arch" not in sys.modules:
        pkg = types.ModuleType("mask2former.modeling.meta_arch")
        pkg.__path__ = [f"{R.REF}/mask2former/modeling/meta_arch"]
        sys.modules[pkg.__name__] = pkg
    head_mod = importlib.import_module("mask2former.modeling.meta_arch.mask_former_head")
    pdm = R.ref("mask2former.modeling.pixel_decoder.msdeformattn")
    dm = R.ref("dvis_Plus.video_mask2former_transformer_decoder")
    tm, rm = R.ref("dvis_Plus.tracker"), R.ref("dvis_Plus.refiner")
    SS = sys.modules["detectron2.layers"].ShapeSpec
    ma.ImageList = _ImageList
    torch.manual_seed(100)
    K, Q, HID, MD, NH, NHT = 7, 6, 32, 16, 1, 2       # segmenter: 1 head of 32; tracker / refiner: 2 heads of 32
    cfg = dict(K=K, Q=Q, hidden=HID, mask_dim=MD, nheads=NH, trk_heads=NHT, enc_layers=2, enc_ffn=64, dec_layers=3, dec_ffn=64,
               tracker_layers=2, refiner_layers=2, trk_ffn=128, T=7, window=3, frame_hw=(70, 100), out_hw=(105, 150),
               n_things=3, max_num=5, overlap_threshold=0.3)
    backbone = ToyBackbone().eval()
    inp = {k: SS(channels=CHANS[k], stride=STRIDES[k]) for k in CHANS}
    pd = pdm.MSDeformAttnPixelDecoder(inp, transformer_dropout=0.0, transformer_nheads=NH, transformer_dim_feedforward=64,
                                      transformer_enc_layers=2, conv_dim=HID, mask_dim=MD, norm="GN",
                                      transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    with torch.no_grad():
        for layer in pd.transformer.encoder.layers:
            layer.self_attn.sampling_offsets.weight.normal_(0, 0.3)
            layer.self_attn.attention_weights.weight.normal_(0, 0.5)
    dec = dm.VideoMultiScaleMaskedTransformerDecoder_dvisPlus(
        HID, True, num_classes=K, hidden_dim=HID, num_queries=Q, nheads=NH, dim_feedforward=64, dec_layers=3,
        pre_norm=False, mask_dim=MD, enforce_input_project=False, num_frames=3, num_reid_head_layers=3,
        reid_hidden_dim=HID)
    head = head_mod.MaskFormerHead(inp, num_classes=K, pixel_decoder=pd, loss_weight=1.0, ignore_value=-1,
                                   transformer_predictor=dec, transformer_in_feature="multi_scale_pixel_decoder").eval()
    trk = tm.ReferringTracker_noiser(hidden_channel=2 * HID, feedforward_channel=128, num_head=NHT, decoder_layer_num=2,
                                     noise_mode="wa", mask_dim=MD, class_num=K).eval()
    ref = rm.TemporalRefiner(hidden_channel=2 * HID, feedforward_channel=128, num_head=NHT, decoder_layer_num=2,
                             mask_dim=MD, class_num=K, windows=3).eval()
    with torch.no_grad():                                 # decisive masks / class scores (random init ties everything)
        for m_ in (dec.mask_embed, trk.mask_embed, ref.mask_embed):
            m_.layers[-1].weight.mul_(30)
        for m_ in (dec.class_embed, trk.class_embed, ref.class_embed):
            m_.weight.mul_(8)
    g = torch.Generator().manual_seed(101)
    H, W = cfg["frame_hw"]
    yy, xx = torch.meshgrid(torch.linspace(0, 6.28, H), torch.linspace(0, 6.28, W), indexing="ij")
    frames = []
    for t in range(cfg["T"]):
        pat = 127 + 100 * torch.sin(xx * (1 + t % 2) + 0.3 * t) * torch.cos(yy * 2 + 0.2 * t)
        frames.append((torch.randint(0, 256, (3, H, W), generator=g).float() * 0.4 + pat[None] * 0.6).to(torch.uint8))
    frames = torch.stack(frames)
    pixel_mean = torch.tensor([123.675, 116.280, 103.530]).view(-1, 1, 1)
    pixel_std = torch.tensor([58.395, 57.120, 57.375]).view(-1, 1, 1)

    def stub_for(cls, task, thr):
        s_ = types.SimpleNamespace(
            backbone=backbone, sem_seg_head=head, tracker=trk, refiner=ref, keep=False, device=torch.device("cpu"),
            pixel_mean=pixel_mean, pixel_std=pixel_std, size_divisibility=32, training=False, window_inference=True,
            window_size=cfg["window"], num_queries=Q, max_num=cfg["max_num"], task=task, object_mask_threshold=thr,
            overlap_threshold=cfg["overlap_threshold"],
            metadata=types.SimpleNamespace(thing_dataset_id_to_contiguous_id={i: i for i in range(cfg["n_things"])}))
        for name in ("run_window_inference", "post_processing", "_get_instance_labels", "inference_video_vis",
                     "inference_video_vps", "inference_video_vss"):
            setattr(s_, name, types.MethodType(getattr(cls, name), s_))
        s_.inference_video_task = getattr(s_, "inference_video_" + task)
        return s_

    def video(lo, hi, keep=None):
        v = {"image": [f for f in frames[lo:hi]], "height": cfg["out_hw"][0], "width": cfg["out_hw"][1]}
        if keep is not None:
            v["keep"] = keep
        return v

    outs = {}

    def put(tag, r, task):
        if task == "vps":
            outs[f"{tag}_masks"] = _small_int(r["pred_masks"])
            outs[f"{tag}_ids"] = np.array([int(x) for x in r["pred_ids"]], dtype=np.int64)
            outs[f"{tag}_seg_id"] = np.array([s_["id"] for s_ in r["segments_infos"]], dtype=np.int64)
            outs[f"{tag}_seg_isthing"] = np.array([s_["isthing"] for s_ in r["segments_infos"]], dtype=np.bool_)
            outs[f"{tag}_seg_cat"] = np.array([s_["category_id"] for s_ in r["segments_infos"]], dtype=np.int64)
        elif task == "vis":
            outs[f"{tag}_scores"] = np.array(r["pred_scores"], dtype=np.float32)
            outs[f"{tag}_labels"] = np.array(r["pred_labels"], dtype=np.int64)
            outs[f"{tag}_ids"] = np.array(r["pred_ids"], dtype=np.int64)
            outs[f"{tag}_masks"] = torch.stack(r["pred_masks"])
        else:
            outs[f"{tag}_masks"] = _small_int(r["pred_masks"])

    T = cfg["T"]
    with torch.no_grad():
        # ---- offline: the floats behind the decisions, straight from run_window_inference
        st = stub_for(ma.DVIS_Plus_offline, "vps", 0.0)
        images = _ImageList.from_tensors([(f.float() - pixel_mean) / pixel_std for f in frames], 32)
        ro, online_logits = ma.DVIS_Plus_offline.run_window_inference(st, images.tensor, window_size=cfg["window"])
        outs.update(off_refiner_logits=ro["pred_logits"], off_refiner_masks=ro["pred_masks"],
                    off_refiner_embds=ro["pred_embds"], off_online_logits=online_logits)
        # a threshold that keeps about half of the non-void queries
        cls_, aux_ = ma.DVIS_Plus_offline.post_processing(st, dict(pred_logits=ro["pred_logits"].clone(),
                                                                   pred_masks=ro["pred_masks"]), aux_logits=online_logits.clone())
        pr = torch.softmax(cls_["pred_logits"][0], -1)
        pr[:, :-1] = torch.maximum(pr[:, :-1], torch.softmax(aux_, -1)[:, :-1])
        sc, lb = pr.max(-1)
        live = sc[lb != K].sort(descending=True)[0]
        keep_n = min(4, len(live) - 1)                                     # about 4 of the 6 queries become candidates
        thr = float((live[keep_n - 1] + live[keep_n]) / 2) if keep_n > 0 else 0.0
        cfg["object_mask_threshold"] = thr
        for task in ("vps", "vis", "vss"):
            st = stub_for(ma.DVIS_Plus_offline, task, thr)
            put(f"off_{task}", ma.DVIS_Plus_offline.forward(st, [video(0, T)]), task)
        # offline + keep: the window loop never resumes from `keep` (meta_architecture.py:1479-1486) — second half of the
        # clip with keep=True must equal the second half run on its own
        st = stub_for(ma.DVIS_Plus_offline, "vps", thr)
        ma.DVIS_Plus_offline.forward(st, [video(0, 4)])
        put("off_keep_vps", ma.DVIS_Plus_offline.forward(st, [video(4, T, keep=True)]), "vps")
        # ---- online
        st = stub_for(ma.DVIS_Plus_online, "vps", thr)
        on = ma.DVIS_Plus_online.run_window_inference(st, images.tensor, window_size=cfg["window"])
        outs.update(on_logits=on["pred_logits"], on_masks=on["pred_masks"], on_embds=on["pred_embds"])
        for task in ("vps", "vis"):
            st = stub_for(ma.DVIS_Plus_online, task, thr)
            put(f"on_{task}", ma.DVIS_Plus_online.forward(st, [video(0, T)]), task)
        # online + keep: frames 0..3, then 4..6 resuming the tracker state (demo_long_video.py:113-126)
        st = stub_for(ma.DVIS_Plus_online, "vps", thr)
        put("on_keep_a_vps", ma.DVIS_Plus_online.forward(st, [video(0, 4)]), "vps")
        put("on_keep_b_vps", ma.DVIS_Plus_online.forward(st, [video(4, T, keep=True)]), "vps")
    sd = {}
    for prefix, mod in (("backbone.", backbone), ("sem_seg_head.pixel_decoder.", pd), ("sem_seg_head.predictor.", dec),
                        ("tracker.", trk), ("refiner.", ref)):
        sd.update({prefix + k: v for k, v in mod.state_dict().items()})
    sd["pixel_mean"], sd["pixel_std"] = pixel_mean, pixel_std
    print("  g10: threshold %.4f, offline vps segments %s, online vps segments %s, keep-b segments %s" % (
        thr, outs["off_vps_seg_cat"].tolist(), outs["on_vps_seg_cat"].tolist(), outs["on_keep_b_vps_seg_cat"].tolist()))
    save("g10_window_loop", ins=dict(frames=frames), outs=outs, sd=sd, seed=100, cfg=cfg)



if __name__ == "__main__":
    import warnings
    warnings.filterwarnings("ignore")
    torch.set_num_threads(1)  # deterministic reduction order in the generating run
    only = sys.argv[1:]
    for fn in (g1_msda, g2_pixel_decoder, g3_decoder, g4_tracker_refiner, g5_match, g6_postprocess, g7_head_dim_32, g8_vit_adapter, g9_minvis, g10_window_loop):
        if not only or fn.__name__.split("_")[0] in only:
            fn()
