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


if __name__ == "__main__":
    import warnings
    warnings.filterwarnings("ignore")
    torch.set_num_threads(1)  # deterministic reduction order in the generating run
    only = sys.argv[1:]
    for fn in (g1_msda, g2_pixel_decoder, g3_decoder, g4_tracker_refiner, g5_match, g6_postprocess, g7_head_dim_32, g8_vit_adapter, g9_minvis):
        if not only or fn.__name__.split("_")[0] in only:
            fn()
