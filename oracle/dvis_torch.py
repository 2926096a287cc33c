"""ORACLE — test infrastructure, not product code.

Plain-PyTorch (CPU, fp32/fp64) restatement of the reference's inference hot path, written as
functions over a ``state_dict`` that uses the REFERENCE'S parameter names, so weights captured from the
reference modules load without translation.  Each function cites the reference lines it follows
(paths under /root/reference/DVIS_Plus/).  Pinned in tests/test_oracle_model.py against
tests/golden/g2..g6 (generated from the imported reference by tests/golden/gen_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment   # third-party solver the reference calls (noiser.py:54)

from .msda import msda_forward_torch


# ----------------------------------------------------------------------------- small pieces
def linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"])


def mlp(sd, p, x, n):
    """MLP of mask2former_video/.../video_mask2former_transformer_decoder.py:193-206."""
    for i in range(n):
        x = linear(sd, f"{p}.layers.{i}", x)
        if i < n - 1:
            x = F.relu(x)
    return x


def mha(sd, p, q, k, v, nheads, attn_mask=None):
    """nn.MultiheadAttention.forward as the reference calls it (seq-first, need_weights default)."""
    C = q.shape[-1]
    return F.multi_head_attention_forward(
        q, k, v, C, nheads, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"], None, None, False, 0.0,
        sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"], training=False, key_padding_mask=None,
        need_weights=True, attn_mask=attn_mask)[0]


def self_attention_layer(sd, p, tgt, nheads, query_pos=None):
    """SelfAttentionLayer.forward_post, video_mask2former_transformer_decoder.py:40-51."""
    qk = tgt if query_pos is None else tgt + query_pos
    return layer_norm(sd, p + ".norm", tgt + mha(sd, p + ".self_attn", qk, qk, tgt, nheads))


def cross_attention_layer(sd, p, tgt, memory, nheads, memory_mask=None, pos=None, query_pos=None):
    """CrossAttentionLayer.forward_post, ibid. :99-111."""
    q = tgt if query_pos is None else tgt + query_pos
    k = memory if pos is None else memory + pos
    return layer_norm(sd, p + ".norm", tgt + mha(sd, p + ".multihead_attn", q, k, memory, nheads, memory_mask))


def referring_cross_attention_layer(sd, p, identity, tgt, key, memory, nheads):
    """ReferringCrossAttentionLayer.forward_post, dvis_Plus/tracker.py:35-53: residual from `identity`."""
    return layer_norm(sd, p + ".norm", identity + mha(sd, p + ".multihead_attn", tgt, key, memory, nheads))


def ffn_layer(sd, p, tgt):
    """FFNLayer.forward_post, video_mask2former_transformer_decoder.py:166-170."""
    return layer_norm(sd, p + ".norm", tgt + linear(sd, p + ".linear2", F.relu(linear(sd, p + ".linear1", tgt))))


def position_embedding_sine(n, h, w, num_pos_feats, dtype=torch.float32, temperature=10000.0):
    """PositionEmbeddingSine(normalize=True), mask2former/.../position_encoding.py:29-52 (mask all False)."""
    scale, eps = 2 * math.pi, 1e-6
    y_embed = torch.arange(1, h + 1, dtype=torch.float32)[None, :, None].expand(n, h, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32)[None, None, :].expand(n, h, w)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2).to(dtype)


# ----------------------------------------------------------------------------- pixel decoder (a3-a5)
def encoder_reference_points(shapes):
    """MSDeformAttnTransformerEncoder.get_reference_points with valid_ratios == 1, msdeformattn.py:141-153."""
    pts = []
    for (h, w) in shapes:
        ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, dtype=torch.float32),
                                      torch.linspace(0.5, w - 0.5, w, dtype=torch.float32), indexing="ij")
        pts.append(torch.stack((ref_x.reshape(-1) / w, ref_y.reshape(-1) / h), -1))
    ref = torch.cat(pts, 0)
    return ref[None, :, None, :].expand(1, -1, len(shapes), -1)       # (1, S, L, 2)


def ms_deform_attn_module(sd, p, query, reference_points, input_flatten, shapes, n_heads, n_points):
    """MSDeformAttn.forward, ops/modules/ms_deform_attn.py:82-125 (2-d reference points, no padding mask)."""
    N, Lq, C = query.shape
    L = len(shapes)
    value = linear(sd, p + ".value_proj", input_flatten).view(N, -1, n_heads, C // n_heads)
    off = linear(sd, p + ".sampling_offsets", query).view(N, Lq, n_heads, L, n_points, 2)
    aw = linear(sd, p + ".attention_weights", query).view(N, Lq, n_heads, L * n_points)
    aw = F.softmax(aw, -1).view(N, Lq, n_heads, L, n_points)
    st = torch.as_tensor(shapes, dtype=torch.long)
    normalizer = torch.stack([st[..., 1], st[..., 0]], -1)
    loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = msda_forward_torch(value, st, loc, aw)
    return linear(sd, p + ".output_proj", out)


def pixel_decoder_forward(sd, features, nheads, enc_layers, n_points=4, p=""):
    """MSDeformAttnPixelDecoder.forward_features, msdeformattn.py:314-358 (+ encoder :61-89, :122-131, :155-161).

    features: dict res2..res5 -> (N, C, H, W).  Returns (mask_features, out[0], multi_scale_features[:3]).
    Transformer levels are res5, res4, res3 (low-res first); one extra FPN level (res2).
    """
    names = ["res5", "res4", "res3"]
    C = sd[p + "input_proj.0.0.weight"].shape[0]
    srcs, pos, shapes = [], [], []
    for i, f in enumerate(names):
        x = features[f].float()
        y = F.conv2d(x, sd[p + f"input_proj.{i}.0.weight"], sd[p + f"input_proj.{i}.0.bias"])
        y = F.group_norm(y, 32, sd[p + f"input_proj.{i}.1.weight"], sd[p + f"input_proj.{i}.1.bias"])
        srcs.append(y)
        pos.append(position_embedding_sine(x.shape[0], x.shape[2], x.shape[3], C // 2))
        shapes.append((x.shape[2], x.shape[3]))
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    lvl_pos = torch.cat([pe.flatten(2).transpose(1, 2) + sd[p + "transformer.level_embed"][i].view(1, 1, -1)
                         for i, pe in enumerate(pos)], 1)
    ref = encoder_reference_points(shapes)
    out = src
    for i in range(enc_layers):
        lp = p + f"transformer.encoder.layers.{i}"
        src2 = ms_deform_attn_module(sd, lp + ".self_attn", out + lvl_pos, ref.expand(out.shape[0], -1, -1, -1), out,
                                     shapes, nheads, n_points)
        out = layer_norm(sd, lp + ".norm1", out + src2)
        out = layer_norm(sd, lp + ".norm2",
                         out + linear(sd, lp + ".linear2", F.relu(linear(sd, lp + ".linear1", out))))
    bs = out.shape[0]
    outs, start = [], 0
    for (h, w) in shapes:
        outs.append(out[:, start:start + h * w].transpose(1, 2).reshape(bs, -1, h, w))
        start += h * w
    # one FPN level on res2 (Conv2d(norm=GN) = conv -> norm -> activation; lateral has no activation)
    x = features["res2"].float()
    lat = F.group_norm(F.conv2d(x, sd[p + "adapter_1.weight"]), 32, sd[p + "adapter_1.norm.weight"],
                       sd[p + "adapter_1.norm.bias"])
    y = lat + F.interpolate(outs[-1], size=lat.shape[-2:], mode="bilinear", align_corners=False)
    y = F.relu(F.group_norm(F.conv2d(y, sd[p + "layer_1.weight"], padding=1), 32, sd[p + "layer_1.norm.weight"],
                            sd[p + "layer_1.norm.bias"]))
    outs.append(y)
    mask_features = F.conv2d(outs[-1], sd[p + "mask_features.weight"], sd[p + "mask_features.bias"])
    return mask_features, outs[0], outs[:3]


# ----------------------------------------------------------------------------- masked-attention decoder (a6-a8)
def prediction_heads(sd, p, output, mask_features, target_size, nheads):
    """forward_prediction_heads, dvis_Plus/video_mask2former_transformer_decoder.py:358-374."""
    dec = layer_norm(sd, p + "decoder_norm", output).transpose(0, 1)
    cls = linear(sd, p + "class_embed", dec)
    emb = mlp(sd, p + "mask_embed", dec, 3)
    masks = torch.einsum("bqc,bchw->bqhw", emb, mask_features)
    attn = F.interpolate(masks, size=target_size, mode="bilinear", align_corners=False)
    attn = (attn.sigmoid().flatten(2).unsqueeze(1).repeat(1, nheads, 1, 1).flatten(0, 1) < 0.5).bool()
    return cls, masks, attn


def decoder_forward(sd, x, mask_features, nheads, dec_layers, p="", dvis_plus=True, reid_layers=3):
    """VideoMultiScaleMaskedTransformerDecoder_dvisPlus.forward (eval, frames = batch), ibid. :258-356;
    with dvis_plus=False: image MultiScaleMaskedTransformerDecoder.forward,
    mask2former/.../mask2former_transformer_decoder.py:363-431 (same math, fewer outputs)."""
    C = sd[p + "query_feat.weight"].shape[1]
    src, pos, sizes = [], [], []
    for i in range(3):
        n, _, h, w = x[i].shape
        sizes.append((h, w))
        pe = position_embedding_sine(n, h, w, C // 2).flatten(2).permute(2, 0, 1)
        xi = x[i]
        if (p + f"input_proj.{i}.weight") in sd:
            xi = F.conv2d(xi, sd[p + f"input_proj.{i}.weight"], sd[p + f"input_proj.{i}.bias"])
        s = (xi.flatten(2) + sd[p + "level_embed.weight"][i][None, :, None]).permute(2, 0, 1)
        pos.append(pe)
        src.append(s)
    bs = src[0].shape[1]
    query_embed = sd[p + "query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    output = sd[p + "query_feat.weight"].unsqueeze(1).repeat(1, bs, 1)
    pred_cls, pred_mask, attn_masks = [], [], []
    c, m, attn = prediction_heads(sd, p, output, mask_features, sizes[0], nheads)
    pred_cls.append(c), pred_mask.append(m)
    for i in range(dec_layers):
        lvl = i % 3
        attn[torch.where(attn.sum(-1) == attn.shape[-1])] = False
        attn_masks.append(attn.clone())
        output = cross_attention_layer(sd, p + f"transformer_cross_attention_layers.{i}", output, src[lvl], nheads,
                                       memory_mask=attn, pos=pos[lvl], query_pos=query_embed)
        output = self_attention_layer(sd, p + f"transformer_self_attention_layers.{i}", output, nheads, query_embed)
        output = ffn_layer(sd, p + f"transformer_ffn_layers.{i}", output)
        c, m, attn = prediction_heads(sd, p, output, mask_features, sizes[(i + 1) % 3], nheads)
        pred_cls.append(c), pred_mask.append(m)
    if not dvis_plus:
        return dict(pred_logits=pred_cls[-1], pred_masks=pred_mask[-1], attn_masks=attn_masks)
    t = pred_mask[-1].shape[0]                                    # eval: bs = 1, all frames are one clip
    normed = layer_norm(sd, p + "decoder_norm", output)
    reid = mlp(sd, p + "reid_embed", normed, reid_layers) if reid_layers > 0 else normed
    to_bctq = lambda z: z.permute(2, 1, 0).unsqueeze(0)           # 'q (b t) c -> b c t q' with b = 1
    return dict(
        pred_logits=pred_cls[-1].unsqueeze(0),                                   # (1, t, q, K+1)
        pred_masks=pred_mask[-1].permute(1, 0, 2, 3).unsqueeze(0),               # (1, q, t, h, w)
        pred_embds=torch.cat([to_bctq(normed), to_bctq(reid)], dim=1),
        pred_embds_without_norm=torch.cat([to_bctq(output), to_bctq(reid)], dim=1),
        pred_reid_embed=to_bctq(reid), mask_features=mask_features,
        aux_logits=[c.unsqueeze(0) for c in pred_cls[:-1]],
        aux_masks=[m.permute(1, 0, 2, 3).unsqueeze(0) for m in pred_mask[:-1]],
        attn_masks=attn_masks)


# ----------------------------------------------------------------------------- tracker (a9, a10)
def match_embds(ref_embds, cur_embds):
    """Noiser.match_embds, dvis_Plus/noiser.py:43-56.  (q, b, c) each -> int64[q]."""
    ref, cur = ref_embds[:, 0, :], cur_embds[:, 0, :]
    ref = ref / (ref.norm(dim=1)[:, None] + 1e-6)
    cur = cur / (cur.norm(dim=1)[:, None] + 1e-6)
    C = 1 - torch.mm(cur, ref.transpose(0, 1))
    C = torch.where(torch.isnan(C), torch.full_like(C, 0), C)
    return np.asarray(linear_sum_assignment(C.transpose(0, 1).numpy())[1]).astype(np.int64)


class Tracker:
    """ReferringTracker_noiser (eval), dvis_Plus/tracker.py:94-380.  Keeps the cross-call state
    (last_outputs / last_frame_embeds / last_reference, :175-185) so `resume=True` continues a video."""

    def __init__(self, sd, nheads, layers, p=""):
        self.sd, self.nheads, self.layers, self.p = sd, nheads, layers, p
        self.last_outputs = self.last_frame_embeds = self.last_reference = None

    def forward(self, frame_embeds, mask_features, resume=False, frame_embeds_no_norm=None, with_masks=True):
        sd, p, H = self.sd, self.p, self.nheads
        fe = frame_embeds.permute(2, 3, 0, 1)                               # t, q, b, c
        fe_nn = fe if frame_embeds_no_norm is None else frame_embeds_no_norm.permute(2, 3, 0, 1)
        outputs, indices_all, refs = [], [], []
        for i in range(fe.shape[0]):
            single, single_nn = fe[i], fe_nn[i]
            frame_key = single_nn
            first = (i == 0 and not resume)
            if first:
                self.last_outputs = self.last_reference = None
                idx = match_embds(single, single)                           # tracker.py:241-247
            else:
                reference = mlp(sd, p + "ref_proj", self.last_outputs[-1], 3)     # :278
                self.last_reference = reference
                idx = match_embds(self.last_frame_embeds, single)           # :283-289
            ms_output = [single_nn[idx]]
            self.last_frame_embeds = single[idx]
            indices_all.append(idx)
            for j in range(self.layers):
                if first:                                                   # :250-276
                    ref_j = mlp(sd, p + "ref_proj", frame_key if j == 0 else ms_output[-1], 3)
                else:
                    ref_j = reference
                out = referring_cross_attention_layer(sd, p + f"transformer_cross_attention_layers.{j}",
                                                      ms_output[-1], ref_j, frame_key, single_nn, H)
                out = self_attention_layer(sd, p + f"transformer_self_attention_layers.{j}", out, H)
                out = ffn_layer(sd, p + f"transformer_ffn_layers.{j}", out)
                ms_output.append(out)
            if first:
                self.last_reference = mlp(sd, p + "ref_proj", frame_key, 3)   # :277
            refs.append(self.last_reference)
            self.last_outputs = torch.stack(ms_output, 0)
            outputs.append(self.last_outputs[-1])
        outputs = torch.stack(outputs, 0)                                     # (t, q, b, c) last layer only (eval :341)
        refs = torch.stack(refs, 0)
        dec = layer_norm(sd, p + "decoder_norm", outputs)
        logits = linear(sd, p + "class_embed", torch.cat([refs, dec], -1))    # (t, q, b, K+1)  :368-376
        res = dict(pred_logits=logits.permute(2, 0, 1, 3),                    # (b, t, q, K+1)
                   pred_embds=outputs.permute(2, 3, 0, 1), pred_references=refs.permute(2, 3, 0, 1),
                   indices=np.stack(indices_all))
        if with_masks:
            shp = mask_features.shape
            mf = F.conv2d(mask_features.flatten(0, 1), sd[p + "mask_feature_proj.weight"],
                          sd[p + "mask_feature_proj.bias"]).reshape(*shp)     # :199
            emb = mlp(sd, p + "mask_embed", dec, 3).permute(2, 0, 1, 3)       # (b, t, q, c)
            res["pred_masks"] = torch.einsum("btqc,btchw->bqthw", emb, mf)
        return res


# ----------------------------------------------------------------------------- refiner (a11)
def conv1d_replicate(x, w, b):
    k = w.shape[-1]
    return F.conv1d(F.pad(x, ((k - 1) // 2, k // 2), mode="replicate"), w, b)


def refiner_forward(sd, instance_embeds, frame_embeds, mask_features, nheads, layers, p=""):
    """TemporalRefiner.forward (eval), dvis_Plus/refiner.py:91-158 + prediction :212-227, :169-210."""
    B, C, T, Q = instance_embeds.shape
    output = instance_embeds
    fe = frame_embeds.permute(3, 0, 2, 1).flatten(1, 2)                       # (q, bt, c)
    for i in range(layers):
        output = output.permute(2, 0, 3, 1).flatten(1, 2)                     # (t, bq, c)
        output = self_attention_layer(sd, p + f"transformer_time_self_attention_layers.{i}", output, nheads)
        output = output.permute(1, 2, 0)                                      # (bq, c, t)
        cp = p + f"conv_short_aggregate_layers.{i}"
        y = conv1d_replicate(output, sd[cp + ".0.weight"], sd[cp + ".0.bias"])
        y = conv1d_replicate(F.relu(y), sd[cp + ".2.weight"], sd[cp + ".2.bias"])
        output = layer_norm(sd, p + f"conv_norms.{i}", (y + output).transpose(1, 2)).transpose(1, 2)
        output = output.reshape(B, Q, C, T).permute(1, 0, 3, 2).flatten(1, 2)  # (q, bt, c)
        output = self_attention_layer(sd, p + f"transformer_obj_self_attention_layers.{i}", output, nheads)
        output = cross_attention_layer(sd, p + f"transformer_cross_attention_layers.{i}", output, fe, nheads)
        output = ffn_layer(sd, p + f"transformer_ffn_layers.{i}", output)
        output = output.reshape(Q, B, T, C).permute(1, 3, 2, 0)               # (b, c, t, q)
    last = output.permute(2, 3, 0, 1)                                         # (t, q, b, c)
    dec = layer_norm(sd, p + "decoder_norm", last)                            # eval: last layer only (:226)
    dec_b = dec.permute(2, 0, 1, 3)                                           # (b, t, q, c)
    emb = mlp(sd, p + "mask_embed", dec_b, 3)
    masks = torch.einsum("btqc,btchw->bqthw", emb, mask_features)
    act = linear(sd, p + "activation_proj", dec_b).softmax(dim=1)             # softmax over t (:203)
    pooled = (dec_b * act).sum(dim=1, keepdim=True).repeat(1, T, 1, 1)
    logits = linear(sd, p + "class_embed", pooled)                            # (b, t, q, K+1)
    return dict(pred_logits=logits, pred_masks=masks, pred_embds=dec.permute(2, 3, 0, 1), mask_embed=emb)


# ----------------------------------------------------------------------------- post-processing (a12, a13)
def get_instance_labels(pred_logits):
    """_get_instance_labels, meta_architecture.py:708-714."""
    labels = torch.argmax(F.softmax(pred_logits[0], dim=-1), dim=2)
    labels[labels == pred_logits.shape[-1] - 1] = -1
    return labels


def post_processing(pred_logits, aux_logits=None):
    """post_processing, meta_architecture.py:758-772: mean class logits over T."""
    out = pred_logits[0].mean(dim=0)
    aux = None if aux_logits is None else aux_logits[0].mean(dim=0)
    return out, aux


def _resize2(masks, first_resize_size, img_size, out_hw, sigmoid):
    m = F.interpolate(masks, size=first_resize_size, mode="bilinear", align_corners=False)
    m = m[:, :, :img_size[0], :img_size[1]]
    if sigmoid:
        m = m.sigmoid()
    return F.interpolate(m, size=out_hw, mode="bilinear", align_corners=False)


def inference_video_vis(pred_cls, pred_masks, img_size, out_hw, first_resize_size, num_classes, max_num,
                        aux_pred_cls=None, diag=None):
    """inference_video_vis, meta_architecture.py:818-867.  Returns (scores, labels, ids, bool masks).
    diag: optional dict that receives the resized float logits behind the boolean masks (parity tests measure how
    far from the threshold a differing pixel is)."""
    Q = pred_cls.shape[0]
    scores = F.softmax(pred_cls, dim=-1)[:, :-1]
    if aux_pred_cls is not None:
        scores = torch.maximum(scores, F.softmax(aux_pred_cls, dim=-1)[:, :-1])
    labels = torch.arange(num_classes).unsqueeze(0).repeat(Q, 1).flatten(0, 1)
    scores_per_image, topk = scores.flatten(0, 1).topk(max_num, sorted=False)
    labels_per_image = labels[topk]
    qidx = topk // num_classes
    values = _resize2(pred_masks[qidx], first_resize_size, img_size, out_hw, sigmoid=False)
    if diag is not None:
        diag["vis_values"] = values
    return scores_per_image, labels_per_image, qidx, values > 0.


def inference_video_vps(pred_cls, pred_masks, img_size, out_hw, first_resize_size, num_classes, n_things,
                        object_mask_threshold, overlap_threshold, aux_pred_cls=None, diag=None):
    """inference_video_vps, meta_architecture.py:869-952.
    diag: optional dict that receives the candidates' scores, resized probabilities and arg-max ids."""
    pred_cls = F.softmax(pred_cls, dim=-1)
    if aux_pred_cls is not None:
        pred_cls[:, :-1] = torch.maximum(pred_cls[:, :-1], F.softmax(aux_pred_cls, dim=-1)[:, :-1])
    scores, labels = pred_cls.max(-1)
    keep = labels.ne(num_classes) & (scores > object_mask_threshold)
    ids = torch.arange(pred_cls.shape[0])[keep]
    cur_scores, cur_classes = scores[keep], labels[keep]
    cur_masks = _resize2(pred_masks[keep], first_resize_size, img_size, out_hw, sigmoid=True)
    panoptic = torch.zeros((cur_masks.size(1), *cur_masks.shape[-2:]), dtype=torch.int32)
    segments, out_ids = [], []
    if cur_masks.shape[0] == 0:
        return panoptic, segments, out_ids
    cur_mask_ids = (cur_scores.view(-1, 1, 1, 1) * cur_masks).argmax(0)
    if diag is not None:
        diag.update(vps_scores=cur_scores, vps_probs=cur_masks, vps_ids=cur_mask_ids, vps_query_ids=ids)
    seg_id, stuff = 0, {}
    for k in range(cur_classes.shape[0]):
        cls_k = int(cur_classes[k])
        isthing = cls_k < n_things
        mask_area = int((cur_mask_ids == k).sum())
        original_area = int((cur_masks[k] >= 0.5).sum())
        mask = (cur_mask_ids == k) & (cur_masks[k] >= 0.5)
        if mask_area > 0 and original_area > 0 and int(mask.sum()) > 0:
            if mask_area / original_area < overlap_threshold:
                continue
            if not isthing:
                if cls_k in stuff:
                    panoptic[mask] = stuff[cls_k]
                    continue
                stuff[cls_k] = seg_id + 1
            seg_id += 1
            panoptic[mask] = seg_id
            segments.append(dict(id=seg_id, isthing=bool(isthing), category_id=cls_k))
            out_ids.append(int(ids[k]))
    return panoptic, segments, out_ids


def inference_video_vss(pred_cls, pred_masks, img_size, out_hw, first_resize_size, aux_pred_cls=None, diag=None):
    """inference_video_vss, meta_architecture.py:954-979.  diag: receives the (C, T, H, W) class sums."""
    mask_cls = F.softmax(pred_cls, dim=-1)[..., :-1]
    if aux_pred_cls is not None:
        mask_cls = torch.maximum(mask_cls, F.softmax(aux_pred_cls, dim=-1)[..., :-1])
    cur_masks = _resize2(pred_masks, first_resize_size, img_size, out_hw, sigmoid=True)
    sem = torch.einsum("qc,qthw->cthw", mask_cls, cur_masks)
    if diag is not None:
        diag["vss_sums"] = sem
    return sem.max(0)[1]


# ----------------------------------------------------------------------------- whole offline / online path (a12)
def _sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def preprocess(frames, pixel_mean, pixel_std, size_divisibility=32):
    """(x - mean) / std, then zero-pad bottom/right to a multiple of 32 (meta_architecture.py:1306-1311;
    ImageList.from_tensors semantics, un-vendored detectron2 — SURVEY.md App. B)."""
    x = torch.stack([f.float() for f in frames])
    x = (x - pixel_mean.view(-1, 1, 1)) / pixel_std.view(-1, 1, 1)
    H, W = x.shape[-2:]
    d = size_divisibility
    Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
    return F.pad(x, (0, Wp - W, 0, Hp - H)), (H, W)


def dvis_plus_forward(sd, backbone, frames, *, offline=True, nheads=8, enc_layers=6, dec_layers=9, tracker_layers=6,
                      refiner_layers=6, window_size=3, num_classes=124, n_things=58, task="vps", max_num=20,
                      object_mask_threshold=0.8, overlap_threshold=0.8, out_hw=None, stages=None, keep=False,
                      tracker=None, seg_workers=1):
    """DVIS_Plus_offline.forward (eval) = run_window_inference (meta_architecture.py:1446-1500) + post_processing
    (:758-772) + inference_video_{vis,vps,vss}; offline=False: DVIS_Plus_online (:774-816).  `sd` uses the product's /
    reference's checkpoint names (sem_seg_head.pixel_decoder.*, sem_seg_head.predictor.*, tracker.*, refiner.*);
    `backbone` is any callable images -> {res2..res5} (un-pinned third-party part, shared with the product).
    `stages`: optional dict that receives intermediate tensors (for parity tests).
    keep / tracker: the `keep` entry of the input dict (:629-632, :1301-1304) and the Tracker object of the previous
    call (its cross-call state).  ONLINE: the first window resumes when keep is set (`i != 0 or self.keep`, :793).
    OFFLINE: the window loop resumes only for i != 0 (:1479-1486) — `keep` is read but has no effect there.
    nheads: an int, or (segmenter heads, tracker / refiner heads).
    seg_workers > 1 (the tests' time budget, not a change of the algorithm): the windows' SEGMENTER passes — independent of each
    other and of the tracker — are evaluated by that many host threads at once, window by window as before; the tracker then walks
    the windows in order.  Same calls, same window batches, another schedule (the torch CPU ops of a 3-frame window stop scaling at
    ~8 threads: tools/exp/oracle_threads.py)."""
    images, img_size = preprocess(frames, sd["pixel_mean"].flatten(), sd["pixel_std"].flatten())
    pd, pr = _sub(sd, "sem_seg_head.pixel_decoder."), _sub(sd, "sem_seg_head.predictor.")
    nheads, nheads_t = (nheads, nheads) if isinstance(nheads, int) else nheads
    trk = tracker if tracker is not None else Tracker(_sub(sd, "tracker."), nheads_t, tracker_layers)
    T = len(images)
    all_mf, all_fe_nn, all_inst, online_logits, online_masks, all_fe, all_ms0 = [], [], [], [], [], [], []
    def segment_window(feats):
        mf, _, ms = pixel_decoder_forward(pd, feats, nheads, enc_layers)
        return mf, ms, decoder_forward(pr, ms, mf, nheads, dec_layers)
    starts = list(range(0, T, window_size))
    ahead = None
    if seg_workers > 1 and len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        feats_all = [backbone(images[s:s + window_size]) for s in starts]   # (the backbone callable may drive a GPU: this thread only)
        with ThreadPoolExecutor(max_workers=int(seg_workers)) as ex:
            ahead = list(ex.map(lambda f: torch.no_grad()(segment_window)(f), feats_all))
        del feats_all
    for wi, s in enumerate(starts):                                         # the reference's window loop
        if ahead is not None:
            mf, ms, out = ahead[wi]
            ahead[wi] = None
        else:
            mf, ms, out = segment_window(backbone(images[s:s + window_size]))
        t_out = trk.forward(out["pred_embds"], mf.unsqueeze(0), resume=(s != 0) or (bool(keep) and not offline),
                            frame_embeds_no_norm=out["pred_embds_without_norm"], with_masks=not offline)
        all_mf.append(mf)
        if stages is not None and stages.get("want_attn_masks"):
            # effective attention masks (head 0 of each frame; fully blocked rows already reset), one list per layer
            stages.setdefault("attn_masks", []).append([a[::nheads].clone() for a in out["attn_masks"]])
        all_fe.append(out["pred_embds"])
        all_ms0.append(ms[-1])
        all_fe_nn.append(out["pred_embds_without_norm"])
        all_inst.append(t_out["pred_embds"])
        online_logits.append(t_out["pred_logits"])
        if not offline:
            online_masks.append(t_out["pred_masks"])
    mask_features = torch.cat(all_mf, 0).unsqueeze(0)
    online_logits = torch.cat(online_logits, 1)
    if offline:
        ref = refiner_forward(_sub(sd, "refiner."), torch.cat(all_inst, 2), torch.cat(all_fe_nn, 2), mask_features,
                              nheads_t, refiner_layers)
        cls, aux = post_processing(ref["pred_logits"], online_logits)
        masks = ref["pred_masks"][0]
    else:
        cls, aux = post_processing(online_logits)
        masks = torch.cat(online_masks, 2)[0]
    if stages is not None:
        stages.update(mask_features=mask_features, cls=cls, aux=aux, masks=masks, online_logits=online_logits,
                      tracker=trk, instance_embds=torch.cat(all_inst, 2), frame_embds=torch.cat(all_fe, 2),
                      frame_embds_no_norm=torch.cat(all_fe_nn, 2), encoder_finest=torch.cat(all_ms0, 0))
        if offline:
            stages.update(refiner_logits=ref["pred_logits"], refiner_embds=ref["pred_embds"],
                          refiner_mask_embed=ref.get("mask_embed"))
    first = tuple(images.shape[-2:])
    out_hw = img_size if out_hw is None else out_hw
    if task == "vis":
        return inference_video_vis(cls, masks, img_size, out_hw, first, num_classes, max_num, aux, diag=stages)
    if task == "vps":
        return inference_video_vps(cls, masks, img_size, out_hw, first, num_classes, n_things,
                                   object_mask_threshold, overlap_threshold, aux, diag=stages)
    return inference_video_vss(cls, masks, img_size, out_hw, first, aux, diag=stages)


# ----------------------------------------------------------------------------- image Mask2Former (BASELINE config #1)
def maskformer_image_forward(sd, backbone, image, *, nheads=8, enc_layers=6, dec_layers=9, num_classes=133,
                             out_hw=None, stages=None):
    """MaskFormer.forward eval (mask2former/maskformer_model.py:194-262) for one image with semantic inference
    (:280-284): backbone -> pixel decoder -> image decoder -> upsample to the padded size -> crop + resize
    (sem_seg_postprocess, un-vendored detectron2: "parity unpinned") -> einsum(softmax(cls)[:-1], sigmoid(mask)).
    Returns (sem_seg (K,H,W), pred_logits, pred_masks at stride 4)."""
    images, img_size = preprocess([image], sd["pixel_mean"].flatten(), sd["pixel_std"].flatten())
    feats = backbone(images)
    mf, _, ms = pixel_decoder_forward(_sub(sd, "sem_seg_head.pixel_decoder."), feats, nheads, enc_layers)
    out = decoder_forward(_sub(sd, "sem_seg_head.predictor."), ms, mf, nheads, dec_layers, dvis_plus=False)
    if stages is not None:      # parity tests: mask_features and every layer's effective attention mask (head 0)
        stages.update(mask_features=mf, attn_masks=[a[::nheads].clone() for a in out["attn_masks"]])
    masks = F.interpolate(out["pred_masks"], size=tuple(images.shape[-2:]), mode="bilinear", align_corners=False)[0]
    out_hw = img_size if out_hw is None else out_hw
    masks = F.interpolate(masks[:, :img_size[0], :img_size[1]][None], size=tuple(out_hw), mode="bilinear",
                          align_corners=False)[0]
    cls = F.softmax(out["pred_logits"][0], dim=-1)[..., :-1]
    return torch.einsum("qc,qhw->chw", cls, masks.sigmoid()), out["pred_logits"], out["pred_masks"]


# --------------------------------------------------------------------------- MinVIS (meta_architecture.py:23-407)
def minvis_match(tgt_embds, cur_embds):
    """MinVIS.match_from_embds, meta_architecture.py:255-264: cosine cost WITHOUT the 1e-6 of Noiser.match_embds."""
    from scipy.optimize import linear_sum_assignment
    cur = cur_embds / cur_embds.norm(dim=1)[:, None]
    tgt = tgt_embds / tgt_embds.norm(dim=1)[:, None]
    C = (1 - torch.mm(cur, tgt.transpose(0, 1))).cpu()
    return linear_sum_assignment(C.transpose(0, 1))[1]


def minvis_post_processing(pred_logits, pred_masks, pred_embds):
    """MinVIS.post_processing, meta_architecture.py:266-301.  (1,T,Q,K+1), (1,Q,T,h,w), (1,C,T,Q) ->
    logits (1,Q,K+1) averaged over the aligned frames, masks (1,Q,T,h,w) aligned, per-frame permutations."""
    logits = list(torch.unbind(pred_logits[0]))
    masks = list(torch.unbind(pred_masks[0].permute(1, 0, 2, 3)))
    embds = list(torch.unbind(pred_embds[0].permute(1, 2, 0)))
    out_logits, out_masks, out_embds, perms = [logits[0]], [masks[0]], [embds[0]], [np.arange(logits[0].shape[0])]
    for i in range(1, len(logits)):
        idx = minvis_match(out_embds[-1], embds[i])
        perms.append(np.asarray(idx))
        out_logits.append(logits[i][idx, :])
        out_masks.append(masks[i][idx, :, :])
        out_embds.append(embds[i][idx, :])
    return (sum(out_logits) / len(out_logits)).unsqueeze(0), torch.stack(out_masks, dim=1).unsqueeze(0), np.stack(perms)


def minvis_inference_video(pred_cls, pred_masks, img_size, out_hw, first_resize_size, num_classes, topk=10):
    """MinVIS.inference_video, meta_architecture.py:362-407 -> (scores, labels, masks bool (k,T,H,W), query index)."""
    Q = pred_cls.shape[0]
    scores = F.softmax(pred_cls, dim=-1)[:, :-1]
    labels = torch.arange(num_classes).unsqueeze(0).repeat(Q, 1).flatten(0, 1)
    s, idx = scores.flatten(0, 1).topk(topk, sorted=False)
    q = idx // num_classes
    m = F.interpolate(pred_masks[q], size=tuple(first_resize_size), mode="bilinear", align_corners=False)
    m = m[:, :, :img_size[0], :img_size[1]]
    m = F.interpolate(m, size=tuple(out_hw), mode="bilinear", align_corners=False)
    return s, labels[idx], m > 0., q
