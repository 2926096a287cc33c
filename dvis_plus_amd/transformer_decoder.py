"""Masked-attention transformer decoder — host side of SURVEY.md §8 rows a6, a7, a8.

Mirrors (constructor arguments, attribute names, ``state_dict`` keys):
  SelfAttentionLayer / CrossAttentionLayer / FFNLayer / MLP
        mask2former_video/modeling/transformer_decoder/video_mask2former_transformer_decoder.py:18-206
  VideoMultiScaleMaskedTransformerDecoder_dvisPlus     dvis_Plus/video_mask2former_transformer_decoder.py:174-374
  VideoMultiScaleMaskedTransformerDecoder_minvis/_dvis ibid. :11-171 (same layers, no re-id head)
  MultiScaleMaskedTransformerDecoder (image, BASELINE config #1)
        mask2former/modeling/transformer_decoder/mask2former_transformer_decoder.py:207-461

Inference-only re-organisation for MI355X (same numbers as the reference's eval path):
  * frames are the batch; the K / V projections of ALL layers that read a feature level are ONE GEMM per level
    (the memory does not depend on the queries), done before the layer loop;
  * attention = hand-written fp32-MFMA kernel on strided slices of those GEMM outputs (no (B*8, Q, HW)
    probability tensor, no per-head mask copies);
  * forward_prediction_heads = one fused kernel per layer (contraction + bilinear down-sizing + threshold);
    full-resolution logits are only produced on request, class logits only for the last layer
    (the reference computes and discards the others in eval);
  * the "row blocked everywhere" reset needs no host sync (allowed_count travels with the mask).
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import functions as Fn
from .pixel_decoder import ConvNorm, PositionEmbeddingSine, c2_xavier_fill
from .d2 import configurable
from .registry import TRANSFORMER_DECODER_REGISTRY


def _xavier_(module):
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


class SelfAttentionLayer(nn.Module):
    own_gemm = None      # None: the frame-invariant own GEMM of Fn.linear (segmenter); the tracker / refiner set True on their layers

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        if normalize_before:
            raise NotImplementedError("PRE_NORM=True is not used by the DVIS++ configs")
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)   # parameter container
        self.norm = nn.LayerNorm(d_model)
        self.nhead = nhead
        _xavier_(self)

    def forward(self, tgt, tgt_mask=None, tgt_key_padding_mask=None, query_pos=None):
        assert tgt_mask is None and tgt_key_padding_mask is None
        C = tgt.shape[-1]
        W, b = self.self_attn.in_proj_weight, self.self_attn.in_proj_bias
        own = self.own_gemm
        if query_pos is None:
            qkv = Fn.linear(tgt, W, b, own=own)
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        else:
            qk = Fn.linear(tgt + query_pos, W[:2 * C], b[:2 * C], own=own)
            q, k = qk[..., :C], qk[..., C:]
            v = Fn.linear(tgt, W[2 * C:], b[2 * C:], own=own)
        att = Fn.attention(q, k, v, self.nhead)
        op = self.self_attn.out_proj
        return Fn.add_layer_norm(Fn.linear(att, op.weight, op.bias, own=own), tgt, self.norm)


class CrossAttentionLayer(nn.Module):
    own_gemm = None

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        if normalize_before:
            raise NotImplementedError("PRE_NORM=True is not used by the DVIS++ configs")
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        self.nhead = nhead
        _xavier_(self)

    def project_q(self, x):
        C = x.shape[-1]
        return Fn.linear(x, self.multihead_attn.in_proj_weight[:C], self.multihead_attn.in_proj_bias[:C],
                         own=self.own_gemm)

    def kv_weights(self):
        C = self.multihead_attn.embed_dim
        W, b = self.multihead_attn.in_proj_weight, self.multihead_attn.in_proj_bias
        return W[C:2 * C], b[C:2 * C], W[2 * C:], b[2 * C:]

    def attend(self, tgt, q_in, k_proj, v_proj, mask=None, allowed=None, identity=None):
        """k_proj / v_proj are already projected (Lk, B, C) views."""
        att = Fn.attention(self.project_q(q_in), k_proj, v_proj, self.nhead, mask, allowed)
        res = tgt if identity is None else identity
        op = self.multihead_attn.out_proj
        return Fn.add_layer_norm(Fn.linear(att, op.weight, op.bias, own=self.own_gemm), res, self.norm)

    def forward(self, tgt, memory, memory_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None):
        assert memory_key_padding_mask is None
        Wk, bk, Wv, bv = self.kv_weights()
        k = Fn.linear(memory if pos is None else memory + pos, Wk, bk, own=self.own_gemm)
        v = Fn.linear(memory, Wv, bv, own=self.own_gemm)
        q_in = tgt if query_pos is None else tgt + query_pos
        return self.attend(tgt, q_in, k, v, memory_mask)


class FFNLayer(nn.Module):
    own_gemm = None

    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        if normalize_before:
            raise NotImplementedError("PRE_NORM=True is not used by the DVIS++ configs")
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        _xavier_(self)

    def forward(self, tgt):
        h = Fn.linear_relu(tgt, self.linear1, own=self.own_gemm)
        return Fn.add_layer_norm(Fn.linear(h, self.linear2.weight, self.linear2.bias, own=self.own_gemm), tgt, self.norm)


class MLP(nn.Module):
    own_gemm = None

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = Fn.linear(x, layer.weight, layer.bias, relu=i < self.num_layers - 1, own=self.own_gemm)
        return x


class _MaskedDecoderBase(nn.Module):
    """Shared body of the image / video masked-attention decoders."""

    _version = 2

    @configurable
    def __init__(self, in_channels, mask_classification=True, *, num_classes, hidden_dim, num_queries, nheads,
                 dim_feedforward, dec_layers, pre_norm, mask_dim, enforce_input_project):
        super().__init__()
        assert mask_classification, "Only support mask classification model"
        self.mask_classification = mask_classification
        self.pe_layer = PositionEmbeddingSine(hidden_dim // 2, normalize=True)
        self.num_heads, self.num_layers = nheads, dec_layers
        self.transformer_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        for _ in range(self.num_layers):
            self.transformer_self_attention_layers.append(
                SelfAttentionLayer(d_model=hidden_dim, nhead=nheads, dropout=0.0, normalize_before=pre_norm))
            self.transformer_cross_attention_layers.append(
                CrossAttentionLayer(d_model=hidden_dim, nhead=nheads, dropout=0.0, normalize_before=pre_norm))
            self.transformer_ffn_layers.append(
                FFNLayer(d_model=hidden_dim, dim_feedforward=dim_feedforward, dropout=0.0, normalize_before=pre_norm))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.num_queries = num_queries
        self.query_feat = nn.Embedding(num_queries, hidden_dim)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.num_feature_levels = 3
        self.level_embed = nn.Embedding(self.num_feature_levels, hidden_dim)
        self.input_proj = nn.ModuleList()
        for _ in range(self.num_feature_levels):
            if in_channels != hidden_dim or enforce_input_project:
                self.input_proj.append(ConvNorm(in_channels, hidden_dim, kernel_size=1))
                c2_xavier_fill(self.input_proj[-1])
            else:
                self.input_proj.append(nn.Sequential())
        self.class_embed = nn.Linear(hidden_dim, num_classes + 1)
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)
        self._kv_cache = None
        self.debug_masks = None      # tests: a list here receives every layer's effective attention mask (N, Q, hw) bool

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # version shim of the reference (video_mask2former_transformer_decoder.py:213-234): static_query -> query_feat
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            for k in list(state_dict.keys()):
                if k.startswith(prefix) and "static_query" in k:
                    state_dict[k.replace("static_query", "query_feat")] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    @staticmethod
    def _base_from_config(cfg, in_channels, mask_classification):
        mf = cfg.MODEL.MASK_FORMER
        assert mf.DEC_LAYERS >= 1
        return dict(in_channels=in_channels, mask_classification=mask_classification,
                    num_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES, hidden_dim=mf.HIDDEN_DIM,
                    num_queries=mf.NUM_OBJECT_QUERIES, nheads=mf.NHEADS, dim_feedforward=mf.DIM_FEEDFORWARD,
                    dec_layers=mf.DEC_LAYERS - 1, pre_norm=mf.PRE_NORM, enforce_input_project=mf.ENFORCE_INPUT_PROJ,
                    mask_dim=cfg.MODEL.SEM_SEG_HEAD.MASK_DIM)

    # ---- K / V projection weights of all layers reading level l, concatenated once
    def _level_kv_weights(self):
        ver = tuple(t._version for l in self.transformer_cross_attention_layers
                    for t in (l.multihead_attn.in_proj_weight, l.multihead_attn.in_proj_bias)) \
            + (self.level_embed.weight._version,)
        dev = self.decoder_norm.weight.device
        if self._kv_cache is None or self._kv_cache[0] != (ver, dev):
            per_level = []
            for lvl in range(self.num_feature_levels):
                idx = [i for i in range(self.num_layers) if i % self.num_feature_levels == lvl]
                ws = [self.transformer_cross_attention_layers[i].kv_weights() for i in idx]
                if not ws:
                    per_level.append(None)
                    continue
                Wv, bv = torch.cat([w[2] for w in ws], 0).detach(), torch.cat([w[3] for w in ws], 0).detach()
                # V = (tok + level_embed) Wv^T + bv = tok Wv^T + (bv + Wv level_embed): the folded bias, made once per weights
                bv_le = (bv.double() + Wv.double() @ self.level_embed.weight[lvl].detach().double()).float()
                per_level.append((idx, torch.cat([w[0] for w in ws], 0).detach(), torch.cat([w[1] for w in ws], 0).detach(),
                                  Wv, bv, bv_le))
            self._kv_cache = ((ver, dev), per_level)
        return self._kv_cache[1]

    @staticmethod
    def _f32_inputs(x, mask_features):
        """Maps handed over by a half-precision region outside the island (a foreign backbone / pixel decoder under autocast)."""
        if any(m.dtype != torch.float32 for m in x):
            tokens = getattr(x, "tokens", None)
            x = type(x)(Fn.f32(m) for m in x)
            if tokens is not None:
                x.tokens = [Fn.f32(t) for t in tokens]
        return x, Fn.f32(mask_features)

    def _run_layers(self, x, mask_features):
        """x: 3 feature maps (N, C, h_l, w_l), low-res first; mask_features (N, Cm, H, W).
        Returns the un-normed residual stream (Q, N, C) after the last layer."""
        C = self.decoder_norm.weight.shape[0]
        N = x[0].shape[0]
        size_list, kproj, vproj = [], {}, {}
        kvw = self._level_kv_weights()
        tokens = getattr(x, "tokens", None)
        with Fn.x3_stage("decoder_kv"):
            for lvl in range(self.num_feature_levels):
                h, w = x[lvl].shape[-2:]
                size_list.append((h, w))
                if kvw[lvl] is None:
                    continue
                idx, Wk, bk, Wv, bv, bv_le = kvw[lvl]
                le = self.level_embed.weight[lvl]
                ident = isinstance(self.input_proj[lvl], nn.Sequential) and len(self.input_proj[lvl]) == 0
                if tokens is not None and ident and not torch.is_grad_enabled():
                    # token-major fast path: K = (tok + level_embed + pos) Wk^T + bk with ONE add pass on the (N, hw, C) tokens;
                    # V = (tok + level_embed) Wv^T + bv = tok Wv^T + (bv + Wv level_embed): no pass at all
                    tok = tokens[lvl]                                                               # (N, hw, C)
                    pos_t = self.pe_layer.compute(h, w, tok.device).flatten(2).transpose(1, 2)      # (1, hw, C)
                    if Fn.x3_on() and Fn.x3_ok(tok, Wk.shape[0], Wk.shape[1], add=True):
                        # every pixel's keys / values for all layers of the level: split-f16 matrix-core GEMM (csrc/gemm_x3.hip)
                        tok = tok.contiguous()          # a level's rows of the encoder memory: ONE compaction serves both projections
                        kall = Fn.x3_linear(tok, Wk, bk, xadd=pos_t + le).transpose(0, 1)   # (the (N, hw, C) sum is never written)
                        vall = Fn.x3_linear(tok, Wv, bv_le).transpose(0, 1)
                    else:
                        kall = Fn.linear(tok + (pos_t + le), Wk, bk, tall=True).transpose(0, 1)                # (hw, N, n_l * C) view
                        vall = Fn.linear(tok, Wv, bv_le, tall=True).transpose(0, 1)
                else:
                    src = (self.input_proj[lvl](x[lvl]).flatten(2) + le[None, :, None]).permute(2, 0, 1)
                    pos = self.pe_layer.compute(h, w, x[lvl].device).flatten(2).permute(2, 0, 1)   # (hw, 1, C)
                    kall = Fn.linear(src + pos, Wk, bk, tall=True)                                          # (hw, N, n_l * C)
                    vall = Fn.linear(src, Wv, bv, tall=True)
                for n, i in enumerate(idx):
                    kproj[i], vproj[i] = kall[..., n * C:(n + 1) * C], vall[..., n * C:(n + 1) * C]
        query_embed = self.query_embed.weight.unsqueeze(1)                                      # (Q, 1, C) broadcasts
        output = self.query_feat.weight.unsqueeze(1).repeat(1, N, 1)
        # the attention masks' feature pyramid: the four centre pixels of every 8 x 8 / 4 x 4 / 2 x 2 block of mask_features
        # averaged ONCE (one read of the map) — every layer then contracts its level's small map instead of the stride-4 one
        pyramid = Fn.center_pool3(mask_features)
        if pyramid is not None and [tuple(p.shape[-2:]) for p in pyramid] != [tuple(sz) for sz in size_list[:3]]:
            pyramid = None                                                                      # (levels that are not 1/8, 1/4, 1/2)
        for i in range(self.num_layers):
            lvl = i % self.num_feature_levels
            emb = self.mask_embed(self.decoder_norm(output).transpose(0, 1))                    # (N, Q, Cm)
            if pyramid is not None:
                mask, allowed = Fn.attn_mask_pooled(emb.contiguous(), pyramid[lvl])
            else:
                mask, allowed = Fn.attn_mask(emb.contiguous(), mask_features, size_list[lvl])
            if self.debug_masks is not None:     # rows blocked everywhere attend everywhere (…decoder.py:297)
                self.debug_masks.append(mask.bool() & (allowed > 0)[..., None])
            layer = self.transformer_cross_attention_layers[i]
            output = layer.attend(output, output + query_embed, kproj[i], vproj[i], mask, allowed)
            output = self.transformer_self_attention_layers[i](output, query_pos=query_embed)
            output = self.transformer_ffn_layers[i](output)
        return output

    def _final_heads(self, output, mask_features, need_masks):
        dec = self.decoder_norm(output).transpose(0, 1)                                         # (N, Q, C)
        logits = Fn.linear(dec, self.class_embed.weight, self.class_embed.bias)
        masks = Fn.mask_logits(self.mask_embed(dec).contiguous(), mask_features) if need_masks else None
        return dec, logits, masks


@TRANSFORMER_DECODER_REGISTRY.register()
class MultiScaleMaskedTransformerDecoder(_MaskedDecoderBase):
    """Image Mask2Former decoder (BASELINE config #1).  Inference outputs only (no aux_outputs)."""

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        return cls._base_from_config(cfg, in_channels, mask_classification)

    @Fn.fp32_island
    def forward(self, x, mask_features, mask=None):
        assert len(x) == self.num_feature_levels
        x, mask_features = self._f32_inputs(x, mask_features)
        output = self._run_layers(x, mask_features)
        _, logits, masks = self._final_heads(output, mask_features, True)
        return {"pred_logits": logits, "pred_masks": masks, "aux_outputs": []}


@TRANSFORMER_DECODER_REGISTRY.register()
class VideoMultiScaleMaskedTransformerDecoder_dvisPlus(_MaskedDecoderBase):
    @configurable
    def __init__(self, in_channels, mask_classification=True, *, num_classes, hidden_dim, num_queries, nheads,
                 dim_feedforward, dec_layers, pre_norm, mask_dim, enforce_input_project, num_frames,
                 num_reid_head_layers, reid_hidden_dim):
        super().__init__(in_channels, mask_classification, num_classes=num_classes, hidden_dim=hidden_dim,
                         num_queries=num_queries, nheads=nheads, dim_feedforward=dim_feedforward,
                         dec_layers=dec_layers, pre_norm=pre_norm, mask_dim=mask_dim,
                         enforce_input_project=enforce_input_project)
        self.num_frames = num_frames
        if num_reid_head_layers > 0:
            self.reid_embed = MLP(hidden_dim, reid_hidden_dim, hidden_dim, num_reid_head_layers)
            for layer in self.reid_embed.layers:
                c2_xavier_fill(layer)
        else:
            self.reid_embed = nn.Identity()
        self.compute_pred_masks = True     # the offline / online meta-architectures switch this off (unused there)

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        ret = cls._base_from_config(cfg, in_channels, mask_classification)
        ret.update(num_frames=cfg.INPUT.SAMPLING_FRAME_NUM, reid_hidden_dim=cfg.MODEL.MASK_FORMER.REID_HIDDEN_DIM,
                   num_reid_head_layers=cfg.MODEL.MASK_FORMER.NUM_REID_HEAD_LAYERS)
        return ret

    @Fn.fp32_island
    def forward(self, x, mask_features, mask=None):
        """Eval semantics of the reference (bs = 1: all frames form one clip).  Shapes as in the reference:
        pred_logits (1,T,Q,K+1), pred_masks (1,Q,T,H,W) or None, pred_embds / pred_embds_without_norm (1,2C,T,Q),
        pred_reid_embed (1,C,T,Q), mask_features (T,Cm,H,W)."""
        assert len(x) == self.num_feature_levels
        if self.training:
            raise NotImplementedError("dvis_plus_amd decoders implement the inference path")
        x, mask_features = self._f32_inputs(x, mask_features)
        output = self._run_layers(x, mask_features)
        dec, logits, masks = self._final_heads(output, mask_features, self.compute_pred_masks)
        reid = self.reid_embed(dec)                                                              # (T, Q, C)
        to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)                                      # (T,Q,C) -> (1,C,T,Q)
        reid_b = to_bctq(reid)
        return {
            "pred_logits": logits.unsqueeze(0),
            "pred_masks": None if masks is None else masks.permute(1, 0, 2, 3).unsqueeze(0),
            "aux_outputs": [],
            "pred_embds": torch.cat([to_bctq(dec), reid_b], dim=1),
            "pred_embds_without_norm": torch.cat([output.permute(2, 1, 0).unsqueeze(0), reid_b], dim=1),
            "pred_reid_embed": reid_b,
            "mask_features": mask_features,
        }


@TRANSFORMER_DECODER_REGISTRY.register()
class VideoMultiScaleMaskedTransformerDecoder_dvis(_MaskedDecoderBase):
    """DVIS (v1) per-frame decoder: single-branch embeddings, no re-id head (dvis_Plus/…decoder.py:11-145)."""

    @configurable
    def __init__(self, in_channels, mask_classification=True, *, num_classes, hidden_dim, num_queries, nheads,
                 dim_feedforward, dec_layers, pre_norm, mask_dim, enforce_input_project, num_frames):
        super().__init__(in_channels, mask_classification, num_classes=num_classes, hidden_dim=hidden_dim,
                         num_queries=num_queries, nheads=nheads, dim_feedforward=dim_feedforward,
                         dec_layers=dec_layers, pre_norm=pre_norm, mask_dim=mask_dim,
                         enforce_input_project=enforce_input_project)
        self.num_frames = num_frames

    @classmethod
    def from_config(cls, cfg, in_channels, mask_classification):
        ret = cls._base_from_config(cfg, in_channels, mask_classification)
        ret.update(num_frames=cfg.INPUT.SAMPLING_FRAME_NUM)
        return ret

    @Fn.fp32_island
    def forward(self, x, mask_features, mask=None):
        if self.training:
            raise NotImplementedError("dvis_plus_amd decoders implement the inference path")
        x, mask_features = self._f32_inputs(x, mask_features)
        output = self._run_layers(x, mask_features)
        dec, logits, masks = self._final_heads(output, mask_features, True)
        return {"pred_logits": logits.unsqueeze(0), "pred_masks": masks.permute(1, 0, 2, 3).unsqueeze(0),
                "aux_outputs": [], "pred_embds": dec.permute(2, 0, 1).unsqueeze(0),
                "pred_embds_without_norm": output.permute(2, 1, 0).unsqueeze(0), "mask_features": mask_features}


@TRANSFORMER_DECODER_REGISTRY.register()
class VideoMultiScaleMaskedTransformerDecoder_minvis(VideoMultiScaleMaskedTransformerDecoder_dvis):
    """MinVIS: the `_dvis` decoder without `mask_features` in its outputs (…decoder.py:164-171)."""

    def forward(self, x, mask_features, mask=None):
        out = super().forward(x, mask_features, mask=mask)
        del out["mask_features"]
        return out
