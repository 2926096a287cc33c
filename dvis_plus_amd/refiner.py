"""Temporal refiner — host side of SURVEY.md §8 row a11.

Mirrors ``TemporalRefiner`` (dvis_Plus/refiner.py:6-227): constructor arguments, ``state_dict`` keys, call signature
and returned dict (eval branch: last layer only, window-free).

MI355X re-organisation:
  * one activation layout (T, Q, C) for all four sub-layers: the attention kernel takes row / batch strides, so "over
    time" and "over queries" are two views of the same buffer (the reference permutes + copies between them), and the
    short-aggregate nn.Conv1d pair runs as im2col GEMMs over clamped time indices (= replicate padding);
  * every projection runs on the deterministic own GEMM (csrc/gemm.hip): no library kernel on the refiner's stream;
  * the K / V projections of the 6 cross-attention layers over the frame queries are one GEMM;
  * time / object / cross attention run on the fp32-MFMA attention kernel;
  * mask_features never leave HBM: the reference moves every window of them host->device and every mask
    device->host (refiner.py:188-194); here the final contraction is one kernel over device-resident features, and it
    can be restricted to the queries post-processing keeps (``query_index``).
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import functions as Fn
from .graphs import ConvAsGemm, FusedKV, GraphRunner
from .tracker import use_own_gemm
from .transformer_decoder import MLP, CrossAttentionLayer, FFNLayer, SelfAttentionLayer


class TemporalRefiner(nn.Module):
    def __init__(self, hidden_channel=256, feedforward_channel=2048, num_head=8, decoder_layer_num=6, mask_dim=256,
                 class_num=25, windows=5):
        super().__init__()
        self.windows = windows            # kept for the ctor surface; no windowing is needed with 288 GB of HBM
        self.num_heads, self.num_layers = num_head, decoder_layer_num
        self.transformer_obj_self_attention_layers = nn.ModuleList()
        self.transformer_time_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        self.conv_short_aggregate_layers = nn.ModuleList()
        self.conv_norms = nn.ModuleList()
        for _ in range(self.num_layers):
            self.transformer_time_self_attention_layers.append(
                SelfAttentionLayer(d_model=hidden_channel, nhead=num_head, dropout=0.0, normalize_before=False))
            self.conv_short_aggregate_layers.append(nn.Sequential(
                nn.Conv1d(hidden_channel, hidden_channel, kernel_size=5, stride=1, padding="same",
                          padding_mode="replicate"),
                nn.ReLU(inplace=True),
                nn.Conv1d(hidden_channel, hidden_channel, kernel_size=3, stride=1, padding="same",
                          padding_mode="replicate")))
            self.conv_norms.append(nn.LayerNorm(hidden_channel))
            self.transformer_obj_self_attention_layers.append(
                SelfAttentionLayer(d_model=hidden_channel, nhead=num_head, dropout=0.0, normalize_before=False))
            self.transformer_cross_attention_layers.append(
                CrossAttentionLayer(d_model=hidden_channel, nhead=num_head, dropout=0.0, normalize_before=False))
            self.transformer_ffn_layers.append(
                FFNLayer(d_model=hidden_channel, dim_feedforward=feedforward_channel, dropout=0.0,
                         normalize_before=False))
        self.decoder_norm = nn.LayerNorm(hidden_channel)
        self.class_embed = nn.Linear(hidden_channel, class_num + 1)
        self.mask_embed = MLP(hidden_channel, hidden_channel, mask_dim, 3)
        self.activation_proj = nn.Linear(hidden_channel, 1)
        self._kv_cache = FusedKV()
        self._conv_weights = ConvAsGemm()
        self.use_graphs = True
        self._graph = GraphRunner(self.refine)
        use_own_gemm(self)

    def _kv_weights(self):
        return self._kv_cache.get(self.transformer_cross_attention_layers, self.decoder_norm.weight.shape[0])

    def _attention(self, q, k, v, over_time, clips=1):
        """q / k / v: (T, clips * Q, C) views with unit inner stride.  over_time: sequences run over T (one per query slot
        of every clip), otherwise over the Q queries of one frame of one clip.  -> (T, clips * Q, C) contiguous.  The
        attention kernel takes row / batch strides, so the reference's permute + flatten copies between its three layouts
        (refiner.py:104-139) never happen; several clips are more batch entries ((T, clips, Q, C) memory order: a frame of a
        clip is a contiguous (Q, C) block)."""
        out = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        if over_time:
            return Fn.attention(q, k, v, self.num_heads, out=out)
        T, QB, C = q.shape
        per_frame = lambda z: z.view(T * clips, QB // clips, C).transpose(0, 1)    # (Q, T * clips, C): batch = (frame, clip)
        Fn.attention(per_frame(q), per_frame(k), per_frame(v), self.num_heads, out=per_frame(out))
        return out

    def _self_attention(self, layer, x, over_time, clips=1):
        C = x.shape[-1]
        qkv = Fn.linear(x, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias, own=True)
        att = self._attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], over_time, clips)
        op = layer.self_attn.out_proj
        return Fn.add_layer_norm(Fn.linear(att, op.weight, op.bias, own=True), x, layer.norm)

    def _short_aggregate(self, i, x):
        """``conv1d_k3(relu(conv1d_k5(x)))`` over time with replicate 'same' padding (refiner.py:42-54) on x (T, Q, C):
        each convolution is one GEMM over the im2col rows [x[t-p], ..., x[t+p]] (time indices clamped = replicate)."""
        T = x.shape[0]
        for conv, relu in ((self.conv_short_aggregate_layers[i][0], True), (self.conv_short_aggregate_layers[i][2], False)):
            k = conv.kernel_size[0]
            idx = (torch.arange(T, device=x.device)[:, None] + torch.arange(k, device=x.device)[None] - k // 2).clamp_(0, T - 1)
            cols = x[idx].permute(0, 2, 1, 3).flatten(2)                           # (T, Q, k * C)
            x = Fn.linear(cols, self._conv_weights.get(conv), conv.bias, relu=relu, own=True)
        return x

    def refine(self, instance_embeds, frame_embeds):
        """The 6 refinement layers.  (b, c, t, q) x2 -> last layer's queries (t, q, b, c), un-normed.  b = the clips of a
        round (the reference runs one video at a time): every op is row-wise, per (clip, query slot) over time or per (clip,
        frame) over queries, and the GEMMs' tile configuration is pinned to ONE clip's rows — a clip's bits do not depend on
        its round mates."""
        B, C, T, Q = instance_embeds.shape
        x = instance_embeds.permute(2, 0, 3, 1).reshape(T, B * Q, C)               # (T, clips * Q, C): the one layout used below
        fe = frame_embeds.permute(2, 0, 3, 1).reshape(T, B * Q, C)
        W, b = self._kv_weights()
        with Fn.gemm_sizes_as(rows=T * Q):
            kv = Fn.linear(fe, W, b, own=True)                                     # (T, clips * Q, layers * 2C): one GEMM
            for i in range(self.num_layers):
                x = self._self_attention(self.transformer_time_self_attention_layers[i], x, over_time=True)
                x = Fn.add_layer_norm(self._short_aggregate(i, x), x, self.conv_norms[i])
                x = self._self_attention(self.transformer_obj_self_attention_layers[i], x, over_time=False, clips=B)
                layer = self.transformer_cross_attention_layers[i]
                att = self._attention(layer.project_q(x), kv[..., (2 * i) * C:(2 * i + 1) * C],
                                      kv[..., (2 * i + 1) * C:(2 * i + 2) * C], over_time=False, clips=B)
                op = layer.multihead_attn.out_proj
                x = Fn.add_layer_norm(Fn.linear(att, op.weight, op.bias, own=True), x, layer.norm)
                x = self.transformer_ffn_layers[i](x)
        return x.view(T, B, Q, C).transpose(1, 2)                                  # (t, q, b, c)

    def pred_class(self, decoder_output):
        """(l, b, t, q, c): softmax-over-time pooled class logits, repeated T times (refiner.py:196-210)."""
        T = decoder_output.size(2)
        ap, ce = self.activation_proj, self.class_embed
        activation = Fn.linear(decoder_output, ap.weight, ap.bias, own=True).softmax(dim=2)
        class_output = (decoder_output * activation).sum(dim=2, keepdim=True).repeat(1, 1, T, 1, 1)
        return Fn.linear(class_output, ce.weight, ce.bias, own=True).transpose(2, 3)

    @Fn.fp32_island
    def forward(self, instance_embeds, frame_embeds, mask_features, need_masks=True, query_index=None):
        """instance_embeds / frame_embeds (b, c, t, q), mask_features (b, t, c, h, w) device-resident.
        Returns pred_logits (b,t,q,K+1), pred_masks (b,q',t,h,w) [q' = len(query_index) if given] or None,
        pred_embds (b,c,t,q), mask_embed (b,t,q,Cm)."""
        if self.training:
            raise NotImplementedError("dvis_plus_amd implements the refiner's inference path")
        instance_embeds, frame_embeds, mask_features = Fn.f32(instance_embeds), Fn.f32(frame_embeds), Fn.f32(mask_features)
        self._kv_weights()
        for seq in self.conv_short_aggregate_layers:                               # cached GEMM weights, outside capture
            self._conv_weights.get(seq[0]), self._conv_weights.get(seq[2])
        self._graph.enabled = self.use_graphs
        last = self._graph("refine", instance_embeds.contiguous(), frame_embeds.contiguous()).clone()   # (t, q, b, c)
        dec = self.decoder_norm(last)
        dec_b = dec.permute(2, 0, 1, 3)                                            # (b, t, q, c)
        with Fn.gemm_sizes_as(rows=dec_b.shape[1] * dec_b.shape[2]):               # (one clip's rows: see refine)
            emb = self.mask_embed(dec_b)                                           # (b, t, q, Cm)
            logits = self.pred_class(dec_b[None])[0].transpose(1, 2)               # (b, t, q, K+1)
        out = {"pred_logits": logits, "pred_masks": None, "aux_outputs": [],
               "pred_embds": dec.permute(2, 3, 0, 1), "mask_embed": emb}
        if need_masks:
            out["pred_masks"] = self.predict_masks(emb, mask_features, query_index)
        return out

    @staticmethod
    def predict_masks(mask_embed, mask_features, query_index=None):
        """einsum('btqc,btchw->bqthw') on device-resident features (b must be 1), optionally for a query subset."""
        b, t, q, cm = mask_embed.shape
        assert b == 1
        e = mask_embed[0] if query_index is None else mask_embed[0][:, query_index]
        m = Fn.mask_logits(e.contiguous(), mask_features[0])                       # (t, q', h, w)
        return m.permute(1, 0, 2, 3).unsqueeze(0)
