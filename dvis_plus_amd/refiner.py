"""Temporal refiner — host side of SURVEY.md §8 row a11.

Mirrors ``TemporalRefiner`` (dvis_Plus/refiner.py:6-227): constructor arguments, ``state_dict`` keys, call signature
and returned dict (eval branch: last layer only, window-free).

MI355X re-organisation:
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
from .graphs import FusedKV, GraphRunner
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
        self.use_graphs = True
        self._graph = GraphRunner(self.refine)

    def _kv_weights(self):
        return self._kv_cache.get(self.transformer_cross_attention_layers, self.decoder_norm.weight.shape[0])

    def refine(self, instance_embeds, frame_embeds):
        """The 6 refinement layers.  (b, c, t, q) x2 -> last layer's queries (t, q, b, c), un-normed."""
        B, C, T, Q = instance_embeds.shape
        output = instance_embeds
        fe = frame_embeds.permute(3, 0, 2, 1).flatten(1, 2)                        # (q, bt, c)
        W, b = self._kv_weights()
        kv = F.linear(fe, W, b)                                                    # (q, bt, layers * 2C)
        for i in range(self.num_layers):
            output = output.permute(2, 0, 3, 1).flatten(1, 2)                      # (t, bq, c)
            output = self.transformer_time_self_attention_layers[i](output)
            output = output.permute(1, 2, 0)                                       # (bq, c, t)
            output = self.conv_norms[i](
                (self.conv_short_aggregate_layers[i](output) + output).transpose(1, 2)).transpose(1, 2)
            output = output.reshape(B, Q, C, T).permute(1, 0, 3, 2).flatten(1, 2)  # (q, bt, c)
            output = self.transformer_obj_self_attention_layers[i](output)
            layer = self.transformer_cross_attention_layers[i]
            output = layer.attend(output, output, kv[..., (2 * i) * C:(2 * i + 1) * C],
                                  kv[..., (2 * i + 1) * C:(2 * i + 2) * C])
            output = self.transformer_ffn_layers[i](output)
            output = output.reshape(Q, B, T, C).permute(1, 3, 2, 0)                # (b, c, t, q)
        return output.permute(2, 3, 0, 1)                                          # (t, q, b, c)

    def pred_class(self, decoder_output):
        """(l, b, t, q, c): softmax-over-time pooled class logits, repeated T times (refiner.py:196-210)."""
        T = decoder_output.size(2)
        activation = self.activation_proj(decoder_output).softmax(dim=2)
        class_output = (decoder_output * activation).sum(dim=2, keepdim=True).repeat(1, 1, T, 1, 1)
        return self.class_embed(class_output).transpose(2, 3)

    def forward(self, instance_embeds, frame_embeds, mask_features, need_masks=True, query_index=None):
        """instance_embeds / frame_embeds (b, c, t, q), mask_features (b, t, c, h, w) device-resident.
        Returns pred_logits (b,t,q,K+1), pred_masks (b,q',t,h,w) [q' = len(query_index) if given] or None,
        pred_embds (b,c,t,q), mask_embed (b,t,q,Cm)."""
        if self.training:
            raise NotImplementedError("dvis_plus_amd implements the refiner's inference path")
        self._kv_weights()
        self._graph.enabled = self.use_graphs
        last = self._graph("refine", instance_embeds.contiguous(), frame_embeds.contiguous()).clone()   # (t, q, b, c)
        dec = self.decoder_norm(last)
        dec_b = dec.permute(2, 0, 1, 3)                                            # (b, t, q, c)
        emb = self.mask_embed(dec_b)                                               # (b, t, q, Cm)
        logits = self.pred_class(dec_b[None])[0].transpose(1, 2)                   # (b, t, q, K+1)
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
