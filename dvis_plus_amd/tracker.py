"""Referring tracker — host side of SURVEY.md §8 rows a9, a10.

Mirrors ``ReferringCrossAttentionLayer`` / ``ReferringTracker_noiser`` (dvis_Plus/tracker.py:8-380) and the eval
branch of ``Noiser`` (dvis_Plus/noiser.py:43-77): same constructor arguments, ``state_dict`` keys, call signature,
returned dict and cross-call state (``resume`` continues a video).

The reference walks the clip frame by frame and, per frame, syncs the device for a scipy assignment, recomputes
six K/V projections and (offline) a mask einsum whose result is thrown away.  Here:
  * the assignment chain depends only on the segmenter's embeddings, so ALL cosine-cost matrices of a clip are one
    batched GEMM; one D2H copy feeds the host solver (libdvis_hip's Jonker-Volgenant, bit-identical to scipy on the
    golden vectors) which runs the T-step recurrence in C++ and returns every frame's permutation: one sync per
    clip instead of one per frame;
  * K / V projections of the 6 cross-attention layers for all T frames are one GEMM;
  * the recurrence itself (what genuinely is sequential) runs on the fp32-MFMA attention kernel;
  * masks are only contracted when asked for (online mode); offline mode never needs them
    (dvis_Plus/meta_architecture.py:1486 deletes them).
"""
import ctypes

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import functions as Fn
from . import native
from .graphs import FusedKV, GraphRunner
from .pixel_decoder import c2_xavier_fill
from .transformer_decoder import MLP, FFNLayer, SelfAttentionLayer, _xavier_


def use_own_gemm(module):
    """Every projection under `module` runs on the deterministic own GEMM (csrc/gemm.hip): the tracker / refiner stream
    of DVIS_Plus_offline.stream() must never carry a library stream-K kernel, and their results must not depend on the
    library's run-to-run summation order."""
    for m in module.modules():
        if hasattr(m, "own_gemm"):
            m.own_gemm = True


class ReferringCrossAttentionLayer(nn.Module):
    """Cross-attention whose residual comes from `indentify` instead of the query (tracker.py:35-53)."""
    own_gemm = None

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        if normalize_before:
            raise NotImplementedError("normalize_before=True is not used by DVIS++")
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.norm = nn.LayerNorm(d_model)
        self.nhead = nhead
        _xavier_(self)

    def attend(self, indentify, tgt, k_proj, v_proj, q_proj=None):
        """q_proj: the already projected query (the tracker projects one reference for all its layers in one GEMM)."""
        q = q_proj
        if q is None:
            C = tgt.shape[-1]
            q = Fn.linear(tgt, self.multihead_attn.in_proj_weight[:C], self.multihead_attn.in_proj_bias[:C],
                          own=self.own_gemm)
        att = Fn.attention(q, k_proj, v_proj, self.nhead)
        op = self.multihead_attn.out_proj
        return Fn.add_layer_norm(Fn.linear(att, op.weight, op.bias, own=self.own_gemm), indentify, self.norm)

    def forward(self, indentify, tgt, key, memory, memory_mask=None, memory_key_padding_mask=None, pos=None,
                query_pos=None):
        assert memory_mask is None and memory_key_padding_mask is None and pos is None and query_pos is None
        C = tgt.shape[-1]
        W, b = self.multihead_attn.in_proj_weight, self.multihead_attn.in_proj_bias
        return self.attend(indentify, tgt, Fn.linear(key, W[C:2 * C], b[C:2 * C], own=self.own_gemm),
                           Fn.linear(memory, W[2 * C:], b[2 * C:], own=self.own_gemm))


def match_chain(cost):
    """cost (T, Q, Q) fp32 GPU/CPU tensor (see dvis_match_chain) -> int64 numpy (T, Q).  ONE device->host copy."""
    c = cost.detach().to("cpu", torch.float32).contiguous().numpy()
    T, Q, _ = c.shape
    out = np.empty((T, Q), dtype=np.int64)
    rc = native.lib().dvis_match_chain(c.ctypes.data_as(ctypes.c_void_p), T, Q, out.ctypes.data_as(ctypes.c_void_p))
    native.check(rc, "dvis_match_chain")
    return out


def match_chains(costs):
    """costs (B, T, Q, Q): the chains of B independent clips.  ONE device->host copy, B solver calls -> (B, T, Q)."""
    c = costs.detach().to("cpu", torch.float32).contiguous().numpy()
    B, T, Q, _ = c.shape
    out = np.empty((B, T, Q), dtype=np.int64)
    for b in range(B):
        rc = native.lib().dvis_match_chain(c[b].ctypes.data_as(ctypes.c_void_p), T, Q, out[b].ctypes.data_as(ctypes.c_void_p))
        native.check(rc, "dvis_match_chain")
    return out


def cosine_costs(cur, ref_first):
    """cur (T, Q, C) embeddings of the clip, ref_first (Q, C) what frame 0 is matched against.
    -> (T, Q, Q) with cost[i] = 1 - norm(cur_i) @ norm(ref_i)^T, ref_i = cur_{i-1} (un-permuted) for i > 0.
    Normalisation as Noiser.match_embds: x / (||x|| + 1e-6)."""
    nrm = cur / (cur.norm(dim=-1, keepdim=True) + 1e-6)
    r0 = ref_first / (ref_first.norm(dim=-1, keepdim=True) + 1e-6)
    ref = torch.cat([r0[None], nrm[:-1]], 0)
    if cur.is_cuda and cur.dtype == torch.float32 and cur.shape[-1] % 4 == 0 and not torch.is_grad_enabled():
        return 1 - Fn.bmm_nt(nrm, ref)                                        # own deterministic GEMM (no library kernel)
    return 1 - torch.bmm(nrm, ref.transpose(1, 2))


class ReferringTracker_noiser(nn.Module):
    def __init__(self, hidden_channel=256, feedforward_channel=2048, num_head=8, decoder_layer_num=6, mask_dim=256,
                 class_num=25, noise_mode="hard", noise_ratio=0.5):
        super().__init__()
        self.num_heads, self.num_layers = num_head, decoder_layer_num
        self.transformer_self_attention_layers = nn.ModuleList()
        self.transformer_cross_attention_layers = nn.ModuleList()
        self.transformer_ffn_layers = nn.ModuleList()
        for _ in range(self.num_layers):
            self.transformer_self_attention_layers.append(
                SelfAttentionLayer(d_model=hidden_channel, nhead=num_head, dropout=0.0, normalize_before=False))
            self.transformer_cross_attention_layers.append(
                ReferringCrossAttentionLayer(d_model=hidden_channel, nhead=num_head, dropout=0.0,
                                             normalize_before=False))
            self.transformer_ffn_layers.append(
                FFNLayer(d_model=hidden_channel, dim_feedforward=feedforward_channel, dropout=0.0,
                         normalize_before=False))
        self.use_memory = False
        self.decoder_norm = nn.LayerNorm(hidden_channel)
        self.class_embed = nn.Linear(2 * hidden_channel, class_num + 1)
        self.mask_embed = MLP(hidden_channel, hidden_channel, mask_dim, 3)
        self.ref_proj = MLP(hidden_channel, hidden_channel, hidden_channel, 3)
        for layer in self.ref_proj.layers:
            c2_xavier_fill(layer)
        self.mask_feature_proj = nn.Conv2d(mask_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        self.last_outputs = None         # (1 + layers, q, b, c) in the reference; only [-1] is ever read -> (q, b, c)
        self.last_frame_embeds = None
        self.last_reference = None
        self.noise_mode, self.noise_ratio = noise_mode, noise_ratio   # training-only knobs (kept for the ctor surface)
        self._kv_cache = FusedKV()
        self._q_cache = FusedKV("q")
        self.use_graphs = True
        self._graph = GraphRunner(self._recurrence_entry)
        use_own_gemm(self)

    def _clear_memory(self):
        self.last_outputs = None
        self.last_reference = None

    def _kv_weights(self):
        return self._kv_cache.get(self.transformer_cross_attention_layers, self.decoder_norm.weight.shape[0])

    def _q_weights(self):
        return self._q_cache.get(self.transformer_cross_attention_layers, self.decoder_norm.weight.shape[0])

    def _recurrence_entry(self, fe_nn, idx_dev, last_outputs):
        return self._recurrence(fe_nn, idx_dev, last_outputs, self._rec_first)

    def _recurrence(self, fe_nn, idx_dev, last_outputs, first_is_start):
        """The genuinely sequential part.  fe_nn (T,Q,B,C) un-normed frame queries, idx_dev (T,Q,B) assignments,
        last_outputs (Q,B,C) carried state (ignored when the clip starts a video).  B > 1: B independent clips of equal
        length advance together — every op below is row-wise or per (batch, head), so a clip's rows see the same arithmetic
        as alone, and the ~65 launch-bound kernels per frame are paid once for all B clips.
        Returns (outputs (T,Q,B,C), references (T,Q,B,C), new last_outputs)."""
        T, Q, B, C = fe_nn.shape
        W, b = self._kv_weights()
        with Fn.gemm_sizes_as(rows=T * Q):
            kv = Fn.linear(fe_nn, W, b, own=True)                              # (T, Q, B, layers * 2C): one GEMM
        Wq, bq = self._q_weights()
        outputs, refs = [], []
        gidx = idx_dev[..., None].expand(T, Q, B, C)
        with Fn.gemm_sizes_as(rows=Q):                                         # tile configuration of ONE clip's Q rows
            for i in range(T):
                single_nn = fe_nn[i]                                           # (q, b, c)
                out = torch.gather(single_nn, 0, gidx[i])                      # out[q, b] = single_nn[idx[q, b], b]
                first = i == 0 and first_is_start
                if not first:
                    # the same reference feeds every layer's cross-attention (tracker.py:278, 293-318): its 6 query
                    # projections are ONE GEMM (N = layers * C) instead of six launches in the sequential chain
                    reference = self.ref_proj(last_outputs)
                    q_all = Fn.linear(reference, Wq, bq, own=True)
                for j in range(self.num_layers):
                    ref_j = self.ref_proj(single_nn if j == 0 else out) if first else reference
                    kj = kv[i, :, :, (2 * j) * C:(2 * j + 1) * C]
                    vj = kv[i, :, :, (2 * j + 1) * C:(2 * j + 2) * C]
                    qj = None if first else q_all[..., j * C:(j + 1) * C]
                    out = self.transformer_cross_attention_layers[j].attend(out, ref_j, kj, vj, q_proj=qj)
                    out = self.transformer_self_attention_layers[j](out)
                    out = self.transformer_ffn_layers[j](out)
                refs.append(self.ref_proj(single_nn) if first else reference)
                last_outputs = out
                outputs.append(out)
        return torch.stack(outputs, 0), torch.stack(refs, 0), last_outputs

    def forward(self, frame_embeds, mask_features, resume=False, return_indices=False, frame_classes=None,
                frame_embeds_no_norm=None, need_masks=True):
        """frame_embeds (b, c, t, q); mask_features (b, t, c, h, w) [may be None when need_masks=False].
        Same outputs as the reference (eval): pred_logits (b,t,q,K+1), pred_masks (b,q,t,h,w) | None,
        pred_embds (b,c,t,q), pred_references (b,c,t,q), aux_outputs []."""
        if self.training:
            raise NotImplementedError("dvis_plus_amd implements the tracker's inference path")
        fe = frame_embeds.permute(2, 3, 0, 1)                                  # (t, q, b, c)
        fe_nn = fe if frame_embeds_no_norm is None else frame_embeds_no_norm.permute(2, 3, 0, 1)
        T, Q, B, C = fe.shape
        # The reference runs one video at a time (it matches on batch entry 0 only, noiser.py:43-56).  B > 1 here means B
        # INDEPENDENT clips of equal length that each start a video, advanced together (stream() with tracker_batch > 1):
        # same results per clip, the recurrence's launch-bound kernels paid once.
        assert B == 1 or not resume, "only clips that start a video can share a tracker pass"
        first_is_start = not resume
        if first_is_start:
            self._clear_memory()

        # ---- 1. every frame's assignment: batched cosine costs on the GPU, ONE sync, chains solved on the host
        with Fn.gemm_sizes_as(batch=T):
            costs = torch.stack([cosine_costs(fe[:, :, b, :], fe[0, :, b, :] if first_is_start
                                              else self.last_frame_embeds[:, b, :]) for b in range(B)])
        indices = match_chains(costs)                                          # (B, T, Q) int64, host
        idx_dev = torch.from_numpy(indices).to(fe.device).permute(1, 2, 0).contiguous()    # (T, Q, B)
        self.last_indices = indices[0] if B == 1 else indices                  # this call's assignments (tests)

        # ---- 2 + 3. K / V of all layers for all frames (one GEMM) and the recurrence, replayed from a hipGraph
        self._kv_weights(), self._q_weights()                                  # build the cached weights outside capture
        state = self.last_outputs if not first_is_start else torch.zeros_like(fe_nn[0])
        self._rec_first = first_is_start                                       # part of the graph key: fixes control flow
        self._graph.enabled = self.use_graphs
        outputs, refs, last = self._graph((T, first_is_start), fe_nn.contiguous(), idx_dev, state)
        # carried state = the LAST batch entry's (a later `resume` call continues that video)
        self.last_outputs = last[:, B - 1:].clone()
        self.last_reference = refs[T - 1][:, B - 1:].clone()
        self.last_frame_embeds = torch.gather(fe[T - 1], 0, idx_dev[T - 1][..., None].expand(Q, B, C))[:, B - 1:].clone()
        outputs, refs = outputs.clone(), refs.clone()                          # static graph buffers -> owned tensors

        # ---- 4. heads
        dec = self.decoder_norm(outputs)
        with Fn.gemm_sizes_as(rows=T * Q):
            logits = Fn.linear(torch.cat([refs, dec], dim=-1), self.class_embed.weight, self.class_embed.bias, own=True)  # (t,q,b,K+1)
        out = {
            "pred_logits": logits.permute(2, 0, 1, 3),
            "pred_masks": None,
            "aux_outputs": [],
            "pred_embds": outputs.permute(2, 3, 0, 1),
            "pred_references": refs.permute(2, 3, 0, 1),
        }
        if need_masks:
            assert B == 1, "tracker masks (online mode) are produced one video at a time"
            b_, t_, cm, h, w = mask_features.shape
            mf = self.mask_feature_proj(mask_features.flatten(0, 1))           # (t, cm, h, w); online mode, main stream
            emb = self.mask_embed(dec[:, :, 0, :])                             # (t, q, cm)
            out["pred_masks"] = Fn.mask_logits(emb.contiguous(), mf).permute(1, 0, 2, 3).unsqueeze(0)
        if return_indices:
            return out, [indices[0][i] for i in range(T)]
        return out
