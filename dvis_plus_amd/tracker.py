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
import os

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
        self._kv_cache = FusedKV("k_v")
        self._q_cache = FusedKV("q")
        self._o_cache = FusedKV("out")
        self.use_graphs = True
        # The chain with its LayerNorm seams folded into the consuming projections (csrc/gemm_ln.hip) and the layer-independent
        # cross-attentions hoisted out of the layer loop: 35 instead of ~65 dependent launches per frame.  False (or
        # DVIS_TRACKER_FUSED=0): the layer-by-layer form (one launch per GEMM / attention / add+LayerNorm) — same results to
        # fp32 rounding; kept as the cross-check of the fused form (tests/test_golden_gpu.py runs both).
        self.fused_chain = os.environ.get("DVIS_TRACKER_FUSED", "1") != "0"
        self._graph = GraphRunner(self._recurrence_entry)
        use_own_gemm(self)

    def _clear_memory(self):
        self.last_outputs = None
        self.last_reference = None

    def project_mask_features(self, x):
        """``mask_feature_proj`` (tracker.py:199: a 1 x 1 convolution, 256 -> 256 at the stride-4 map) on the own fp32-MFMA 1 x 1
        kernel with the bias in its epilogue where the shape is served (csrc/conv1x1_mfma.hip) — no MIOpen call in the online
        path either; other shapes / CPU tensors: the library convolution."""
        c = self.mask_feature_proj
        return Fn.conv1x1(x, c.weight, c.bias)

    def _kv_weights(self):
        return self._kv_cache.get(self.transformer_cross_attention_layers, self.decoder_norm.weight.shape[0])

    def _q_weights(self):
        return self._q_cache.get(self.transformer_cross_attention_layers, self.decoder_norm.weight.shape[0])

    def _o_weights(self):
        return self._o_cache.get(self.transformer_cross_attention_layers, self.decoder_norm.weight.shape[0])

    def _recurrence_entry(self, fe_nn, idx_dev, last_outputs):
        return self._recurrence(fe_nn, idx_dev, last_outputs, self._rec_first)

    def _first_frame(self, single_nn, out, kv_i):
        """Frame 0 of a video (tracker.py:221-275): every layer's reference is ref_proj of the layer's own input, so nothing
        hoists; layer by layer, once per video."""
        C, L = single_nn.shape[-1], self.num_layers
        for j in range(L):
            ref_j = self.ref_proj(single_nn if j == 0 else out)
            out = self.transformer_cross_attention_layers[j].attend(out, ref_j, kv_i[..., j * C:(j + 1) * C],
                                                                    kv_i[..., (L + j) * C:(L + j + 1) * C])
            out = self.transformer_self_attention_layers[j](out)
            out = self.transformer_ffn_layers[j](out)
        return out, self.ref_proj(single_nn)

    def _recurrence(self, fe_nn, idx_dev, last_outputs, first_is_start):
        """The genuinely sequential part.  fe_nn (T,Q,B,C) un-normed frame queries, idx_dev (T,Q,B) assignments,
        last_outputs (Q,B,C) carried state (ignored when the clip starts a video).  B > 1: B independent clips of equal
        length advance together — every op below is row-wise or per (batch, head), so a clip's rows see the same arithmetic
        as alone, and the launch-bound kernels of a frame are paid once for all B clips.
        Returns (outputs (T,Q,B,C), references (T,Q,B,C), new last_outputs).

        Per frame i > 0 (tracker.py:276-318) the SAME reference = ref_proj(last output) is the query of all six layers'
        cross-attention and the frame's own queries are their keys / values, so those six attentions and out-projections do
        not depend on the layer chain: ONE attention call over layers x heads = 48 heads and ONE batched GEMM give every
        layer's attention term t_j up front.  What remains sequential per layer is
            x  = LN_cross(x + t_j);  qkv = in_proj(x);  a = attention;  v = x + out_proj(a);  x = LN_self(v);
            h  = relu(linear1(x));   y = x + linear2(h);                x = LN_ffn(y)
        and every LayerNorm of it runs in the prologue of the projection that consumes it (Fn.gemm_ln): five launches per
        layer, 6 + 30 per frame."""
        T, Q, B, C = fe_nn.shape
        L, H = self.num_layers, self.num_heads
        W, b = self._kv_weights()
        with Fn.gemm_sizes_as(rows=T * Q):
            kv = Fn.linear(fe_nn, W, b, own=True)                              # (T, Q, B, 2 L C): K of all layers, then V
        Wq, bq = self._q_weights()
        Wo, bo = self._o_weights()
        cross, selfa, ffns = (self.transformer_cross_attention_layers, self.transformer_self_attention_layers,
                              self.transformer_ffn_layers)
        rp = self.ref_proj.layers
        # the assignment-permuted frame queries of ALL frames (each frame's layer-0 input): one gather, outside the chain
        x0_all = torch.gather(fe_nn, 1, idx_dev[..., None].expand(T, Q, B, C))
        outputs, refs = torch.empty_like(fe_nn), torch.empty_like(fe_nn)
        fused = self.fused_chain and Fn.gemm_ln_ok(fe_nn, rp[0].weight) and len(rp) == 3 \
            and Fn.gemm_ln_ok(fe_nn.new_empty(1, ffns[0].linear2.in_features), ffns[0].linear2.weight, norms=False)
        y_raw = None                 # previous frame's last FFN sum BEFORE its LayerNorm (fused form), else None
        with Fn.gemm_sizes_as(rows=Q):                                         # tile configuration of ONE clip's Q rows
            for i in range(T):
                if i == 0 and first_is_start:
                    last_outputs, ref0 = self._first_frame(fe_nn[0], x0_all[0], kv[0])
                    outputs[0], refs[0] = last_outputs, ref0
                    continue
                if not fused:
                    reference = self.ref_proj(last_outputs)
                    q_all = Fn.linear(reference, Wq, bq, own=True)
                    out = x0_all[i]
                    for j in range(L):
                        out = cross[j].attend(out, reference, kv[i][..., j * C:(j + 1) * C],
                                              kv[i][..., (L + j) * C:(L + j + 1) * C], q_proj=q_all[..., j * C:(j + 1) * C])
                        out = ffns[j](selfa[j](out))
                    outputs[i], refs[i] = out, reference
                    last_outputs = out
                    continue
                # ---- reference and the six layer-independent cross-attention terms
                if y_raw is None:
                    h, _ = Fn.gemm_ln(last_outputs, rp[0].weight, rp[0].bias, relu=True)
                else:                 # ... which also materialises the previous frame's output LN_ffn(y) (a_out)
                    h, _ = Fn.gemm_ln(y_raw, rp[0].weight, rp[0].bias, norm1=ffns[L - 1].norm, relu=True, a_out=outputs[i - 1])
                h, _ = Fn.gemm_ln(h, rp[1].weight, rp[1].bias, relu=True)
                reference, _ = Fn.gemm_ln(h, rp[2].weight, rp[2].bias, out=refs[i])
                q_all = Fn.linear(reference, Wq, bq, own=True)                                     # (Q, B, L C)
                att = Fn.attention(q_all, kv[i][..., :L * C], kv[i][..., L * C:], L * H, short=Q <= 128)   # 48 heads, one call
                t = Fn.gemm_nt_stacked(att.view(Q * B, L * C), Wo, bo).view(L, Q, B, C)            # t_j = out_proj_j(att_j)
                # ---- the layer chain
                y = x0_all[i]
                for j in range(L):
                    sa, ff = selfa[j], ffns[j]
                    qkv, xc = Fn.gemm_ln(y, sa.self_attn.in_proj_weight, sa.self_attn.in_proj_bias,
                                         norm1=ffns[j - 1].norm if j else None, add=t[j], norm2=cross[j].norm)
                    a = Fn.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H, short=Q <= 128)
                    v, _ = Fn.gemm_ln(a, sa.self_attn.out_proj.weight, sa.self_attn.out_proj.bias, res=xc)
                    h, xs = Fn.gemm_ln(v, ff.linear1.weight, ff.linear1.bias, norm1=sa.norm, relu=True)
                    y, _ = Fn.gemm_ln(h, ff.linear2.weight, ff.linear2.bias, res=xs)
                y_raw = y
            if y_raw is not None:
                outputs[T - 1] = Fn.add_layer_norm(y_raw, None, ffns[L - 1].norm)
                last_outputs = outputs[T - 1]
        return outputs, refs, last_outputs

    @Fn.fp32_island
    def forward(self, frame_embeds, mask_features, resume=False, return_indices=False, frame_classes=None,
                frame_embeds_no_norm=None, need_masks=True):
        """frame_embeds (b, c, t, q); mask_features (b, t, c, h, w) [may be None when need_masks=False].
        Same outputs as the reference (eval): pred_logits (b,t,q,K+1), pred_masks (b,q,t,h,w) | None,
        pred_embds (b,c,t,q), pred_references (b,c,t,q), aux_outputs []."""
        if self.training:
            raise NotImplementedError("dvis_plus_amd implements the tracker's inference path")
        frame_embeds, frame_embeds_no_norm, mask_features = Fn.f32(frame_embeds), Fn.f32(frame_embeds_no_norm), Fn.f32(mask_features)
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
        self._kv_weights(), self._q_weights(), self._o_weights()               # build the cached weights outside capture
        state = self.last_outputs if not first_is_start else torch.zeros_like(fe_nn[0])
        self._rec_first = first_is_start                                       # part of the graph key: fixes control flow
        self._graph.enabled = self.use_graphs
        outputs, refs, last = self._graph((T, first_is_start, self.fused_chain), fe_nn.contiguous(), idx_dev, state)
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
            mf = self.project_mask_features(mask_features.flatten(0, 1))       # (t, cm, h, w); online mode, main stream
            emb = self.mask_embed(dec[:, :, 0, :])                             # (t, q, cm)
            out["pred_masks"] = Fn.mask_logits(emb.contiguous(), mf).permute(1, 0, 2, 3).unsqueeze(0)
        if return_indices:
            return out, [indices[0][i] for i in range(T)]
        return out
