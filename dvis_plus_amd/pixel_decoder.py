"""MSDeformAttn module + MSDeformAttn pixel decoder — host side of SURVEY.md §8 rows a3, a4, a5, a14.

Mirrors (same constructor arguments, attribute names and ``state_dict`` keys, so reference checkpoints load
with ``strict=True``):
  MSDeformAttn                         mask2former/modeling/pixel_decoder/ops/modules/ms_deform_attn.py:34-125
  MSDeformAttnTransformerEncoderLayer  mask2former/modeling/pixel_decoder/msdeformattn.py:92-131
  MSDeformAttnTransformerEncoder       ibid. :134-161
  MSDeformAttnTransformerEncoderOnly   ibid. :23-89
  MSDeformAttnPixelDecoder             ibid. :164-358
  PositionEmbeddingSine                mask2former/modeling/transformer_decoder/position_encoding.py:12-52

What is different (MI355X-first, same numbers):
  * no silent fallback: the op raises if the HIP library is missing or tensors are not on the GPU
    (the reference's bare ``except`` at ms_deform_attn.py:119 hides every failure);
  * inference fast path: ONE projection produces offsets|logits (weights concatenated once), and the fused
    kernel applies softmax + location arithmetic in LDS — sampling_locations / attention_weights are never
    materialised; the general path (autograd, 4-d reference boxes, padding masks, fp16/bf16/fp64) goes through
    ``MSDeformAttnFunction`` exactly like the reference;
  * sine position embeddings and encoder reference points depend on shapes only and are cached per shape.
"""
import math
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, normal_, xavier_uniform_

from . import functions as Fn
from .d2 import configurable
from .registry import SEM_SEG_HEADS_REGISTRY, ShapeSpec

# DVIS_MSDA_POS=1: add the position embedding's projection inside the MSDA kernel instead of forming src + pos.  Measured
# on MI355X (T=30, 720p): end to end 180.5 vs 179.1 frames/s (+0.8 %), but the MSDA launch itself 1168 vs 1059 us (its
# set-up phase reads 22 MB more through scattered 16-byte loads) — a wash, so the plain form stays the default.
_POS_IN_KERNEL = os.environ.get("DVIS_MSDA_POS", "0") == "1"
# rows of the fused offsets | logits projection permuted into per-head slots (MSDeformAttn._fused_projection)
# 1: slot = [2LP offsets | LP logits | pad]; 2: slot = [LP logits | 2LP offsets | pad] (the logits 16-byte aligned at the slot
# start — round 4's probe of round 3's "+6 %": profiles/r04_msda_slot_orders.txt)
_MSDA_SLOTS = int(os.environ.get("DVIS_MSDA_SLOTS", "0") or 0)
# the value projection written HEAD-MAJOR (M, N, S, D) by the own GEMM's epilogue and gathered from that layout
_MSDA_HM = os.environ.get("DVIS_MSDA_HM", "0") == "1"


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class PositionEmbeddingSine(nn.Module):
    """2-D sine embedding; output depends on (N, H, W) only (mask is always all-False on this path) → cached."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = {}

    def compute(self, h, w, device):
        """(1, 2*num_pos_feats, h, w) — identical for every batch entry."""
        key = (h, w, str(device))
        if key not in self._cache:
            y_embed = torch.arange(1, h + 1, dtype=torch.float32, device=device)[None, :, None].expand(1, h, w)
            x_embed = torch.arange(1, w + 1, dtype=torch.float32, device=device)[None, None, :].expand(1, h, w)
            if self.normalize:
                eps = 1e-6
                y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
                x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
            dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
            dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / self.num_pos_feats)
            pos_x = x_embed[:, :, :, None] / dim_t
            pos_y = y_embed[:, :, :, None] / dim_t
            pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
            pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
            self._cache[key] = torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2).contiguous()
        return self._cache[key]

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("padding masks are not used on the DVIS++ inference path")
        return self.compute(x.shape[-2], x.shape[-1], x.device).expand(x.shape[0], -1, -1, -1)


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, ratio=1.0):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a power-of-2 head dimension (32 / 64) takes the tiled gfx950 kernel; "
                          "other sizes run the generic kernel.")
        self.im2col_step = 128          # accepted for API compatibility, not needed by the HIP op
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._fused = None               # (version key, W_cat, b_cat) for the inference fast path
        self._reset_parameters()

    def _reset_parameters(self):
        """Init rule of ops/modules/ms_deform_attn.py:66-80: offsets and attention logits start input-independent
        (zero weights); head m looks along the direction of angle 2*pi*m/M, stretched to the unit SQUARE, and its p-th
        point sits p + 1 steps out, the same on every level; both projections get Xavier-uniform weights."""
        M, L, P = self.n_heads, self.n_levels, self.n_points
        angle = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
        direction = torch.stack([angle.cos(), angle.sin()], -1)
        direction = direction / direction.abs().max(-1, keepdim=True)[0]                    # (M, 2) on the unit square
        steps = torch.arange(1, P + 1, dtype=torch.float32)                                 # (P,)
        bias = direction[:, None, None, :] * steps[None, None, :, None]                     # (M, 1, P, 2)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(bias.expand(M, L, P, 2).reshape(-1).clone())
            self.sampling_offsets.weight.zero_()
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            for proj in (self.value_proj, self.output_proj):
                xavier_uniform_(proj.weight)
                proj.bias.zero_()

    def _fused_projection(self):
        """-> (weight, bias, head_stride) of the ONE GEMM that produces offsets and logits.  head_stride 0: rows in the
        reference's order [all offsets | all logits | zero rows up to a multiple of 64]; head_stride s: per-head slots
        [head m: 2LP offset rows | LP logit rows | zero rows] of s = (padded width) / M output columns — the same GEMM
        width (8 heads x 36 = 288 -> 320 = 8 x 40), but a (query, head) pair's parameters are one contiguous 160-byte run."""
        so, aw = self.sampling_offsets, self.attention_weights
        key = (so.weight._version, so.bias._version, aw.weight._version, aw.bias._version, so.weight.device, _MSDA_SLOTS)
        if self._fused is None or self._fused[0] != key:
            M, LP = self.n_heads, self.n_levels * self.n_points
            width = -(-(3 * M * LP) // 64) * 64
            # zero rows up to a multiple of 64 output columns: the library's GEMM for (579 600 x 256) x (256 x 288) runs
            # at 97 TFLOP/s, for 320 columns at 118 (974 -> 829 us per encoder layer at 30 frames of 720p)
            slot = width // M if _MSDA_SLOTS and width % M == 0 and (width // M) % 4 == 0 and width // M >= 3 * LP else 0
            if slot:
                C = so.weight.shape[1]
                w = so.weight.new_zeros(M, slot, C)
                b = so.bias.new_zeros(M, slot)
                o0, l0 = (LP, 0) if _MSDA_SLOTS == 2 else (0, 2 * LP)       # where the offsets / the logits start in a slot
                w[:, o0:o0 + 2 * LP] = so.weight.detach().view(M, 2 * LP, C)
                w[:, l0:l0 + LP] = aw.weight.detach().view(M, LP, C)
                b[:, o0:o0 + 2 * LP] = so.bias.detach().view(M, 2 * LP)
                b[:, l0:l0 + LP] = aw.bias.detach().view(M, LP)
                w, b = w.view(M * slot, C), b.view(M * slot)
            else:
                w = torch.cat([so.weight.detach(), aw.weight.detach()], 0)
                b = torch.cat([so.bias.detach(), aw.bias.detach()], 0)
                pad = width - w.shape[0]
                if pad:
                    w = torch.cat([w, w.new_zeros(pad, w.shape[1])], 0)
                    b = torch.cat([b, b.new_zeros(pad)], 0)
            w, b = w.contiguous(), b.contiguous()
            rows = w.shape[0] if slot else 3 * self.n_heads * self.n_levels * self.n_points
            self._fused = (key, w, b, slot, w[:rows], b[:rows])     # [4:]: without the zero rows (csrc/gemm_x3.hip needs none)
        return self._fused[1], self._fused[2], self._fused[3]

    def _fast_path_ok(self, query, reference_points, input_padding_mask):
        d = self.d_model // self.n_heads
        return (not torch.is_grad_enabled() and query.is_cuda and query.dtype in (torch.float32, torch.float16, torch.bfloat16)
                and input_padding_mask is None and reference_points.shape[-1] == 2 and d in (32, 64)
                and (self.n_levels, self.n_points) in ((1, 4), (3, 4), (4, 4)))

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, spatial_shapes_py=None, query_pos=None, post=None):
        """Reference signature + optional extras.  post: (residual, LayerNorm) — return ``norm(residual + forward(...))``,
        the encoder layer's next step (msdeformattn.py:124-125), fused into the output projection where that is served;
        (residual, None) — return ``residual + forward(...)`` (the ViT-Adapter extractor's ``query + attn``), the add in the
        projection's epilogue.
        `spatial_shapes_py`: a python copy of the shapes.  When the
        queries are the pixels themselves (encoder self-attention) it lets the kernel give each block an 8x8 pixel
        tile (cache locality); results do not depend on it.
        `query_pos` (second extra): when given, `query` is the query WITHOUT its position embedding ((1, Lq, C), shared by
        the batch); the fast path projects it once (bias-free) and the kernel adds it to the projected rows, so the
        (N, Lq, C) sum `query + query_pos` is never formed.  Other paths add it up front."""
        if post is not None:
            out = self._forward(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                                input_padding_mask, spatial_shapes_py, query_pos, post)
            if isinstance(out, tuple):
                return out[0]
            return out + post[0] if post[1] is None else Fn.add_layer_norm(out, post[0], post[1])
        return self._forward(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                             input_padding_mask, spatial_shapes_py, query_pos, None)

    def _forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                 input_padding_mask, spatial_shapes_py, query_pos, post):
        """-> projected attention output, or a 1-tuple (norm(residual + it),) when `post` was fused."""
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        M, L, P = self.n_heads, self.n_levels, self.n_points
        fast = self._fast_path_ok(query, reference_points, input_padding_mask)
        # half-precision storage (a .half() / .bfloat16() module, or fp32 parameters under torch.autocast — how the reference
        # evaluates, train_net_video.py:259): the projections come out in fp16 / bf16 and the fused kernel takes them as they
        # are (dvis_msda_fused_forward_h); the layout experiments below are fp32-only
        lowp = query.dtype != torch.float32 or torch.is_autocast_enabled()
        hm = fast and _MSDA_HM and input_flatten.is_contiguous() and not lowp
        # the tall projections on the F16 matrix cores with split fp32 operands (csrc/gemm_x3.hip)
        x3 = fast and Fn.x3_on() and not lowp and not hm and not _MSDA_SLOTS and Fn.x3_ok(input_flatten, self.d_model, self.d_model) \
            and Fn.x3_ok(query, 3 * M * L * P, self.d_model)
        if x3:
            # (Fn.linear: the tiled split-f16 kernel from K = 512 on — the ViT-Adapter extractors' d_model = 1024 — else the streaming one)
            value = Fn.linear(input_flatten, self.value_proj.weight, self.value_proj.bias, tall=True).view(N, Len_in, M, -1)
        elif hm:
            # own GEMM with a head-major epilogue: value[m, n, s, :] — neighbouring pixels of a head are adjacent lines
            value = Fn.gemm_nt(input_flatten.view(N * Len_in, self.d_model), self.value_proj.weight.detach(),
                               self.value_proj.bias.detach(), head_major=self.d_model // M).view(M, N, Len_in, -1)
        else:
            value = Fn.linear(input_flatten, self.value_proj.weight, self.value_proj.bias, tall=True)   # exact own GEMM (one-wave family)
            if input_padding_mask is not None:
                value = value.masked_fill(input_padding_mask[..., None], float(0))
            value = value.view(N, Len_in, M, self.d_model // M)
        if fast:
            w, b, slot = self._fused_projection()
            # where a row's offsets / logits start (slots: inside the head's slot)
            o_off, l_off = ((L * P, 0) if _MSDA_SLOTS == 2 else (0, 2 * L * P)) if slot else (0, M * L * P * 2)
            po = pl = None
            if lowp and slot:
                raise RuntimeError("DVIS_MSDA_SLOTS is an fp32-only layout experiment")
            if query_pos is not None and query_pos.shape[0] == 1 and _POS_IN_KERNEL and not lowp:
                pp = Fn.linear(query_pos[0], w)                                    # (Lq, 3*M*L*P): tiny, once per call
                po, pl = pp[:, o_off:], pp[:, l_off:]
            elif query_pos is not None and not (x3 and query_pos.shape[0] == 1 and query.dim() == 3 and self.d_model == 256):
                query = query + query_pos
                query_pos = None
            if x3 and query_pos is not None and po is None:
                # `with_pos_embed(src, pos)` inside the projection kernel: the (N, Lq, C) sum is never written
                proj = Fn.x3_linear(query, self._fused[4], self._fused[5], xadd=query_pos).view(N * Len_q, -1)
            elif x3:
                proj = Fn.x3_linear(query.reshape(N * Len_q, self.d_model), self._fused[4], self._fused[5])
            else:
                proj = Fn.linear(query.reshape(N * Len_q, self.d_model), w, b, tall=True)     # offsets | logits in one GEMM
            ref = reference_points if reference_points.is_contiguous() else reference_points.contiguous()
            output = Fn.msda_fused_forward(value, input_spatial_shapes, input_level_start_index, ref,
                                           proj[:, o_off:], proj[:, l_off:], L, P, shapes_host=spatial_shapes_py,
                                           pos_offsets=po, pos_logits=pl, head_stride=slot, value_head_major=hm)
            if x3 and post is not None and post[1] is None and post[0].shape[:-1] == output.shape[:-1] and post[0].dtype == torch.float32:
                return (Fn.linear(output, self.output_proj.weight, self.output_proj.bias, tall=True, residual=post[0]),)
            if x3 and post is not None and post[1] is not None and Fn.x3_ok(output, self.d_model, self.d_model, ln=True) \
                    and post[0].shape == output.shape and post[0].dtype == torch.float32 and post[1].weight is not None:
                return (Fn.x3_linear_ln(output, self.output_proj.weight, self.output_proj.bias, post[0], post[1]),)
            if x3:
                return Fn.linear(output, self.output_proj.weight, self.output_proj.bias, tall=True)
            return Fn.linear(output, self.output_proj.weight, self.output_proj.bias, tall=True)
        if query_pos is not None:
            query = query + query_pos
        sampling_offsets = self.sampling_offsets(query).view(N, Len_q, M, L, P, 2)
        attention_weights = self.attention_weights(query).view(N, Len_q, M, L * P)
        attention_weights = F.softmax(attention_weights, -1).view(N, Len_q, M, L, P)
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            sampling_locations = reference_points[:, :, None, :, None, :] \
                + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        if value.device.type == "cpu":
            # CPU tensors: the reference's torch formulation under its own name (ms_deform_attn_func.py:52-72; BASELINE config #1).
            # Chosen by device — the reference gets there through a bare `except` around the CUDA op (ms_deform_attn.py:116-121),
            # for GPU tensors too; here a GPU tensor takes the HIP op below or raises.
            output = Fn.ms_deform_attn_core_pytorch(value, input_spatial_shapes, sampling_locations, attention_weights)
        else:
            output = Fn.MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                                   sampling_locations.contiguous(), attention_weights.contiguous(),
                                                   self.im2col_step)
        return self.output_proj(output)


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError("only relu is used by the DVIS++ configs")
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout_p = dropout     # inference path: dropout is the identity

    def pos_in_projection(self, src, pos):
        """True when `with_pos_embed(src, pos)` is formed inside the offsets | logits projection kernel (dvis_x3_linear_add): no
        layer then needs the (N, S, C) tensor src + pos, neither from the previous layer nor from maps_to_tokens."""
        a = self.self_attn
        return bool(Fn.x3_on() and pos is not None and pos.dim() == 3 and pos.shape[0] == 1 and src.dim() == 3 and src.is_cuda
                    and src.dtype == torch.float32 and not torch.is_grad_enabled() and not torch.is_autocast_enabled()
                    and not _POS_IN_KERNEL and not _MSDA_SLOTS and not _MSDA_HM and a.d_model == 256
                    and 3 * a.n_heads * a.n_levels * a.n_points in (128, 192, 256, 288) and a.d_model // a.n_heads in (32, 64)
                    and (a.n_levels, a.n_points) in ((1, 4), (3, 4), (4, 4)))

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None,
                shapes_py=None, query=None, emit_next_query=False):
        """Reference signature + three optional extras (shapes_py: see MSDeformAttn.forward).  `query`: src + pos when
        the previous layer already wrote it; emit_next_query: return (out, out + pos), the sum written by the final
        add+LayerNorm kernel — the encoder then never runs an add pass for `with_pos_embed` after the first layer."""
        pos_in_proj = self.pos_in_projection(src, pos)
        if query is not None and not pos_in_proj:
            src = self.self_attn(query, reference_points, src, spatial_shapes, level_start_index, padding_mask,
                                 spatial_shapes_py=shapes_py, post=(src, self.norm1))
        else:
            src = self.self_attn(src, reference_points, src, spatial_shapes, level_start_index, padding_mask,
                                 spatial_shapes_py=shapes_py, query_pos=pos, post=(src, self.norm1))
        if Fn.x3_on() and Fn.x3_ffn_ok(src, self.linear1, self.linear2) and self.norm2.weight is not None:
            # linear1 -> ReLU -> linear2 -> + src -> norm2 (-> + pos) in one kernel; the hidden tensor stays on chip
            with_pos = emit_next_query and pos is not None and pos.shape[0] == 1 and src.dim() == 3 and not pos_in_proj
            r = Fn.x3_ffn_ln(src, self.linear1, self.linear2, self.norm2, pos=pos if with_pos else None)
            return r if with_pos or not emit_next_query else (r, None)
        src2 = Fn.linear(Fn.linear_relu(src, self.linear1, tall=True), self.linear2.weight, self.linear2.bias, tall=True)
        if emit_next_query and pos is not None and pos.shape[0] == 1:
            return Fn.add_layer_norm(src2, src, self.norm2, pos=pos)
        out = Fn.add_layer_norm(src2, src, self.norm2)
        return (out, None) if emit_next_query else out


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer_factory, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([encoder_layer_factory() for _ in range(num_layers)])
        self.num_layers = num_layers
        self._ref_cache = {}

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        pts = []
        for lvl, (H_, W_) in enumerate(spatial_shapes):
            H_, W_ = int(H_), int(W_)
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device),
                                          indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            pts.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(pts, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def reference_points_unpadded(self, shapes_py, device):
        """valid_ratios == 1 (no padding masks on this path): shape-only, batch-independent, cached. (1, S, L, 2)"""
        key = (tuple(shapes_py), str(device))
        if key not in self._ref_cache:
            vr = torch.ones(1, len(shapes_py), 2, device=device)
            self._ref_cache[key] = self.get_reference_points(shapes_py, vr, device).contiguous()
        return self._ref_cache[key]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios=None, pos=None, padding_mask=None,
                shapes_py=None, query0=None):
        if valid_ratios is None:
            reference_points = self.reference_points_unpadded(shapes_py, src.device)
        else:
            reference_points = self.get_reference_points(spatial_shapes.tolist(), valid_ratios, src.device)
        output, query = src, query0            # query0: src + pos when the caller already has it
        for i, layer in enumerate(self.layers):
            last = i + 1 == len(self.layers)
            r = layer(output, pos, reference_points, spatial_shapes, level_start_index, padding_mask,
                      shapes_py=shapes_py, query=query, emit_next_query=not last)
            output, query = (r, None) if last else r
        return output


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024, dropout=0.1,
                 activation="relu", num_feature_levels=4, enc_n_points=4):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.encoder = MSDeformAttnTransformerEncoder(
            lambda: MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation,
                                                        num_feature_levels, nhead, enc_n_points),
            num_encoder_layers)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self._shape_cache = {}
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    def _shape_tensors(self, shapes_py, device):
        key = (tuple(shapes_py), str(device))
        if key not in self._shape_cache:
            s = torch.as_tensor(shapes_py, dtype=torch.long, device=device)
            lsi = torch.cat((s.new_zeros((1,)), s.prod(1).cumsum(0)[:-1]))
            self._shape_cache[key] = (s, lsi)
        return self._shape_cache[key]

    def forward(self, srcs, pos_embeds, affines=None):
        """srcs: per level (N, C, H, W); pos_embeds: per level (1|N, C, H, W).  No padding (masks all False).
        affines: per level None or the (scale, shift) of a GroupNorm still to be applied to the map
        (Fn.group_norm_affine) — done while the map is transposed into tokens."""
        shapes_py = [(int(s.shape[2]), int(s.shape[3])) for s in srcs]
        lvl_pos = torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[lvl].view(1, 1, -1)
                             for lvl, p in enumerate(pos_embeds)], 1)
        query0 = None
        layer0 = self.encoder.layers[0] if len(self.encoder.layers) else None
        probe = srcs[0].new_empty((1, 1, srcs[0].shape[1]))            # (shape / dtype / device of the token matrix)
        if layer0 is not None and layer0.pos_in_projection(probe, lvl_pos):
            src_flatten = Fn.maps_to_tokens(srcs, affines)              # src + pos is formed inside the projection kernels
        elif lvl_pos.shape[0] == 1 and srcs[0].is_cuda and not torch.is_grad_enabled():
            src_flatten, query0 = Fn.maps_to_tokens(srcs, affines, pos=lvl_pos)   # tokens and tokens + pos in one pass
        else:
            src_flatten = Fn.maps_to_tokens(srcs, affines)
        spatial_shapes, level_start_index = self._shape_tensors(shapes_py, src_flatten.device)
        memory = self.encoder(src_flatten, spatial_shapes, level_start_index, None, lvl_pos, None, shapes_py=shapes_py,
                              query0=query0)
        return memory, spatial_shapes, level_start_index, shapes_py


class ConvNorm(nn.Conv2d):
    """Conv2d -> norm -> activation with detectron2.layers.Conv2d's parameter names (weight, bias, norm.*)."""

    def __init__(self, *args, norm=None, activation=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm, self.activation = norm, activation

    def conv(self, x):
        if self.kernel_size == (1, 1) and self.stride == (1, 1) and self.groups == 1 and self.padding == (0, 0):
            return Fn.conv1x1(x, self.weight, self.bias)
        if self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1) and self.dilation == (1, 1) \
                and self.groups == 1 and x.is_cuda and not torch.is_grad_enabled():
            return Fn.conv3x3_bias_act(x, self.weight, self.bias)   # own Winograd kernel where the shape is served
        return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)

    def conv_and_affine(self, x):
        """(conv(x), (scale, shift)) with forward(x) == conv(x) * scale + shift per plane, for a GroupNorm without
        activation whose application the consumer fuses (Fn.upsample_add); (forward(x), None) when that does not apply."""
        if isinstance(self.norm, nn.GroupNorm) and self.activation is None:
            y = self.conv(x)
            aff = Fn.group_norm_affine(y, self.norm)
            if aff is not None:
                return y, aff
            return self.norm(y), None
        return self.forward(x), None

    def image_input_ok(self, N, C, H, W, device):
        """May the input arrive as an operand image (Fn.upsample_add_image -> Fn.conv_x3_image)?  The plain 3x3 / 1x1 forms only."""
        return self.stride == (1, 1) and self.dilation == (1, 1) and self.groups == 1 and self.weight.dtype == torch.float32 \
            and ((self.kernel_size == (3, 3) and self.padding == (1, 1)) or (self.kernel_size == (1, 1) and self.padding == (0, 0))) \
            and C == self.in_channels and Fn.x3_images_ok(N, C, self.out_channels, H, W, device, taps=self.kernel_size[0] ** 2)

    def forward(self, x):
        if isinstance(x, Fn.OperandImage):
            x = Fn.conv_x3_image(x, self.weight, self.bias)
        else:
            x = self.conv(x)
        if isinstance(self.norm, nn.GroupNorm) and self.activation in (None, F.relu):
            # statistics in one read, normalisation (+ReLU) in one in-place pass (torch: moments, apply, clamp)
            aff = Fn.group_norm_affine(x, self.norm)
            if aff is not None:
                return Fn.scale_shift_act_(x, aff[0], aff[1], relu=self.activation is not None)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    if norm != "GN":
        raise NotImplementedError(f"norm {norm!r}: the DVIS++ configs use GN")
    return nn.GroupNorm(32, out_channels)


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


@SEM_SEG_HEADS_REGISTRY.register()
class MSDeformAttnPixelDecoder(nn.Module):
    @configurable
    def __init__(self, input_shape, *, transformer_dropout, transformer_nheads, transformer_dim_feedforward,
                 transformer_enc_layers, conv_dim, mask_dim, norm=None, transformer_in_features, common_stride):
        super().__init__()
        transformer_input_shape = {k: v for k, v in input_shape.items() if k in transformer_in_features}
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        self.in_features = [k for k, v in input_shape]
        self.feature_strides = [v.stride for k, v in input_shape]
        self.feature_channels = [v.channels for k, v in input_shape]
        transformer_input_shape = sorted(transformer_input_shape.items(), key=lambda x: x[1].stride)
        self.transformer_in_features = [k for k, v in transformer_input_shape]
        transformer_in_channels = [v.channels for k, v in transformer_input_shape]
        self.transformer_feature_strides = [v.stride for k, v in transformer_input_shape]
        self.transformer_num_feature_levels = len(self.transformer_in_features)
        chans = transformer_in_channels[::-1] if self.transformer_num_feature_levels > 1 \
            else [transformer_in_channels[-1]]
        self.input_proj = nn.ModuleList([
            nn.Sequential(nn.Conv2d(c, conv_dim, kernel_size=1), nn.GroupNorm(32, conv_dim)) for c in chans])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dim, dropout=transformer_dropout, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_encoder_layers=transformer_enc_layers,
            num_feature_levels=self.transformer_num_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dim // 2, normalize=True)
        self.mask_dim = mask_dim
        self.mask_features = ConvNorm(conv_dim, mask_dim, kernel_size=1, stride=1, padding=0)
        c2_xavier_fill(self.mask_features)
        self.maskformer_num_feature_levels = 3
        self.common_stride = common_stride
        stride = min(self.transformer_feature_strides)
        self.num_fpn_levels = int(math.log2(stride) - math.log2(self.common_stride))
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(self.feature_channels[:self.num_fpn_levels]):
            lateral_conv = ConvNorm(in_channels, conv_dim, kernel_size=1, bias=use_bias, norm=get_norm(norm, conv_dim))
            output_conv = ConvNorm(conv_dim, conv_dim, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                   norm=get_norm(norm, conv_dim), activation=F.relu)
            c2_xavier_fill(lateral_conv)
            c2_xavier_fill(output_conv)
            self.add_module("adapter_{}".format(idx + 1), lateral_conv)
            self.add_module("layer_{}".format(idx + 1), output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]

    @classmethod
    def from_config(cls, cfg, input_shape):
        hd = cfg.MODEL.SEM_SEG_HEAD
        return dict(
            input_shape={k: v for k, v in input_shape.items() if k in hd.IN_FEATURES},
            conv_dim=hd.CONVS_DIM, mask_dim=hd.MASK_DIM, norm=hd.NORM,
            transformer_dropout=cfg.MODEL.MASK_FORMER.DROPOUT, transformer_nheads=cfg.MODEL.MASK_FORMER.NHEADS,
            transformer_dim_feedforward=1024, transformer_enc_layers=hd.TRANSFORMER_ENC_LAYERS,
            transformer_in_features=hd.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES, common_stride=hd.COMMON_STRIDE)

    def forward_features(self, features):
        """features: dict name -> (N, C, H, W).  fp32 island like the reference (msdeformattn.py:314-320).
        Returns (mask_features, out[0], multi_scale_features[:3])."""
        with torch.autocast(device_type="cuda", enabled=False), Fn.x3_stage("pd_proj"):
            srcs, pos, affines = [], [], []
            for idx, f in enumerate(self.transformer_in_features[::-1]):
                x = features[f].float().contiguous()      # (a backbone may hand over channels-last strided maps)
                conv, gn = self.input_proj[idx][0], self.input_proj[idx][1]
                # (nn.Conv2d's own forward is the library convolution; Fn.conv1x1 = the 1x1 kernels, torch ops on the CPU)
                s_ = Fn.conv1x1(x, conv.weight, conv.bias) if conv.kernel_size == (1, 1) and conv.stride == (1, 1) \
                    and conv.padding == (0, 0) and conv.groups == 1 else conv(x)
                affine = Fn.group_norm_affine(s_, gn)        # GroupNorm applied while the map is laid down as tokens
                srcs.append(s_ if affine is not None else gn(s_))
                affines.append(affine)
                pos.append(self.pe_layer.compute(x.shape[2], x.shape[3], x.device))
            with Fn.x3_stage("encoder"):
                y, _, _, shapes_py = self.transformer(srcs, pos, affines)
            bs = y.shape[0]
            out, tokens, start = [], [], 0
            for li, (h, w) in enumerate(shapes_py):
                tokens.append(y[:, start:start + h * w])
                if li + 1 == len(shapes_py) and self.num_fpn_levels > 0:
                    out.append(Fn.tokens_to_map(y, start, h, w))    # the FPN's top-down path reads this one: a real map
                else:
                    out.append(tokens[-1].transpose(1, 2).reshape(bs, -1, h, w))  # a strided view, no copy
                start += h * w
            with Fn.x3_stage("mask_path"):        # lateral 1x1 -> top-down add -> 3x3 output conv -> mask_features 1x1
                for idx, f in enumerate(self.in_features[:self.num_fpn_levels][::-1]):
                    x = features[f].float().contiguous()      # NCHW for the fused FPN path (no-op for the R50's maps)
                    cur_fpn, affine = self.lateral_convs[idx].conv_and_affine(x)    # GroupNorm applied inside upsample_add
                    oc = self.output_convs[idx]
                    if cur_fpn.is_cuda and cur_fpn.dtype == torch.float32 and cur_fpn.is_contiguous() and out[-1].dtype == torch.float32 \
                            and oc.image_input_ok(*cur_fpn.shape, cur_fpn.device):
                        # the top-down sum as an operand image: the 3x3 output convolution reads pre-split fragments (csrc/conv1x1_x3.hip)
                        out.append(oc(Fn.upsample_add_image(cur_fpn, out[-1], affine)))
                    else:
                        out.append(oc(Fn.upsample_add(cur_fpn, out[-1], affine)))
                multi_scale_features = TokenMaps(out[:self.maskformer_num_feature_levels])
                multi_scale_features.tokens = tokens[:self.maskformer_num_feature_levels]
                return self.mask_features(out[-1]), out[0], multi_scale_features


class TokenMaps(list):
    """The multi-scale feature maps as the reference returns them — a list of (N, C, h, w) tensors — plus `.tokens`:
    the same data as (N, h*w, C) row-major views of the encoder's output memory.  The masked-attention decoder flattens
    and transposes every map again (video_mask2former_transformer_decoder.py:267-275); with the tokens at hand it
    projects K / V straight from the encoder's layout (no NCHW round trip)."""
    tokens = None


def r50_input_shape():
    return {"res2": ShapeSpec(channels=256, stride=4), "res3": ShapeSpec(channels=512, stride=8),
            "res4": ShapeSpec(channels=1024, stride=16), "res5": ShapeSpec(channels=2048, stride=32)}
