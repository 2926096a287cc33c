"""Torch formulations of the four hot ops for CPU TENSORS — BASELINE config #1 read literally ("Mask2Former R50 single 480p
frame, 100 queries, PyTorch CPU MSDeformAttn fallback (plumbing, no GPU)").

The reference ships one such function, ``ms_deform_attn_core_pytorch`` (ops/functions/ms_deform_attn_func.py:52-72), and reaches
it through a bare ``except`` around the CUDA op (ops/modules/ms_deform_attn.py:116-121) — on ANY failure, silently, GPU tensors
included.  Here the dispatch is by device and explicit: a tensor on the CPU takes these functions, a tensor on the GPU takes the
HIP kernel or raises (functions.py: no torch formulation ever runs on a GPU tensor behind the caller's back, a missing / broken
libdvis_hip.so is an error).  Nothing in here is on the measured path, and nothing in here imports ``oracle/`` (test infrastructure).
"""
import torch
import torch.nn.functional as F


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Same name, arguments and result as the reference's torch formulation (ms_deform_attn_func.py:52-72):
    value (N, S, M, D), value_spatial_shapes (L, 2) rows (H_l, W_l), sampling_locations (N, Lq, M, L, P, 2) in [0, 1] (x, y),
    attention_weights (N, Lq, M, L, P) -> (N, Lq, M * D).

    out[n, q, m, :] = sum_{l, p} w[n, q, m, l, p] * bilinear(value_l[n, :, m, :], loc[n, q, m, l, p]) with zero padding and
    pixel centres at (i + 0.5) / size: ``F.grid_sample(align_corners=False)`` on 2 loc - 1 per level.  The levels' terms are
    accumulated one after the other (the reference stacks all L * P samples and reduces once: same sum, other association)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in value_spatial_shapes.tolist()]
    if sum(h * w for h, w in shapes) != S or len(shapes) != L:
        raise RuntimeError("ms_deform_attn_core_pytorch: value rows do not match the spatial shapes")
    grid = 2 * sampling_locations - 1
    out, start = None, 0
    for lvl, (h, w) in enumerate(shapes):
        v = value[:, start:start + h * w].permute(0, 2, 3, 1).reshape(N * M, D, h, w)                  # (N M, D, h, w)
        g = grid[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)                         # (N M, Lq, P, 2)
        s = F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False)           # (N M, D, Lq, P)
        wl = attention_weights[:, :, :, lvl].permute(0, 2, 1, 3).reshape(N * M, 1, Lq, P)
        term = (s * wl).sum(-1)                                                                        # (N M, D, Lq)
        out = term if out is None else out + term
        start += h * w
    return out.view(N, M * D, Lq).transpose(1, 2).contiguous()


def attention(q, k, v, nheads, mask=None, allowed_count=None, out=None):
    """functions.attention for CPU tensors: softmax(q k^T / sqrt(d)) v per head on (L, B, C) tensors; mask (B, Lq, Lk) with
    1 = blocked, rows whose allowed_count is 0 attend everywhere (video_mask2former_transformer_decoder.py:297)."""
    Lq, B, C = q.shape
    Lk, d = k.shape[0], C // nheads
    heads = lambda t, n: t.reshape(n, B, nheads, d).permute(1, 2, 0, 3)                                # (B, H, L, d)
    s = (heads(q, Lq) * (1.0 / d ** 0.5)) @ heads(k, Lk).transpose(-1, -2)
    if mask is not None:
        blocked = mask.bool()
        if allowed_count is not None:
            blocked = blocked & (allowed_count > 0)[..., None]
        s = s.masked_fill(blocked[:, None], float("-inf"))
    o = (torch.softmax(s, -1) @ heads(v, Lk)).permute(2, 0, 1, 3).reshape(Lq, B, C)
    if out is not None:
        out.copy_(o)
        return out
    return o


def mask_logits(mask_embed, mask_features):
    """einsum("bqc,bchw->bqhw") (video_mask2former_transformer_decoder.py:363)."""
    B, Q, C = mask_embed.shape
    H, W = mask_features.shape[-2:]
    return torch.bmm(mask_embed, mask_features.flatten(2)).view(B, Q, H, W)


def attn_mask(mask_embed, mask_features, target_size):
    """functions.attn_mask for CPU tensors (ibid. :363-371): (mask uint8 (B, Q, h w) with 1 = blocked, allowed_count int32 (B, Q))."""
    small = F.interpolate(mask_logits(mask_embed, mask_features), size=(int(target_size[0]), int(target_size[1])), mode="bilinear",
                          align_corners=False)
    blocked = small.sigmoid().flatten(2) < 0.5
    return blocked.to(torch.uint8), (~blocked).sum(-1).to(torch.int32)
