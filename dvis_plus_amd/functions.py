"""Op front-ends over the C ABI: the Python surface the reference's callers bind to.

``ms_deform_attn_forward`` / ``ms_deform_attn_backward`` have the signature of the reference's
pybind exports (ops/src/vision.cpp:18-21, ops/src/ms_deform_attn.h:25-66) and ``MSDeformAttnFunction``
that of ops/functions/ms_deform_attn_func.py:32-49, so ``MSDeformAttn`` / ViT-Adapter call sites work
unchanged.  Differences, all deliberate:
  * any batch size (the ``N % im2col_step`` assertion of ms_deform_attn_cuda.cu:55-57 is not needed;
    ``im2col_step`` is accepted and ignored),
  * fp16 / bf16 tensors are accepted in forward (fp32 accumulation) besides float / double,
  * failures raise ``RuntimeError`` — there is no torch/CPU fallback to hide them.
"""
import ctypes
import math
import os
import threading

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import cpu_ops, native
from .cpu_ops import ms_deform_attn_core_pytorch    # noqa: F401  (the reference's name for the CPU-tensor formulation, ms_deform_attn_func.py:52)


def _torch_path(op, x, why):
    """Called where a fused glue op is about to take its torch formulation instead of the HIP kernel.  That is the right
    thing for CPU tensors (host-logic tests) and under autograd (the kernels are inference-only), and otherwise a shape /
    layout the kernel does not serve.  The latter RAISES (the default) instead of silently running torch ops on the GPU —
    the invisible dual path the reference has in ops/modules/ms_deform_attn.py:116-121.  DVIS_STRICT=0 is the explicit
    opt-out (odd research shapes: the torch formulation then runs, on the GPU)."""
    if os.environ.get("DVIS_STRICT", "1") != "0" and x.is_cuda and not torch.is_grad_enabled():
        raise RuntimeError(f"DVIS_STRICT: {op} would run its torch formulation on a GPU tensor ({why}; shape "
                           f"{tuple(x.shape)}, dtype {x.dtype}, contiguous {x.is_contiguous()})")


class no_autocast:
    """``with no_autocast(): ...`` — torch.autocast switched off (CUDA and CPU) for the enclosed calls: the fp32 island around the
    hand-written fp32 / split-f16 path.  The reference's launcher evaluates under ``with autocast():`` (train_net_video.py:259)
    and only its pixel decoder opts out (msdeformattn.py:314,320); here the WHOLE model does: under the launcher's context the
    results are the ones of the plain call, bit for bit (fp32 storage, within BASELINE's 1e-3 of the fp32 CPU reference — the
    reference's own GPU run is the half-precision one and is further from it)."""

    def __enter__(self):
        import contextlib
        self._stack = contextlib.ExitStack()
        self._stack.enter_context(torch.autocast(device_type="cuda", enabled=False))
        self._stack.enter_context(torch.autocast(device_type="cpu", enabled=False))
        return self

    def __exit__(self, *exc):
        return self._stack.__exit__(*exc)


def fp32_island(fn):
    """Decorator form of ``no_autocast`` for a module's ``forward`` (not for generators: a context must not stay entered
    across a ``yield`` — the consumer's code would run with autocast off)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if not (torch.is_autocast_enabled() or torch.is_autocast_enabled("cpu")):
            return fn(*args, **kwargs)
        with no_autocast():
            return fn(*args, **kwargs)
    return wrapper


def f32(t):
    """A half / double tensor handed over by an autocast region outside the island -> float32 (None and fp32 tensors pass)."""
    return t if t is None or not torch.is_tensor(t) or not t.is_floating_point() or t.dtype == torch.float32 else t.float()


def _check_msda_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a GPU tensor (no CPU implementation; got {t.device})")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64")
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("expected value (N,S,M,D), sampling_loc (N,Lq,M,L,P,2), attn_weight (N,Lq,M,L,P)")
    if not (value.dtype == sampling_loc.dtype == attn_weight.dtype):
        raise RuntimeError("value, sampling_loc and attn_weight must share one dtype")
    N, S, M, D = value.shape
    N2, Lq, M2, L, P, two = sampling_loc.shape
    if (N2, M2, two) != (N, M, 2) or tuple(attn_weight.shape) != (N, Lq, M, L, P) or spatial_shapes.shape != (L, 2) \
            or level_start_index.shape != (L,):
        raise RuntimeError("inconsistent shapes between value / sampling_loc / attn_weight / spatial_shapes")
    return N, S, M, D, L, Lq, P


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=128):
    """-> (N, Lq, M*D).  Drop-in for ``MultiScaleDeformableAttention.ms_deform_attn_forward``."""
    N, S, M, D, L, Lq, P = _check_msda_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        rc = native.lib().dvis_msda_forward(
            native.dtype_code(value), native.dev_ptr(value, "value"), native.dev_ptr(spatial_shapes, "spatial_shapes"),
            native.dev_ptr(level_start_index, "level_start_index"), native.dev_ptr(sampling_loc, "sampling_loc"),
            native.dev_ptr(attn_weight, "attn_weight"), N, S, M, D, L, Lq, P, native.dev_ptr(out, "out"),
            native.stream_ptr(value.device))
    native.check(rc, "dvis_msda_forward")
    return out


# Reproducible MSDeformAttn backward (fp32): 1 = always, 0 = never, default = when torch.use_deterministic_algorithms(True) is set.
MSDA_BWD_DETERMINISTIC = os.environ.get("DVIS_MSDA_BWD_DETERMINISTIC")


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step=128, deterministic=None):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] (ops/src/ms_deform_attn.h:47-66).
    deterministic (fp32): grad_value through 64-bit fixed-point integer atomics (dvis_msda_backward_det) — the same bits every run;
    default: DVIS_MSDA_BWD_DETERMINISTIC, else torch.are_deterministic_algorithms_enabled()."""
    N, S, M, D, L, Lq, P = _check_msda_inputs(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    if not grad_output.is_cuda:
        raise RuntimeError("grad_output must be a GPU tensor")
    grad_output = grad_output.contiguous()
    if grad_output.dtype != value.dtype or grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output must be (N, Lq, M*D) with value's dtype")
    if deterministic is None:
        deterministic = (MSDA_BWD_DETERMINISTIC == "1") if MSDA_BWD_DETERMINISTIC in ("0", "1") \
            else torch.are_deterministic_algorithms_enabled()
    grad_loc = torch.empty_like(sampling_loc)
    grad_w = torch.empty_like(attn_weight)
    if deterministic and value.dtype == torch.float32:
        grad_value = torch.empty_like(value)
        nbytes = native.lib().dvis_msda_backward_det_ws_bytes(N, S, M, D)
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=value.device)
        with torch.cuda.device(value.device):
            rc = native.lib().dvis_msda_backward_det(
                native.dev_ptr(value, "value"), native.dev_ptr(spatial_shapes, "spatial_shapes"),
                native.dev_ptr(level_start_index, "level_start_index"), native.dev_ptr(sampling_loc, "sampling_loc"),
                native.dev_ptr(attn_weight, "attn_weight"), native.dev_ptr(grad_output, "grad_output"),
                N, S, M, D, L, Lq, P, native.dev_ptr(grad_value, "grad_value"), native.dev_ptr(grad_loc, "grad_loc"),
                native.dev_ptr(grad_w, "grad_w"), ctypes.c_void_p(ws.data_ptr()), native.stream_ptr(value.device))
        native.check(rc, "dvis_msda_backward_det")
        return [grad_value, grad_loc, grad_w]
    grad_value = torch.zeros_like(value)
    with torch.cuda.device(value.device):
        rc = native.lib().dvis_msda_backward(
            native.dtype_code(value), native.dev_ptr(value, "value"), native.dev_ptr(spatial_shapes, "spatial_shapes"),
            native.dev_ptr(level_start_index, "level_start_index"), native.dev_ptr(sampling_loc, "sampling_loc"),
            native.dev_ptr(attn_weight, "attn_weight"), native.dev_ptr(grad_output, "grad_output"),
            N, S, M, D, L, Lq, P, native.dev_ptr(grad_value, "grad_value"), native.dev_ptr(grad_loc, "grad_loc"),
            native.dev_ptr(grad_w, "grad_w"), native.stream_ptr(value.device))
    native.check(rc, "dvis_msda_backward")
    return [grad_value, grad_loc, grad_w]


class MSDeformAttnFunction(Function):
    """Same contract as the reference's autograd wrapper (ops/functions/ms_deform_attn_func.py:32-49)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, loc, w = ctx.saved_tensors
        grad_value, grad_loc, grad_w = ms_deform_attn_backward(value, shapes, level_start, loc, w, grad_output,
                                                               ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_w, None


def msda_fused_forward(value, spatial_shapes, level_start_index, reference_points, offsets, logits, n_levels,
                       n_points, shapes_host=None, pos_offsets=None, pos_logits=None, head_stride=0,
                       value_head_major=False):
    """Inference fast path of ``MSDeformAttn.forward`` (ops/modules/ms_deform_attn.py:101-117): fp32, or fp16 / bf16
    storage with fp32 arithmetic (value, offsets, logits and the output in ONE dtype — the module under autocast).

    value (N,S,M,D); reference_points (1|N, Lq, L, 2); ``offsets`` / ``logits`` are 2-D row views
    (N*Lq, >= M*L*P*2) / (N*Lq, >= M*L*P) of the raw linear outputs (row stride may exceed the width, e.g.
    both sliced out of one fused projection).  softmax + location arithmetic happen inside the kernel.
    shapes_host: optional [(H, W), ...] python copy of spatial_shapes (enables 8x8 query tiling for self-attention).
    pos_offsets / pos_logits: optional (Lq, >= M*L*P*2) / (Lq, >= M*L*P) row views with the SAME row stride: the
    bias-free projections of the queries' position embedding, added to the raw rows inside the kernel
    (linear(src + pos) = linear(src) + pos W^T), so the caller projects `src` and never forms `src + pos`.
    head_stride: 0 = the reference's row layout (all heads' offsets, then all heads' logits); s > 0 = per-head SLOTS of s
    floats [2LP offsets | LP logits | pad]: `offsets` then points at a row's first slot and `logits` 2LP floats further.
    value_head_major: `value` is (M, N, S, D) — head outermost — instead of (N, S, M, D).
    """
    if value_head_major:            # (M, N, S, D): head outermost (gemm_nt(head_major=D) of the value projection)
        M, N, S, D = value.shape
    else:
        N, S, M, D = value.shape
    L, P = n_levels, n_points
    nref, Lq = reference_points.shape[0], reference_points.shape[1]
    for name, t in (("value", value), ("reference_points", reference_points)):
        native.dev_ptr(t, name)
    half = value.dtype in (torch.float16, torch.bfloat16)
    for name, t in (("offsets", offsets), ("logits", logits)):
        if not t.is_cuda or t.dtype != value.dtype or t.dim() != 2 or t.stride(1) != 1 or t.shape[0] != N * Lq:
            raise RuntimeError(f"{name} must be a GPU (N*Lq, width) row view of value's dtype with unit inner stride")
    if value.dtype not in (torch.float32, torch.float16, torch.bfloat16) or reference_points.dtype != torch.float32:
        raise RuntimeError("msda_fused_forward: value / offsets / logits fp32, fp16 or bf16 (one dtype); reference points fp32")
    if half:
        # fp16 / bf16 storage (what the projections produce under autocast), fp32 arithmetic: dvis_msda_fused_forward_h
        if head_stride or value_head_major or pos_offsets is not None or pos_logits is not None:
            raise RuntimeError("msda_fused_forward: slots / head-major value / position rows are fp32-only layouts")
        if offsets.shape[1] < M * L * P * 2 or logits.shape[1] < M * L * P or reference_points.shape[2:] != (L, 2):
            raise RuntimeError("msda_fused_forward: inconsistent shapes")
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        hs = None
        if shapes_host is not None:
            hs = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in shapes_host for v in hw])
        with torch.cuda.device(value.device):
            rc = native.lib().dvis_msda_fused_forward_h(
                native.dtype_code(value), native.dev_ptr(value, "value"), native.dev_ptr(spatial_shapes, "spatial_shapes"),
                native.dev_ptr(level_start_index, "level_start_index"), native.dev_ptr(reference_points, "ref"), nref,
                ctypes.c_void_p(offsets.data_ptr()), offsets.stride(0), ctypes.c_void_p(logits.data_ptr()), logits.stride(0),
                N, S, M, D, L, Lq, P, native.dev_ptr(out, "out"), hs, native.stream_ptr(value.device))
        native.check(rc, "dvis_msda_fused_forward_h")
        return out
    if head_stride:
        if offsets.shape[1] < (M - 1) * head_stride + L * P * 2 or logits.shape[1] < (M - 1) * head_stride + L * P:
            raise RuntimeError("msda_fused_forward: rows shorter than M slots")
    elif offsets.shape[1] < M * L * P * 2 or logits.shape[1] < M * L * P:
        raise RuntimeError("msda_fused_forward: inconsistent shapes")
    if reference_points.shape[2:] != (L, 2):
        raise RuntimeError("msda_fused_forward: inconsistent shapes")
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    hs = None
    if shapes_host is not None:
        hs = (ctypes.c_int64 * (2 * L))(*[int(v) for hw in shapes_host for v in hw])
    po = pl = None
    pstride = 0
    if pos_offsets is not None or pos_logits is not None:
        wo = (M - 1) * head_stride + L * P * 2 if head_stride else M * L * P * 2
        wl = (M - 1) * head_stride + L * P if head_stride else M * L * P
        for name, t, width in (("pos_offsets", pos_offsets, wo), ("pos_logits", pos_logits, wl)):
            if t is None or not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1 \
                    or t.shape[0] != Lq or t.shape[1] < width:
                raise RuntimeError(f"{name} must be a float32 GPU (Lq, >= {width}) row view with unit inner stride")
        if pos_offsets.stride(0) != pos_logits.stride(0):
            raise RuntimeError("pos_offsets / pos_logits must share their row stride (slices of one projection)")
        po, pl, pstride = ctypes.c_void_p(pos_offsets.data_ptr()), ctypes.c_void_p(pos_logits.data_ptr()), pos_offsets.stride(0)
    with torch.cuda.device(value.device):
        rc = native.lib().dvis_msda_fused_forward_slots(
            native.dev_ptr(value, "value"), native.dev_ptr(spatial_shapes, "spatial_shapes"),
            native.dev_ptr(level_start_index, "level_start_index"), native.dev_ptr(reference_points, "ref"), nref,
            ctypes.c_void_p(offsets.data_ptr()), offsets.stride(0), ctypes.c_void_p(logits.data_ptr()),
            logits.stride(0), int(head_stride), int(head_stride), 1 if value_head_major else 0, po, pl, pstride,
            N, S, M, D, L, Lq, P,
            native.dev_ptr(out, "out"), hs, native.stream_ptr(value.device))
    native.check(rc, "dvis_msda_fused_forward")
    return out


def _on_cpu(*tensors):
    """Every tensor lives on the CPU: the call takes the torch formulation of cpu_ops.py (BASELINE config #1, "PyTorch CPU
    MSDeformAttn fallback (plumbing, no GPU)").  Decided by DEVICE alone: a GPU tensor never ends up there — the HIP kernel runs
    or the call raises."""
    return all(t.device.type == "cpu" for t in tensors)


def _f32_gpu(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    return native.dev_ptr(t, name)


def mask_logits(mask_embed, mask_features):
    """einsum("bqc,bchw->bqhw") of forward_prediction_heads
    (dvis_Plus/video_mask2former_transformer_decoder.py:363), exact-fp32 MFMA.  (B,Q,C) x (B,C,H,W) -> (B,Q,H,W)."""
    B, Q, C = mask_embed.shape
    Bf, Cf, H, W = mask_features.shape
    if (Bf, Cf) != (B, C):
        raise RuntimeError("mask_logits: mask_embed (B,Q,C) and mask_features (B,C,H,W) disagree")
    if _on_cpu(mask_embed, mask_features):
        return cpu_ops.mask_logits(mask_embed, mask_features)
    pe, pf = _f32_gpu(mask_embed, "mask_embed"), _f32_gpu(mask_features, "mask_features")
    out = torch.empty((B, Q, H, W), dtype=torch.float32, device=mask_embed.device)
    with torch.cuda.device(mask_embed.device):
        rc = native.lib().dvis_mask_logits(pe, pf, B, Q, C, H * W, native.dev_ptr(out, "out"),
                                           native.stream_ptr(mask_embed.device))
    native.check(rc, "dvis_mask_logits")
    return out


def attn_mask(mask_embed, mask_features, target_size):
    """Attention mask of the next decoder layer in ONE launch: contraction + bilinear down-sizing + threshold
    (ibid. :363-371).  Returns (mask uint8 (B,Q,h*w) with 1 = blocked — a single copy, not repeated per head —,
    allowed_count int32 (B,Q)); a row with allowed_count == 0 must be treated as un-masked (:297), which
    ``attention`` does on the device."""
    B, Q, C = mask_embed.shape
    Bf, Cf, H, W = mask_features.shape
    h, w = int(target_size[0]), int(target_size[1])
    if (Bf, Cf) != (B, C):
        raise RuntimeError("attn_mask: mask_embed (B,Q,C) and mask_features (B,C,H,W) disagree")
    if _on_cpu(mask_embed, mask_features):
        return cpu_ops.attn_mask(mask_embed, mask_features, (h, w))
    pe, pf = _f32_gpu(mask_embed, "mask_embed"), _f32_gpu(mask_features, "mask_features")
    mask = torch.empty((B, Q, h * w), dtype=torch.uint8, device=mask_embed.device)
    allowed = torch.empty((B, Q), dtype=torch.int32, device=mask_embed.device)
    with torch.cuda.device(mask_embed.device):
        rc = native.lib().dvis_attn_mask(pe, pf, B, Q, C, H, W, h, w, native.dev_ptr(mask, "mask"),
                                         native.dev_ptr(allowed, "allowed"), native.stream_ptr(mask_embed.device))
    native.check(rc, "dvis_attn_mask")
    return mask, allowed


ATTN_MASK_PYRAMID = os.environ.get("DVIS_ATTN_MASK_PYRAMID", "1") != "0"


def center_pool3(mask_features):
    """The decoder's attention-mask pyramid, once per call of the decoder: (p8, p4, p2) — low resolution first, the decoder's
    level order — with p_s[n, c, i, j] = 0.25 * ((f_a + f_b) + (f_c + f_d)) over the four centre pixels of block (i, j) of s x s
    pixels: what F.interpolate(bilinear, align_corners=False) to 1/s samples, on the FEATURES (the contraction is linear).  None
    when the map is not served (H, W not multiples of 8, CPU tensors, autograd): the caller uses ``attn_mask``."""
    if not (ATTN_MASK_PYRAMID and mask_features.is_cuda and mask_features.dtype == torch.float32 and mask_features.dim() == 4
            and mask_features.is_contiguous() and not torch.is_grad_enabled()):
        return None
    N, C, H, W = mask_features.shape
    if H % 8 or W % 8 or mask_features.data_ptr() % 16:
        return None
    p2 = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=mask_features.device)
    p4 = torch.empty((N, C, H // 4, W // 4), dtype=torch.float32, device=mask_features.device)
    p8 = torch.empty((N, C, H // 8, W // 8), dtype=torch.float32, device=mask_features.device)
    if N:
        with torch.cuda.device(mask_features.device):
            native.check(native.lib().dvis_center_pool3(native.dev_ptr(mask_features, "mask_features"), N * C, H, W,
                                                        native.dev_ptr(p2, "p2"), native.dev_ptr(p4, "p4"), native.dev_ptr(p8, "p8"),
                                                        native.stream_ptr(mask_features.device)), "dvis_center_pool3")
    return p8, p4, p2


def attn_mask_pooled(mask_embed, pooled):
    """``attn_mask`` on a level's pooled map (center_pool3): (mask uint8 (B, Q, h*w), allowed_count int32 (B, Q))."""
    B, Q, C = mask_embed.shape
    Bf, Cf, h, w = pooled.shape
    if (Bf, Cf) != (B, C):
        raise RuntimeError("attn_mask_pooled: mask_embed (B,Q,C) and pooled (B,C,h,w) disagree")
    pe, pf = _f32_gpu(mask_embed, "mask_embed"), _f32_gpu(pooled, "pooled")
    mask = torch.empty((B, Q, h * w), dtype=torch.uint8, device=mask_embed.device)
    allowed = torch.empty((B, Q), dtype=torch.int32, device=mask_embed.device)
    with torch.cuda.device(mask_embed.device):
        native.check(native.lib().dvis_attn_mask_pooled(pe, pf, B, Q, C, h, w, native.dev_ptr(mask, "mask"),
                                                        native.dev_ptr(allowed, "allowed"), native.stream_ptr(mask_embed.device)),
                     "dvis_attn_mask_pooled")
    return mask, allowed


def _strides3(t, B, C, d):
    """(L, B, C) tensor -> {batch, head, row} strides in floats for the kernel's (B, heads, L, d) view."""
    return (ctypes.c_int64 * 3)(t.stride(1) if B > 1 else C, d, t.stride(0))


def attention(q, k, v, nheads, mask=None, allowed_count=None, out=None, short=False):
    """Multi-head softmax(q k^T / sqrt(d)) v on projected, sequence-first tensors (what
    nn.MultiheadAttention computes between its in- and out-projection).

    q (Lq, B, C), k / v (Lk, B, C) float32 GPU tensors whose last dim is contiguous — arbitrary row / batch
    strides, so slices of a fused in-projection work without copies; heads are the C = nheads * d split,
    d in {32, 64}.  mask: uint8 / bool (B, Lq, Lk), 1 = blocked, shared by all heads of a batch entry;
    allowed_count int32 (B, Lq) from ``attn_mask`` (rows with 0 ignore the mask).
    Returns (Lq, B, C) ready for the out-projection.
    short=True pins the short-key latency kernel (Lk <= 128) whatever the batch size: same bits per (batch, head) for a clip
    alone or stacked with others (the tracker's recurrence).
    """
    Lq, B, C = q.shape
    Lk = k.shape[0]
    d = C // nheads
    if _on_cpu(q, k, v):
        return cpu_ops.attention(q, k, v, nheads, mask, allowed_count, out)
    for name, t in (("q", q), ("k", k), ("v", v)):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1:
            raise RuntimeError(f"attention: {name} must be a float32 GPU (L, B, C) tensor with a contiguous last dim")
    if k.shape != v.shape or k.shape[1:] != q.shape[1:]:
        raise RuntimeError("attention: inconsistent q / k / v shapes")
    if out is None:
        out = torch.empty((Lq, B, C), dtype=torch.float32, device=q.device)
    mptr = aptr = None
    if mask is not None:
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        if mask.shape != (B, Lq, Lk) or mask.dtype != torch.uint8:
            raise RuntimeError("attention: mask must be uint8/bool (B, Lq, Lk)")
        mptr = native.dev_ptr(mask, "mask")
        if allowed_count is not None:
            if allowed_count.shape != (B, Lq) or allowed_count.dtype != torch.int32:
                raise RuntimeError("attention: allowed_count must be int32 (B, Lq)")
            aptr = native.dev_ptr(allowed_count, "allowed_count")
    lib = native.lib()
    kern = 1 if short else 0
    if not short and mask is None and d == 64 and Lq >= 1024 and Lk >= 1024 and x3_on():
        # long self-attention at head dim 64 (the DINOv2 / ViT-Adapter blocks: 3681 tokens x 16 heads): split-f16 matrix-core kernel
        kern = 2
        X3_GUARD.word(q.device)          # (its range guard reports under the tag of the last packed weight: the block's qkv)
    nbytes = lib.dvis_attention_ws_bytes_k(B * nheads, Lq, Lk, d, kern)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=q.device) if nbytes else None
    with torch.cuda.device(q.device):
        rc = lib.dvis_attention_forward_k(
            ctypes.c_void_p(q.data_ptr()), _strides3(q, B, C, d), ctypes.c_void_p(k.data_ptr()), _strides3(k, B, C, d),
            ctypes.c_void_p(v.data_ptr()), _strides3(v, B, C, d), ctypes.c_void_p(out.data_ptr()),
            _strides3(out, B, C, d), mptr, aptr, B, nheads, Lq, Lk, d, 1.0 / (d ** 0.5),
            ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, native.stream_ptr(q.device), kern)
    native.check(rc, "dvis_attention_forward")
    return out


def vps_argmax(mask_logits_kthw, scores, first_resize_size, img_size, out_hw):
    """Fused panoptic arg-max (see dvis_vps_argmax).  mask_logits_kthw: float32 GPU (K, T, h, w) view whose last two
    dims are contiguous; scores float32 (K,).  Returns ids int32 (T,H,W), conf bool (T,H,W), areas int32 (3, K)."""
    K, T, h, w = mask_logits_kthw.shape
    m = mask_logits_kthw
    if not m.is_cuda or m.dtype != torch.float32 or m.stride(3) != 1 or m.stride(2) != w:
        raise RuntimeError("vps_argmax: logits must be a float32 GPU (K, T, h, w) view with contiguous (h, w) maps")
    scores = scores.to(torch.float32).contiguous()
    oh, ow = int(out_hw[0]), int(out_hw[1])
    ids = torch.empty((T, oh, ow), dtype=torch.int32, device=m.device)
    conf = torch.empty((T, oh, ow), dtype=torch.uint8, device=m.device)
    areas = torch.empty((3, K), dtype=torch.int32, device=m.device)
    with torch.cuda.device(m.device):
        rc = native.lib().dvis_vps_argmax(
            ctypes.c_void_p(m.data_ptr()), m.stride(0), m.stride(1), native.dev_ptr(scores, "scores"), K, T, h, w,
            int(first_resize_size[0]), int(first_resize_size[1]), int(img_size[0]), int(img_size[1]), oh, ow,
            native.dev_ptr(ids, "ids"), native.dev_ptr(conf, "conf"), native.dev_ptr(areas, "areas"),
            native.stream_ptr(m.device))
    native.check(rc, "dvis_vps_argmax")
    return ids, conf.bool(), areas


def resize2_gt0(mask_logits_kthw, first_resize_size, img_size, out_hw):
    """``F.interpolate(F.interpolate(m, first)[:, :, :img_h, :img_w], out_hw) > 0`` in one pass (see dvis_resize2_gt0):
    the instance masks of inference_video_vis.  m: float32 GPU (K, T, h, w) view with contiguous (h, w) maps.
    Returns bool (K, T, H, W)."""
    K, T, h, w = mask_logits_kthw.shape
    m = mask_logits_kthw
    if not m.is_cuda or m.dtype != torch.float32 or m.stride(3) != 1 or m.stride(2) != w:
        raise RuntimeError("resize2_gt0: logits must be a float32 GPU (K, T, h, w) view with contiguous (h, w) maps")
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((K, T, oh, ow), dtype=torch.uint8, device=m.device)
    with torch.cuda.device(m.device):
        rc = native.lib().dvis_resize2_gt0(
            ctypes.c_void_p(m.data_ptr()), m.stride(0), m.stride(1), K, T, h, w, int(first_resize_size[0]),
            int(first_resize_size[1]), int(img_size[0]), int(img_size[1]), oh, ow, native.dev_ptr(out, "out"),
            native.stream_ptr(m.device))
    native.check(rc, "dvis_resize2_gt0")
    return out.view(torch.bool)


def resize2(mask_logits_kthw, first_resize_size, img_size, out_hw, sigmoid=False):
    """``F.interpolate(f(F.interpolate(m, first)[:, :, :img_h, :img_w]), out_hw)``, f = sigmoid or identity, in one pass and
    in torch's CPU operation order (see dvis_resize2).  Returns float32 (K, T, H, W)."""
    K, T, h, w = mask_logits_kthw.shape
    m = mask_logits_kthw
    if not m.is_cuda or m.dtype != torch.float32 or m.stride(3) != 1 or m.stride(2) != w:
        raise RuntimeError("resize2: logits must be a float32 GPU (K, T, h, w) view with contiguous (h, w) maps")
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((K, T, oh, ow), dtype=torch.float32, device=m.device)
    with torch.cuda.device(m.device):
        rc = native.lib().dvis_resize2(
            ctypes.c_void_p(m.data_ptr()), m.stride(0), m.stride(1), K, T, h, w, int(first_resize_size[0]),
            int(first_resize_size[1]), int(img_size[0]), int(img_size[1]), oh, ow, 1 if sigmoid else 0,
            native.dev_ptr(out, "out"), native.stream_ptr(m.device))
    native.check(rc, "dvis_resize2")
    return out


def vss_argmax(mask_logits_qthw, mask_cls, first_resize_size, img_size, out_hw):
    """Fused semantic arg-max (see dvis_vss_argmax).  mask_logits_qthw: float32 GPU (Q, T, h, w) view whose last two dims
    are contiguous; mask_cls float32 (Q, C), C <= 128.  Returns int64 (T, H, W) class indices."""
    Q, T, h, w = mask_logits_qthw.shape
    m = mask_logits_qthw
    if not m.is_cuda or m.dtype != torch.float32 or m.stride(3) != 1 or m.stride(2) != w:
        raise RuntimeError("vss_argmax: logits must be a float32 GPU (Q, T, h, w) view with contiguous (h, w) maps")
    C = mask_cls.shape[1]
    Cp = (C + 31) // 32 * 32
    cls = torch.nn.functional.pad(mask_cls.to(torch.float32), (0, Cp - C)).contiguous()     # zero columns: never the max
    oh, ow = int(out_hw[0]), int(out_hw[1])
    out = torch.empty((T, oh, ow), dtype=torch.int64, device=m.device)
    with torch.cuda.device(m.device):
        rc = native.lib().dvis_vss_argmax(
            ctypes.c_void_p(m.data_ptr()), m.stride(0), m.stride(1), native.dev_ptr(cls, "mask_cls"), Cp, Q, C, T,
            h, w, int(first_resize_size[0]), int(first_resize_size[1]), int(img_size[0]), int(img_size[1]), oh, ow,
            native.dev_ptr(out, "out"), native.stream_ptr(m.device))
    native.check(rc, "dvis_vss_argmax")
    return out


def add_layer_norm(x, res, norm, pos=None):
    """``norm(x + res)`` for an ``nn.LayerNorm`` `norm` in ONE pass.  x contiguous float32 GPU (..., C); res: None or a
    tensor broadcast-free of x's shape whose rows (last dim) are contiguous.  CPU tensors / other dtypes use torch ops
    (library code, not one of the named hot ops).
    pos: optional (1, S, C) / (S, C) position embedding for x of shape (N, S, C): returns ``(out, out + pos)`` — the
    second tensor (the next encoder layer's query) is written by the same kernel."""
    C = x.shape[-1]

    def with_pos(out):
        return out if pos is None else (out, out + pos.reshape(1, -1, C))
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and C % 4 == 0 and C <= 1024
            and norm.weight is not None and not torch.is_grad_enabled()):
        _torch_path("add_layer_norm", x, "needs contiguous fp32 rows with C % 4 == 0, C <= 1024 and an affine norm")
        return with_pos(norm(x if res is None else x + res))
    rp, rs = None, 0
    if res is not None:
        if res.shape != x.shape or res.dtype != torch.float32 or not res.is_cuda:
            _torch_path("add_layer_norm", x, "residual must be fp32 of x's shape")
            return with_pos(norm(x + res))
        if not res.is_contiguous():
            res = res.contiguous()
        rp, rs = ctypes.c_void_p(res.data_ptr()), C
    out = torch.empty_like(x)
    rows = x.numel() // C
    with torch.cuda.device(x.device):
        if pos is None:
            rc = native.lib().dvis_add_layernorm(ctypes.c_void_p(x.data_ptr()), rp, rs, native.dev_ptr(norm.weight, "gamma"),
                                                 native.dev_ptr(norm.bias, "beta"), ctypes.c_void_p(out.data_ptr()),
                                                 rows, C, float(norm.eps), native.stream_ptr(x.device))
        else:
            pos_rows = pos.numel() // C
            if x.dim() != 3 or x.shape[1] != pos_rows or pos.dtype != torch.float32 or not pos.is_cuda:
                raise RuntimeError("add_layer_norm: pos must be a float32 GPU (S, C) embedding for x of shape (N, S, C)")
            pos = pos.contiguous()
            out_pos = torch.empty_like(x)
            rc = native.lib().dvis_add_layernorm_pos(
                ctypes.c_void_p(x.data_ptr()), rp, rs, native.dev_ptr(norm.weight, "gamma"),
                native.dev_ptr(norm.bias, "beta"), ctypes.c_void_p(out.data_ptr()), native.dev_ptr(pos, "pos"), pos_rows,
                ctypes.c_void_p(out_pos.data_ptr()), rows, C, float(norm.eps), native.stream_ptr(x.device))
    native.check(rc, "dvis_add_layernorm")
    return out if pos is None else (out, out_pos)


def bias_relu_maxpool(x, bias=None):
    """``F.max_pool2d(relu(x + bias[c]), 3, stride=2, padding=1)`` in one pass (the ResNet stem after its convolution):
    relu(max(x) + b), bit-identical because fp32 add and ReLU are monotonic.  Non-GPU / odd shapes use the torch ops."""
    N, C, H, W = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and H % 2 == 0 and W % 8 == 0
            and not torch.is_grad_enabled()):
        _torch_path("bias_relu_maxpool", x, "needs contiguous fp32 NCHW with H % 2 == 0 and W % 8 == 0")
        import torch.nn.functional as F
        if bias is not None:
            x = x + bias.view(1, -1, 1, 1)
        return F.max_pool2d(torch.relu(x), kernel_size=3, stride=2, padding=1)
    out = torch.empty((N, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = native.lib().dvis_bias_relu_maxpool(native.dev_ptr(x, "x"),
                                                 None if bias is None else native.dev_ptr(bias, "bias"),
                                                 native.dev_ptr(out, "out"), N * C, C, H, W, native.stream_ptr(x.device))
    native.check(rc, "dvis_bias_relu_maxpool")
    return out


def _fused_map_ok(x):
    """fp32 NCHW GPU map outside autograd: the fused glue kernels apply; anything else takes the torch ops."""
    return x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not torch.is_grad_enabled()


def group_norm_affine(x, norm):
    """Statistics of ``nn.GroupNorm`` `norm` on x (N,C,H,W) in one read, as the per-plane affine (scale, shift), each
    (N*C,), with  norm(x)[n, c] == x[n, c] * scale[n*C + c] + shift[n*C + c].  None when the fused path does not apply
    (caller falls back to ``norm(x)``)."""
    N, C, H, W = x.shape
    G = norm.num_groups
    if not (_fused_map_ok(x) and C // G <= 1024):
        _torch_path("group_norm_affine", x, "needs contiguous fp32 NCHW, at most 1024 channels per group")
        return None
    scale = torch.empty(N * C, dtype=torch.float32, device=x.device)
    shift = torch.empty(N * C, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = native.lib().dvis_group_norm_affine(
            native.dev_ptr(x, "x"), None if norm.weight is None else native.dev_ptr(norm.weight.detach(), "gamma"),
            None if norm.bias is None else native.dev_ptr(norm.bias.detach(), "beta"), native.dev_ptr(scale, "scale"),
            native.dev_ptr(shift, "shift"), N, C, G, H * W, float(norm.eps), native.stream_ptr(x.device))
    native.check(rc, "dvis_group_norm_affine")
    return scale, shift


def scale_shift_act_(x, scale, shift, relu=False):
    """In place: x[n, c] = relu?(x[n, c] * scale[n*C + c] + shift[n*C + c]) on an fp32 NCHW GPU map."""
    N, C, H, W = x.shape
    with torch.cuda.device(x.device):
        rc = native.lib().dvis_scale_shift_act(native.dev_ptr(x, "x"), native.dev_ptr(scale, "scale"),
                                               native.dev_ptr(shift, "shift"), N * C, H * W, 1 if relu else 0,
                                               native.stream_ptr(x.device))
    native.check(rc, "dvis_scale_shift_act")
    return x


def upsample_add(lateral, top, lat_affine=None):
    """``lateral + F.interpolate(top, size=lateral.shape[-2:], mode="bilinear", align_corners=False)`` in one pass.
    lat_affine = (scale, shift) from ``group_norm_affine``: the lateral operand is read as lateral * scale + shift (its
    GroupNorm applied on the fly)."""
    N, C, H, W = lateral.shape
    if not (_fused_map_ok(lateral) and top.dtype == torch.float32):
        _torch_path("upsample_add", lateral, "needs contiguous fp32 NCHW")
        import torch.nn.functional as F
        if lat_affine is not None:
            lateral = lateral * lat_affine[0].view(N, C, 1, 1) + lat_affine[1].view(N, C, 1, 1)
        return lateral + F.interpolate(top, size=(H, W), mode="bilinear", align_corners=False)
    top = top.contiguous()
    out = torch.empty_like(lateral)
    with torch.cuda.device(lateral.device):
        if lat_affine is None:
            rc = native.lib().dvis_upsample_add(native.dev_ptr(lateral, "lateral"), native.dev_ptr(top, "top"),
                                                native.dev_ptr(out, "out"), N * C, H, W, top.shape[-2], top.shape[-1],
                                                native.stream_ptr(lateral.device))
        else:
            rc = native.lib().dvis_upsample_add_affine(
                native.dev_ptr(lateral, "lateral"), native.dev_ptr(lat_affine[0], "scale"),
                native.dev_ptr(lat_affine[1], "shift"), native.dev_ptr(top, "top"), native.dev_ptr(out, "out"), N * C, H,
                W, top.shape[-2], top.shape[-1], native.stream_ptr(lateral.device))
    native.check(rc, "dvis_upsample_add")
    return out


def dwconv3x3_tokens(x, level_shapes, weight, bias, gelu=False):
    """Depthwise 3x3 convolution (+ bias, + exact GELU) of every pyramid level of a token tensor x (B, N, C) whose levels
    `level_shapes` = [(h, w), ...] are stored one after the other along N (dvis_dwconv3x3_tokens, one launch per level).
    weight (C, 1, 3, 3).  fp32 contiguous GPU tensors."""
    B, N, C = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and weight.dtype == torch.float32 and C % 4 == 0
            and sum(h * w for h, w in level_shapes) == N and weight.shape == (C, 1, 3, 3)):
        raise RuntimeError(f"dwconv3x3_tokens: needs a contiguous float32 GPU (B, N, C) tensor, C % 4 == 0, levels summing to N "
                           f"(x {tuple(x.shape)}, levels {list(level_shapes)})")
    out = torch.empty_like(x)
    wt = weight.detach().reshape(C, 9).contiguous()
    bptr = None if bias is None else native.dev_ptr(bias.detach().contiguous(), "bias")
    off = 0
    with torch.cuda.device(x.device):
        for h, w in level_shapes:
            rc = native.lib().dvis_dwconv3x3_tokens(ctypes.c_void_p(x.data_ptr() + 4 * off * C), ctypes.c_void_p(out.data_ptr() + 4 * off * C),
                                                    N * C, B, h, w, C, native.dev_ptr(wt, "weight"), bptr, int(gelu),
                                                    native.stream_ptr(x.device))
            native.check(rc, "dvis_dwconv3x3_tokens")
            off += h * w
    return out


def adapter_res2(g, c1, x1_tokens, scale, shift, h8, w8):
    """The ViT-Adapter's stride-4 output in one pass (dvis_adapter_res2, include/dvis_hip.h):
    out[b, co, 2y+dy, 2x+dx] = g[(b, y, x), (dy, dx, co)] + scale[co] * (c1 + up4(x1))[b, co, 2y+dy, 2x+dx] + shift[co].
    g (B*h8*w8, 4C) token-major result of the transposed convolution run as a GEMM, c1 (B, C, 2 h8, 2 w8) contiguous NCHW,
    x1_tokens (B, h8/2 * w8/2, C) or None.  fp32 GPU tensors only (the caller keeps the torch composition for the rest)."""
    B, C = c1.shape[0], c1.shape[1]
    for t, name in ((g, "g"), (c1, "c1"), (scale, "scale"), (shift, "shift")) + (((x1_tokens, "x1"),) if x1_tokens is not None else ()):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError(f"adapter_res2: {name} must be a contiguous float32 GPU tensor")
    if g.shape != (B * h8 * w8, 4 * C) or c1.shape != (B, C, 2 * h8, 2 * w8) or \
            (x1_tokens is not None and x1_tokens.shape != (B, (h8 // 2) * (w8 // 2), C)):
        raise RuntimeError(f"adapter_res2: shapes g {tuple(g.shape)}, c1 {tuple(c1.shape)}, x1 "
                           f"{None if x1_tokens is None else tuple(x1_tokens.shape)} do not match a {h8} x {w8} stride-8 grid")
    out = torch.empty_like(c1)
    with torch.cuda.device(c1.device):
        rc = native.lib().dvis_adapter_res2(native.dev_ptr(g, "g"), native.dev_ptr(c1, "c1"),
                                            None if x1_tokens is None else native.dev_ptr(x1_tokens, "x1"),
                                            native.dev_ptr(scale, "scale"), native.dev_ptr(shift, "shift"), native.dev_ptr(out, "out"),
                                            B, C, h8, w8, native.stream_ptr(c1.device))
    native.check(rc, "dvis_adapter_res2")
    return out


# Library GEMMs or the own deterministic kernel (dvis_gemm_nt)?  The tracker / refiner ALWAYS take the own kernel
# (own=True at their call sites: their stream must never carry a library stream-K kernel, csrc/gemm.hip); everything
# else follows this switch.  DVIS_DETERMINISTIC=1: every GEMM of the pipeline that goes through this module is the own
# kernel -> two runs of a clip give bit-identical tensors (the library's split / stream-K kernels do not promise that).
OWN_GEMM_DEFAULT = os.environ.get("DVIS_DETERMINISTIC", "1") != "0"


class gemm_sizes_as:
    """Inside this context dvis_gemm_nt picks its tile configuration as if the problem had `rows` rows (and `batch` batch
    entries), whatever the actual operand has.  A row's result depends on (N, K, configuration) only, so the tracker gets
    the SAME bits for a clip whether its recurrence runs alone (Q rows per GEMM) or stacked with another clip's (2Q rows)."""
    _tls = threading.local()      # per thread: stream()'s phase-B worker thread must not re-size the main thread's GEMMs

    def __init__(self, rows=None, batch=None):
        self.rows, self.batch = rows, batch

    @staticmethod
    def current():
        return getattr(gemm_sizes_as._tls, "pin", (None, None))

    def __enter__(self):
        self.prev = gemm_sizes_as.current()
        gemm_sizes_as._tls.pin = (self.rows, self.batch)

    def __exit__(self, *exc):
        gemm_sizes_as._tls.pin = self.prev


def _gemm_config(M, N, K, batch, config):
    rows, b = gemm_sizes_as.current()
    if config >= 0 or (rows is None and b is None):
        return config
    return native.lib().dvis_gemm_pick_config(rows or M, N, K, b or batch)


def _rows2d(x, K):
    """(..., K) tensor -> (2-D view (M, K) with unit inner stride, row stride in floats); copies only when the leading
    dims do not collapse into one stride."""
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= K:
        return x, x.stride(0)
    if not x.is_contiguous():
        x = x.contiguous()
    return x.view(-1, K), K


def gemm_nt(a, w, bias=None, relu=False, res=None, config=-1, head_major=0, out=None):
    """``relu?(a @ w.T + bias + res)`` on the deterministic exact-fp32 MFMA kernel (dvis_gemm_nt).  a (..., K) float32 GPU
    tensor (rows with unit inner stride; a row-sliced 2-D view keeps its row stride), w (N, K) in F.linear's layout (row
    stride >= K allowed), bias (N) or None, res (..., N) or None.  Returns (..., N).  Raises when the kernel cannot serve
    the operands (K or a row stride not a multiple of 4, unaligned base) — there is no silent library fallback here.
    head_major = d > 0: the output is written as (N / d, M, d) — N / d matrices of M rows, d columns each (the head-major
    value layout of MSDeformAttn) — and returned with that shape (M = all leading dims of `a` flattened)."""
    K = a.shape[-1]
    N = w.shape[0]
    if not (a.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] == K
            and w.stride(1) == 1):
        raise RuntimeError("gemm_nt: needs float32 GPU operands a (..., K), w (N, K) with unit inner strides")
    a2, lda = _rows2d(a, K)
    M = a2.shape[0]
    if head_major:
        if N % head_major or head_major % 4 or res is not None:
            raise RuntimeError("gemm_nt: head_major needs N % d == 0, d % 4 == 0 and no residual")
        out = torch.empty((N // head_major, M, head_major), dtype=torch.float32, device=a.device)
    elif out is None:
        out = torch.empty((*a.shape[:-1], N), dtype=torch.float32, device=a.device)
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() == M * N):
        raise RuntimeError("gemm_nt: out must be a contiguous float32 GPU tensor of M * N elements")
    rp, ldres = None, 0
    if res is not None:
        if res.numel() != out.numel() or res.shape[-1] != N or res.dtype != torch.float32 or not res.is_cuda:
            raise RuntimeError("gemm_nt: res must be a float32 GPU tensor of the output's shape")
        r2, ldres = _rows2d(res, N)
        rp = ctypes.c_void_p(r2.data_ptr())
    bp = None
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != N or not bias.is_contiguous():
            raise RuntimeError("gemm_nt: bias must be a contiguous float32 (N,) tensor")
        bp = ctypes.c_void_p(bias.data_ptr())
    with torch.cuda.device(a.device):
        rc = native.lib().dvis_gemm_nt_hm(ctypes.c_void_p(a2.data_ptr()), lda, 0, ctypes.c_void_p(w.data_ptr()), w.stride(0), 0,
                                          bp, rp, ldres, 0, ctypes.c_void_p(out.data_ptr()), head_major or N, 0, M, N, K, 1,
                                          1 if relu else 0, _gemm_config(M, N, K, 1, config), int(head_major),
                                          M * head_major, native.stream_ptr(a.device))
    native.check(rc, "dvis_gemm_nt")
    return out


def gemm_nt_stacked(a, w, bias=None, config=-1):
    """`L` projections of one activation matrix as ONE launch: a (M, L * K) holds L operands side by side (operand l =
    columns [l K, (l + 1) K)), w (L, N, K) / bias (L, N) the L layers' ``nn.Linear`` parameters -> (L, M, N) with
    out[l] = a[:, l K:(l + 1) K] @ w[l].T + bias[l] (dvis_gemm_nt_bb: a batched GEMM with a bias per batch entry)."""
    L, N, K = w.shape
    if not (a.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1
            and a.shape[1] == L * K and w.is_contiguous()):
        raise RuntimeError("gemm_nt_stacked: needs float32 GPU operands a (M, L * K) with unit inner stride, contiguous w (L, N, K)")
    if bias is not None and not (bias.dtype == torch.float32 and bias.shape == (L, N) and bias.is_contiguous()):
        raise RuntimeError("gemm_nt_stacked: bias must be a contiguous float32 (L, N) tensor")
    M = a.shape[0]
    out = torch.empty((L, M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = native.lib().dvis_gemm_nt_bb(ctypes.c_void_p(a.data_ptr()), a.stride(0), K, ctypes.c_void_p(w.data_ptr()), K, N * K,
                                          None if bias is None else ctypes.c_void_p(bias.data_ptr()), N, None, 0, 0,
                                          ctypes.c_void_p(out.data_ptr()), N, M * N, M, N, K, L, 0,
                                          _gemm_config(M, N, K, L, config), native.stream_ptr(a.device))
    native.check(rc, "dvis_gemm_nt_bb")
    return out


def gemm_ln_ok(a, w, norms=True):
    """Can dvis_gemm_ln serve ``a (..., K) @ w (N, K).T``?  (fp32 GPU inference tensors, K % 16 == 0, K <= 512 — 2048 for a
    call without norms —, N % 4 == 0)"""
    return (a.is_cuda and a.dtype == torch.float32 and w.dtype == torch.float32 and not torch.is_grad_enabled()
            and w.dim() == 2 and w.stride(1) == 1 and w.stride(0) % 4 == 0
            and bool(native.lib().dvis_gemm_ln_supported(max(1, a.numel() // a.shape[-1]), w.shape[0], a.shape[-1],
                                                         1 if norms else 0)))


def gemm_ln(a, w, bias=None, norm1=None, add=None, norm2=None, relu=False, res=None, a_out=None, out=None, config=-1):
    """``x = norm2(norm1(a) + add)``; returns ``(relu?(x @ w.T + bias + res), x)`` — the LayerNorm seam of a post-norm
    transformer block folded into the prologue of the projection that consumes it (csrc/gemm_ln.hip).  Without norms it
    is the latency form of ``gemm_nt`` for launch-bound chains (every operand requested up front, K <= 2048; x is None).  a (..., K) float32
    GPU rows (the producer's RAW residual sum), norm1 / norm2 ``nn.LayerNorm`` modules or None, add (..., K) or None,
    w (N, K), res (..., N) or None.  a_out / out: optional preallocated destinations ((..., K) / (..., N) row views).
    Raises when the kernel cannot serve the operands — no fallback."""
    K = a.shape[-1]
    N = w.shape[0]
    if (add is None) != (norm2 is None):
        raise RuntimeError("gemm_ln: the forms are norm1, add + norm2, norm1 + add + norm2 or none")
    if not gemm_ln_ok(a, w, norm1 is not None or norm2 is not None) or w.shape[1] != K:
        raise RuntimeError(f"gemm_ln: operands not served (a {tuple(a.shape)} {a.dtype}, w {tuple(w.shape)}; needs fp32 GPU, "
                           "K % 16 == 0, K <= 512, N % 4 == 0, no autograd)")
    a2, lda = _rows2d(a, K)
    M = a2.shape[0]
    lead = a.shape[:-1]

    def rows(t, width, name):
        if t is None:
            return None, 0
        if t.dtype != torch.float32 or not t.is_cuda or t.shape[-1] != width or t.numel() != M * width:
            raise RuntimeError(f"gemm_ln: {name} must be a float32 GPU tensor of {M} rows x {width}")
        t2, ld = _rows2d(t, width)
        return t2, ld
    add2, ldadd = rows(add, K, "add")
    res2, ldres = rows(res, N, "res")
    if a_out is None and (norm1 is not None or norm2 is not None):
        a_out = torch.empty((*lead, K), dtype=torch.float32, device=a.device)
    if out is None:
        out = torch.empty((*lead, N), dtype=torch.float32, device=a.device)
    ao2, ldao = rows(a_out, K, "a_out")
    o2, ldc = rows(out, N, "out")
    if (ao2 is not None and ao2.data_ptr() != a_out.data_ptr()) or o2.data_ptr() != out.data_ptr():
        raise RuntimeError("gemm_ln: a_out / out must be row views (unit inner stride, one row stride)")

    def affine(n):
        if n is None:
            return None, None, 0.0
        if n.weight is None or n.weight.numel() != K or n.weight.dtype != torch.float32:
            raise RuntimeError("gemm_ln: needs an affine float32 LayerNorm over the K columns")
        return ctypes.c_void_p(n.weight.data_ptr()), ctypes.c_void_p(n.bias.data_ptr()), float(n.eps)
    g1, b1, e1 = affine(norm1)
    g2, b2, e2 = affine(norm2)
    rows_pin, _ = gemm_sizes_as.current()
    cfg = config if config >= 0 or rows_pin is None else native.lib().dvis_gemm_ln_pick_config(rows_pin, N, K)
    ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(a.device):
        rc = native.lib().dvis_gemm_ln(ptr(a2), lda, ptr(add2), ldadd, g1, b1, e1, g2, b2, e2, ptr(ao2), ldao,
                                       ctypes.c_void_p(w.data_ptr()), w.stride(0), None if bias is None else ptr(bias.detach()),
                                       ptr(res2), ldres, ptr(o2), ldc, M, N, K, 1 if relu else 0, cfg,
                                       native.stream_ptr(a.device))
    native.check(rc, "dvis_gemm_ln")
    return out, a_out


def bmm_nt(a, b, config=-1):
    """Batched ``a @ b.transpose(1, 2)`` on dvis_gemm_nt: a (B, M, K), b (B, N, K) contiguous float32 GPU -> (B, M, N)."""
    B, M, K = a.shape
    N = b.shape[1]
    if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and b.shape == (B, N, K)):
        raise RuntimeError("bmm_nt: needs float32 GPU operands a (B, M, K), b (B, N, K)")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty((B, M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = native.lib().dvis_gemm_nt(ctypes.c_void_p(a.data_ptr()), K, M * K, ctypes.c_void_p(b.data_ptr()), K, N * K, None,
                                       None, 0, 0, ctypes.c_void_p(out.data_ptr()), N, M * N, M, N, K, B, 0,
                                       _gemm_config(M, N, K, B, config), native.stream_ptr(a.device))
    native.check(rc, "dvis_gemm_nt")
    return out


def _own_gemm_ok(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and not torch.is_grad_enabled()
            and x.shape[-1] % 4 == 0 and weight.stride(-1) == 1 and weight.stride(0) % 4 == 0
            and weight.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def linear(x, weight, bias=None, relu=False, own=None, tall=False, act=None, residual=None):
    """``F.linear`` (+ optional ReLU) on fp32 GPU inference tensors WITHOUT a library GEMM, in a form whose result for a row does
    not depend on how many rows share the call (the segmenter folds frames into the batch; a frame must get the same bits alone,
    in a 30-frame clip or in a rank's shard of it):
      tall=True   a projection over every pixel / token of every frame: the split-f16 matrix-core kernel (csrc/gemm_x3.hip)
                  where it is on and serves the shape, else the own exact GEMM from its ONE-WAVE family (K unsplit);
      default     the decoder's small GEMMs (rows = frames x queries): the own exact GEMM from the K-split family of its K
                  (4 waves, 8 from K = 1024 on) — dvis_gemm_pick_config_nw: every tile size of a family sums in the same order;
      own=True    the tracker / refiner call sites: the own GEMM with the size-driven configuration (their batch is pinned
                  with gemm_sizes_as), raising on operands it cannot serve.
    DVIS_DETERMINISTIC=0 hands the non-forced calls back to the library (hipBLASLt: faster on some tall exact shapes, bits that
    change with the row count).  CPU tensors, autograd, autocast / half precision always take torch ops."""
    forced = own is True
    gpu_inf = x.is_cuda and not torch.is_grad_enabled() and not torch.is_autocast_enabled()
    if tall and not forced and gpu_inf and x3_on() and weight.dim() == 2 and weight._base is None \
            and x3_tile_ok(x, weight.shape[0], weight.shape[1]):
        return x3_tile_linear(x, weight, bias, act="relu" if relu and act is None else act, residual=residual)
    if tall and not forced and gpu_inf and x3_on() and weight.dim() == 2 and weight._base is None \
            and x3_ok(x, weight.shape[0], weight.shape[1]):
        # Weights that are views (slices made per call) would be re-packed per call: they stay on the exact kernels below.
        if (act is not None or residual is not None) and weight.shape[0] % 256 == 0:
            return x3_linear(x, weight, bias, relu=relu, act=act, residual=residual)
        if act is None and residual is None:
            return x3_linear(x, weight, bias, relu=relu)
    if act is not None or residual is not None:
        # act ("gelu": the exact erf form of nn.GELU()) / residual (added after it): fused into the split-f16 kernel's epilogue
        # above; everywhere else the same composition in separate passes
        y = linear(x, weight, bias, relu=relu or act == "relu", own=own, tall=tall)
        if act == "gelu":
            y = torch.nn.functional.gelu(y)
        if residual is None:
            return y
        # in place only where that cannot change the result's dtype: under torch.autocast y is fp16 / bf16 and the reference's
        # `x = x + ls(attn(norm(x)))` promotes to the residual stream's fp32
        if not torch.is_grad_enabled() and y.dtype == residual.dtype and not torch.is_autocast_enabled() \
                and not torch.is_autocast_enabled("cpu"):
            return y.add_(residual)
        return residual + y
    own = OWN_GEMM_DEFAULT if own is None else own
    if own and x.is_cuda and not torch.is_grad_enabled() and (forced or not torch.is_autocast_enabled()):
        # (under torch.autocast — how the reference evaluates, train_net_video.py:259 — the projections are torch's half-precision
        # GEMMs: their fp16 / bf16 outputs feed the half-precision fused MSDeformAttn kernel)
        if not forced and not _own_gemm_ok(x, weight) and x.dtype == torch.float32 and weight.dtype == torch.float32 \
                and x.shape[-1] % 4 == 0:
            # a view whose rows are not 16-byte aligned / unit-strided (a slice of the last dim, a transposed weight): one copy
            # makes it servable — cheaper than letting the call fall to the library with other bits per batch size
            if x.stride(-1) != 1 or x.data_ptr() % 16:
                x = x.contiguous()
            if weight.stride(-1) != 1 or weight.stride(0) % 4 or weight.data_ptr() % 16:
                weight = weight.contiguous()
        if _own_gemm_ok(x, weight):
            cfg = -1
            if not forced:
                K = x.shape[-1]
                cfg = native.lib().dvis_gemm_pick_config_nw(max(1, x.numel() // K), weight.shape[0], K, 1,
                                                            1 if tall else (8 if K >= 1024 else 4))
            return gemm_nt(x, weight.detach(), None if bias is None else bias.detach().contiguous(), relu=relu, config=cfg)
        if forced:
            # tracker / refiner call sites: their stream must NEVER carry a library (stream-K) GEMM and every rank must
            # compute the same bits — a refused operand is an error whatever DVIS_STRICT says (as in gemm_nt itself)
            raise RuntimeError(f"linear(own=True): the own GEMM cannot serve these operands (needs fp32, K % 4 == 0, 16-byte "
                               f"aligned operands, weight row stride % 4 == 0; x {tuple(x.shape)} {x.dtype}, weight "
                               f"{tuple(weight.shape)} {weight.dtype} stride {tuple(weight.stride())})")
        # (K % 4 != 0 or mixed dtypes: no layer of the reference's configurations; the library GEMM below serves them)
    if relu and bias is not None and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled():
        K = x.shape[-1]
        y = torch._addmm_activation(bias, x.reshape(-1, K), weight.t(), use_gelu=False)
        return y.view(*x.shape[:-1], weight.shape[0])
    y = torch.nn.functional.linear(x, weight, bias)
    return torch.relu(y) if relu else y


def linear_relu(x, lin, own=None, tall=False):
    """relu(lin(x)) for an ``nn.Linear``."""
    return linear(x, lin.weight, lin.bias, relu=True, own=own, tall=tall)


# ---- the encoder's tall GEMMs on the F16 matrix cores (csrc/gemm_x3.hip): fp32 operands as two f16 terms, three products
# per pair, fp32 accumulation -> the error of an fp32 GEMM at 3/16 of its matrix-core time.
X3 = os.environ.get("DVIS_X3", "1") != "0"
# Stages of phase A whose layers stay on the exact-fp32 kernels although X3 is on: a comma-separated subset of
# backbone, pd_proj (the pixel decoder's input projections), mask_path (lateral 1x1 + FPN 3x3 + mask_features 1x1), encoder
# (MSDeformAttn projections + FFN), decoder_kv (the masked-attention decoder's key / value projections).  The callers mark their stage with `x3_stage(name)`.
X3_OFF = frozenset(v for v in os.environ.get("DVIS_X3_OFF", "").split(",") if v)
_x3_tls = threading.local()


class x3_stage:
    """``with x3_stage("encoder"): ...`` — names the stage the enclosed calls belong to (per thread), for X3_OFF."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = getattr(_x3_tls, "stage", None)
        _x3_tls.stage = self.name

    def __exit__(self, *exc):
        _x3_tls.stage = self.prev


def x3_on():
    """Split-f16 matrix-core kernels for the calling stage?  (the master switch X3 / DVIS_X3, minus the stages in X3_OFF, and not
    inside an `x3_disabled()` block)"""
    return X3 and not getattr(_x3_tls, "disabled", 0) and not (X3_OFF and getattr(_x3_tls, "stage", None) in X3_OFF)


class x3_disabled:
    """``with x3_disabled(): ...`` — the enclosed calls (this thread) take the exact-fp32 kernels: the re-run after an X3RangeError."""

    def __enter__(self):
        _x3_tls.disabled = getattr(_x3_tls, "disabled", 0) + 1

    def __exit__(self, *exc):
        _x3_tls.disabled -= 1


class X3RangeError(RuntimeError):
    """An activation left the f16 range of the split-f16 kernels (|x| >= 65520 / 2^xexp: 4095 in the linear / FFN kernels at
    DVIS_X3_XEXP = 4, 16380 in the convolutions at DVIS_X3_CONV_XEXP = 2): the layer's output rows are non-finite."""


# what DVIS_Plus_* do with an X3RangeError on a single GPU: "rerun" the clip on the exact-fp32 kernels with a warning (and keep
# that model on them from then on), or "raise".  Sharded runs always raise: the other ranks already hold the gathered queries.
X3_ON_OVERFLOW = os.environ.get("DVIS_X3_ON_OVERFLOW", "rerun")


class _X3RangeGuard:
    """Host side of the kernels' range guard (include/dvis_hip.h: dvis_x3_set_range_flag / dvis_x3_set_tag).  One sticky int32
    word per device receives the TAG of a launch that produced a non-finite pre-activation value; every packed weight gets a
    tag, so the word names the layer.  `snapshot()` (after phase A of a clip is enqueued) copies the word to pinned memory
    behind the clip's kernels — no synchronisation; `verify()` (before the clip's results are used) waits for that copy."""

    def __init__(self):
        self.words, self.pool, self.tags, self.lock = {}, {}, {}, threading.Lock()
        self.next_tag = 1

    def word(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        w = self.words.get(idx)
        if w is None:
            with self.lock:
                w = self.words.get(idx)
                if w is None:
                    w = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
                    with torch.cuda.device(idx):
                        native.check(native.lib().dvis_x3_set_range_flag(ctypes.c_void_p(w.data_ptr())), "dvis_x3_set_range_flag")
                    # (pool before words: a lock-free reader that finds the word must find its pinned buffers too)
                    self.pool[idx] = [[torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(16)], 0]
                    self.words[idx] = w
        return w

    def new_tag(self, weight, kind):
        import weakref
        with self.lock:
            tag = self.next_tag
            self.next_tag += 1
            self.tags[tag] = (kind, weakref.ref(weight), tuple(weight.shape))
        return tag

    def snapshot(self, device):
        if not (X3 and device.type == "cuda"):
            return None
        w = self.word(device)
        idx = w.device.index
        bufs = self.pool[idx]
        host = bufs[0][bufs[1] % len(bufs[0])]
        bufs[1] += 1
        host.copy_(w, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(w.device))
        # (4th entry: the word's value at this point of the stream as a DEVICE tensor — what a sharded run sends through the
        # clip's all-gather so that every rank takes the same decision, clip_shard.all_gather_frames)
        return (host, ev, w, w.clone())

    def describe(self, tag, model=None):
        kind, ref, shape = self.tags.get(tag, ("?", lambda: None, ()))
        wt, name = ref(), None
        if wt is not None and model is not None:
            for n, mod in model.named_modules():
                for pn, prm in list(mod.named_parameters(recurse=False)) + list(mod.named_buffers(recurse=False)):
                    if prm is wt:
                        name = f"{n}.{pn}"
                if name is None and getattr(mod, "_folded", None) is not None and mod._folded[1] is wt:
                    name = f"{n} (FrozenBN folded)"
                if name:
                    break
        return f"{name or 'weight'} {shape} [{kind} kernel]"

    def verify(self, snap, model=None):
        if snap is None:
            return
        host, ev, w = snap[:3]
        ev.synchronize()
        self.raise_if(int(host[0]), w, model)

    def verify_gathered(self, tags, device, model=None):
        """The collective form: `tags` = every rank's guard word as gathered with the clip's queries (float tensor).  Same
        data on every rank -> every rank raises (or none does)."""
        if tags is None:
            return
        self.raise_if(int(tags.max().item()), self.word(device), model)

    def raise_if(self, tag, w, model=None):
        if tag:
            w.zero_()
            raise X3RangeError(
                f"split-f16 range exceeded in {self.describe(tag, model)}: an input activation (or the FFN's hidden activation) "
                f"reached |x| >= 65520 / 2^xexp (xexp = {X3_XEXP} in the linear / FFN kernels, {X3_CONV_XEXP} in the convolutions) and "
                "the layer's outputs are non-finite.  Exact-fp32 kernels: DVIS_X3=0 (everything) or DVIS_X3_OFF=<stage,...> "
                "(backbone, pd_proj, mask_path, encoder, decoder_kv); more range at the cost of small-value precision: "
                "DVIS_X3_XEXP / DVIS_X3_CONV_XEXP.")

    def check_now(self, device, model=None):
        """Synchronous form (tests, one-off calls): reads the word after a device synchronisation."""
        self.verify(self.snapshot(device), model)


X3_GUARD = _X3RangeGuard()


def x3_range_snapshot(device):
    return X3_GUARD.snapshot(device)


def x3_range_verify(snap, model=None):
    X3_GUARD.verify(snap, model)


X3_XEXP = int(os.environ.get("DVIS_X3_XEXP", "4"))      # activations are scaled by 2^4 before the split (|x| < 4094)
# the convolutions see ReLU'd feature maps without a normalisation in front: more range (|x| < 16376), an absolute floor of
# 2^-27 = 7.5e-9 per element below |x| = 0.03
X3_CONV_XEXP = int(os.environ.get("DVIS_X3_CONV_XEXP", "2"))
# conv1x1_bias_act: from this many input channels on a 1x1 layer goes to the split-f16 kernel first — also the memory-bound
# ones csrc/conv1x1.hip (weights resident in LDS, exact fp32) was written for: 64 -> 256 + shortcut at 184 x 320 runs at 5.0
# TB/s there against 4.3, 256 -> 128 in 0.62 ms against 1.13 (tools/x3_time.py conv)
X3_CONV1X1_MIN_CI = int(os.environ.get("DVIS_X3_CONV1X1_MIN_CI", "64"))


class _PackCache:
    """Packed forms of weights, made once per (weight, kind of pack, weight version).  Keyed by (id(weight), kind) — the same
    tensor packed two ways (linear and FFN, 1x1 and 3x3) keeps both — with least-recently-used eviction past `cap` entries
    (a cleared-at-once dict re-packed every layer of every forward once a process held more than `cap` weights; each pack
    reads max|w| back to the host)."""

    def __init__(self, cap=4096):
        from collections import OrderedDict
        self.cap, self.d, self.lock = cap, OrderedDict(), threading.Lock()

    def get(self, key_obj, kind, version_key, make):
        k = (id(key_obj), kind)
        with self.lock:                # (stream()'s phase-B thread and the main thread both pack: the LRU order is shared state)
            ent = self.d.get(k)
            if ent is None or ent[0] != version_key:
                tag = ent[3] if ent is not None else X3_GUARD.new_tag(key_obj, kind)
                self.d[k] = ent = (version_key, make(), key_obj, tag)       # (holds key_obj: id() stays unique)
                while len(self.d) > self.cap:
                    self.d.popitem(last=False)
            else:
                self.d.move_to_end(k)
        # the launch that follows carries this weight's tag (range guard; per host thread) — and the device's guard word exists
        X3_GUARD.word(key_obj.device)
        native.lib().dvis_x3_set_tag(ent[3])
        return ent[1]

    def __len__(self):
        return len(self.d)

    def clear(self):
        self.d.clear()


_X3_PACKED = _PackCache()


def _x3_exp(w):
    """e with max|w| * 2^e in [2^13, 2^14): the weight fills the f16 range, its low term stays a normal number."""
    m = float(w.detach().abs().max())
    return 0 if m == 0.0 or m != m else 14 - math.frexp(m)[1]


def _x3_cache(key_obj, version_key, make, kind="linear"):
    return _X3_PACKED.get(key_obj, kind, version_key, make)


def x3_pack(weight):
    """(packed uint8 buffer, wexp) of an (N, K) float32 GPU weight, made once per weight version."""
    def make():
        w = weight.detach()
        if w.stride(1) != 1:
            w = w.contiguous()
        N, K = w.shape
        nbytes = native.lib().dvis_x3_packed_bytes(N, K)
        if nbytes <= 0:
            raise RuntimeError(f"x3_pack: weight {tuple(w.shape)} is not served (K % 32 == 0)")
        e = _x3_exp(w)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        with torch.cuda.device(w.device):
            native.check(native.lib().dvis_x3_pack(ctypes.c_void_p(w.data_ptr()), w.stride(0), N, K, e,
                                                   ctypes.c_void_p(buf.data_ptr()), native.stream_ptr(w.device)), "dvis_x3_pack")
        return buf, e
    return _x3_cache(weight, (weight._version, weight.data_ptr(), weight.device, tuple(weight.shape)), make)


# The large-K tall GEMMs (ViT blocks: K = 1024 .. 4096, N a multiple of 256) take the TILED split-f16 kernel (csrc/gemm_x3_tile.hip:
# both operands through LDS, one split of the rows per 256 output features); the encoder's K = 256 layers keep the streaming
# kernels with their fused LayerNorm / position forms.  DVIS_X3_TILE_MIN_K (development): K from which the tiled kernel is used.
X3_TILE_MIN_K = int(os.environ.get("DVIS_X3_TILE_MIN_K", "512"))


def x3_tile_ok(x, N, K):
    return (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[-1] == K and K >= X3_TILE_MIN_K
            and bool(native.lib().dvis_x3_tile_supported(N, K)))


def x3_tile_pack(weight, order=0):
    """(packed buffer, wexp) of an (N, K) float32 GPU weight in the tiled kernel's LDS image, made once per weight version.
    order: the k-slot order of the row image the weight meets (RowImage.order; 0 for fp32 rows)."""
    def make():
        w = weight.detach()
        if w.stride(1) != 1:
            w = w.contiguous()
        N, K = w.shape
        nbytes = native.lib().dvis_x3_tile_packed_bytes(N, K)
        if nbytes <= 0:
            raise RuntimeError(f"x3_tile_pack: weight {tuple(w.shape)} is not served (N % 256 == 0, K % 32 == 0)")
        e = _x3_exp(w)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        with torch.cuda.device(w.device):
            native.check(native.lib().dvis_x3_tile_pack_order(ctypes.c_void_p(w.data_ptr()), w.stride(0), N, K, e, order,
                                                              ctypes.c_void_p(buf.data_ptr()), native.stream_ptr(w.device)),
                         "dvis_x3_tile_pack_order")
        return buf, e
    return _x3_cache(weight, (weight._version, weight.data_ptr(), weight.device, tuple(weight.shape)), make,
                     kind="tile" if order == 0 else f"tile order {order}")


# Row images (include/dvis_hip.h, "ROW IMAGES"; csrc/gemm_x3_tile.hip): the tiled GEMM's row operand pre-split by its producer.
# DVIS_X3_ROW_IMAGES=0 (development): the ViT blocks keep fp32 activations between their layers.
X3_ROW_IMAGES = os.environ.get("DVIS_X3_ROW_IMAGES", "1") != "0"


class RowImage:
    """fp32 rows (..., K) as the row operand image of dvis_x3_tile_linear_image: `data` (uint8, dvis_x3_rows_image_bytes), `shape` of
    the tensor it stands for, `exp` (values x 2^exp before the split), `order` (k-slot order: the weights are packed to match)."""

    def __init__(self, data, shape, exp, order):
        self.data, self.shape, self.exp, self.order = data, tuple(shape), exp, order

    @property
    def device(self):
        return self.data.device

    @property
    def rows(self):
        n = 1
        for d in self.shape[:-1]:
            n *= d
        return n


def _rows_image_buffer(M, K, device):
    nbytes = native.lib().dvis_x3_rows_image_bytes(M, K)
    if nbytes < 0:
        raise RuntimeError(f"row image: K % 32 == 0 is required (K {K})")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def x3_rows_image(x, xexp=None):
    """fp32 GPU rows (..., K) -> RowImage (order 0): the stand-alone producer."""
    K = x.shape[-1]
    x2, ldx = _x3_rows(x, "x")
    e = X3_XEXP if xexp is None else xexp
    buf = _rows_image_buffer(x2.shape[0], K, x.device)
    with torch.cuda.device(x.device):
        native.check(native.lib().dvis_x3_rows_image(ctypes.c_void_p(x2.data_ptr()), ldx, x2.shape[0], K, e, ctypes.c_void_p(buf.data_ptr()),
                                                     native.stream_ptr(x.device)), "dvis_x3_rows_image")
    return RowImage(buf, x.shape, e, 0)


def layer_norm_rows_image_ok(x, norm):
    C = x.shape[-1]
    return (X3_ROW_IMAGES and x3_on() and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.data_ptr() % 16 == 0
            and C % 32 == 0 and C <= 1024 and norm.weight is not None and norm.bias is not None and not torch.is_grad_enabled()
            and not torch.is_autocast_enabled())


def layer_norm_rows_image(x, norm, xexp=None):
    """``norm(x)`` (nn.LayerNorm over the last dim) written as the RowImage (order 0) of the GEMM that reads it — the value every
    element holds is add_layer_norm's, split into its two f16 terms."""
    C = x.shape[-1]
    M = x.numel() // C
    e = X3_XEXP if xexp is None else xexp
    buf = _rows_image_buffer(M, C, x.device)
    with torch.cuda.device(x.device):
        native.check(native.lib().dvis_layernorm_rows_image(
            ctypes.c_void_p(x.data_ptr()), native.dev_ptr(norm.weight, "gamma"), native.dev_ptr(norm.bias, "beta"), M, C, float(norm.eps), e,
            ctypes.c_void_p(buf.data_ptr()), native.stream_ptr(x.device)), "dvis_layernorm_rows_image")
    return RowImage(buf, x.shape, e, 0)


def x3_tile_linear(x, weight, bias, act=None, residual=None, xexp=None):
    """``act(x @ weight.T + bias) + residual`` through dvis_x3_tile_linear (act: None | "relu" | "gelu")."""
    N, K = weight.shape
    if isinstance(x, RowImage):
        return _x3_tile_linear_image(x, weight, bias, act, residual)
    x2, ldx = _x3_rows(x, "x")
    buf, wexp = x3_tile_pack(weight)
    out = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)
    rptr, ldr = None, 0
    if residual is not None:
        if residual.shape != out.shape or residual.dtype != torch.float32 or not residual.is_cuda:
            raise RuntimeError("x3_tile_linear: residual must be a float32 GPU tensor of the output's shape")
        r2, ldr = _rows2d(residual, N)
        rptr = ctypes.c_void_p(r2.data_ptr())
    with torch.cuda.device(x.device):
        native.check(native.lib().dvis_x3_tile_linear(
            ctypes.c_void_p(x2.data_ptr()), ldx, x2.shape[0], K, ctypes.c_void_p(buf.data_ptr()), N,
            X3_XEXP if xexp is None else xexp, wexp, None if bias is None else native.dev_ptr(bias.detach(), "bias"),
            {None: 0, "relu": 1, "gelu": 2}[act], rptr, ldr, ctypes.c_void_p(out.data_ptr()), N, native.stream_ptr(x.device)),
            "dvis_x3_tile_linear")
    return out


def _x3_tile_linear_image(img, weight, bias, act, residual):
    """x3_tile_linear on a RowImage.  act None / "relu": fp32 rows (+ residual); act "gelu": the result as the next GEMM's RowImage."""
    N, K = weight.shape
    if img.shape[-1] != K:
        raise RuntimeError(f"x3_tile_linear: image of {img.shape} rows against a weight of {tuple(weight.shape)}")
    M = img.rows
    buf, wexp = x3_tile_pack(weight, img.order)
    bptr = None if bias is None else native.dev_ptr(bias.detach(), "bias")
    lib = native.lib()
    with torch.cuda.device(img.device):
        if act == "gelu":
            if residual is not None:
                raise RuntimeError("x3_tile_linear: the GELU form of a row image writes a row image (no residual)")
            out = _rows_image_buffer(M, N, img.device)
            native.check(lib.dvis_x3_tile_linear_image(ctypes.c_void_p(img.data.data_ptr()), M, K, ctypes.c_void_p(buf.data_ptr()), N, img.exp, wexp,
                                                       bptr, 2, None, 0, None, 0, ctypes.c_void_p(out.data_ptr()), X3_XEXP,
                                                       native.stream_ptr(img.device)), "dvis_x3_tile_linear_image")
            return RowImage(out, (*img.shape[:-1], N), X3_XEXP, 1)
        out = torch.empty((*img.shape[:-1], N), dtype=torch.float32, device=img.device)
        rptr, ldr = None, 0
        if residual is not None:
            if residual.shape != out.shape or residual.dtype != torch.float32 or not residual.is_cuda:
                raise RuntimeError("x3_tile_linear: residual must be a float32 GPU tensor of the output's shape")
            r2, ldr = _rows2d(residual, N)
            rptr = ctypes.c_void_p(r2.data_ptr())
        native.check(lib.dvis_x3_tile_linear_image(ctypes.c_void_p(img.data.data_ptr()), M, K, ctypes.c_void_p(buf.data_ptr()), N, img.exp, wexp,
                                                   bptr, {None: 0, "relu": 1}[act], rptr, ldr, ctypes.c_void_p(out.data_ptr()), N, None, 0,
                                                   native.stream_ptr(img.device)), "dvis_x3_tile_linear_image")
    return out


X3_QKV_FUSED = os.environ.get("DVIS_X3_QKV_FUSED", "1") != "0"


def x3_qkv_attention_ok(x, weight, heads):
    """Can ``self_attention(x @ weight.T + bias)`` take the fused form (dvis_x3_tile_linear_qkv + dvis_attention_x3_packed)?
    x (B, L, C) float32 GPU inference tensor, weight (3C, C), head dim 64, L >= 1024 (the ViT blocks)."""
    if isinstance(x, RowImage):
        if len(x.shape) != 3 or not (X3_QKV_FUSED and weight.dim() == 2 and weight._base is None):
            return False
        B, L, C = x.shape
        return (C == heads * 64 and weight.shape == (3 * C, C) and L >= 1024 and C >= X3_TILE_MIN_K
                and bool(native.lib().dvis_x3_tile_supported(3 * C, C)))
    if not (X3_QKV_FUSED and x.dim() == 3 and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and not torch.is_autocast_enabled() and x3_on() and weight.dim() == 2 and weight._base is None):
        return False
    B, L, C = x.shape
    return (C == heads * 64 and weight.shape == (3 * C, C) and L >= 1024 and x.is_contiguous() and x.data_ptr() % 16 == 0
            and x3_tile_ok(x, 3 * C, C))


def x3_qkv_attention(x, weight, bias, heads, out_image=False):
    """softmax(q k^T / 8) v per head for qkv = x @ weight.T + bias (columns ordered q | k | v, head, dim), x (B, L, C), head dim 64:
    the projection's epilogue writes the split-f16 attention kernel's operand images, the attention kernel reads them — the fp32
    qkv tensor and the pack pass do not exist.  x: fp32 tensor or RowImage.  -> (B, L, C) fp32, or with out_image its RowImage
    (order 2: the out-projection's row operand)."""
    B, L, C = x.shape
    image_in = isinstance(x, RowImage)
    buf, wexp = x3_tile_pack(weight, x.order if image_in else 0)
    lib = native.lib()
    nbytes = lib.dvis_attention_ws_bytes_k(B * heads, L, L, 64, 2)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    qscale = (1.0 / 8.0) * 1.4426950408889634 * 16.0
    bptr = None if bias is None else native.dev_ptr(bias.detach(), "bias")
    with torch.cuda.device(x.device):
        if image_in:
            native.check(lib.dvis_x3_tile_linear_qkv_image(
                ctypes.c_void_p(x.data.data_ptr()), B * L, C, ctypes.c_void_p(buf.data_ptr()), 3 * C, x.exp, wexp, bptr, heads, L, qscale,
                ctypes.c_void_p(ws.data_ptr()), native.stream_ptr(x.device)), "dvis_x3_tile_linear_qkv_image")
        else:
            native.check(lib.dvis_x3_tile_linear_qkv(
                ctypes.c_void_p(x.data_ptr()), C, B * L, C, ctypes.c_void_p(buf.data_ptr()), 3 * C, X3_XEXP, wexp, bptr, heads, L, qscale,
                ctypes.c_void_p(ws.data_ptr()), native.stream_ptr(x.device)), "dvis_x3_tile_linear_qkv")
        if out_image:
            out = _rows_image_buffer(B * L, C, x.device)
            native.check(lib.dvis_attention_x3_packed_image(ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, heads, L, X3_XEXP,
                                                            native.stream_ptr(x.device)), "dvis_attention_x3_packed_image")
            return RowImage(out, (B, L, C), X3_XEXP, 2)
        out = torch.empty((B, L, C), dtype=torch.float32, device=x.device)
        strides = (ctypes.c_int64 * 3)(L * C, 64, C)
        native.check(lib.dvis_attention_x3_packed(ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(out.data_ptr()), strides, B, heads, L,
                                                  native.stream_ptr(x.device)), "dvis_attention_x3_packed")
    return out


def x3_ok(x, N, K, ln=False, add=False):
    """Does csrc/gemm_x3.hip serve ``x (..., K) @ W (N, K).T``?  ln: the output_proj + residual + LayerNorm form; add: the form
    with the in-kernel ``x + xadd`` (dvis_x3_linear_add: K = 256 only)."""
    return (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[-1] == K
            and (not add or K == 256) and bool(native.lib().dvis_x3_linear_supported(N, K, int(ln))))


def _x3_rows(x, name):
    K = x.shape[-1]
    x2, ld = _rows2d(x, K)
    if x2.data_ptr() % 16 or ld % 4:
        raise RuntimeError(f"x3: {name} must be 16-byte aligned with a row stride % 4 == 0")
    return x2, ld


def x3_linear(x, weight, bias, relu=False, xexp=None, xadd=None, act=None, residual=None):
    """``relu?((x + xadd) @ weight.T + bias)`` through dvis_x3_linear[_add]; with act ("gelu" | "relu" | None) / residual:
    ``act(x @ weight.T + bias) + residual`` through dvis_x3_linear_res (GELU and the residual add inside the GEMM's epilogue).  x (..., K) float32 GPU; weight (N, K); xadd: None or
    a (1, S, 256) / (S, 256) embedding for x of shape (B, S, 256), broadcast over B — added inside the kernel while it builds the
    row's fragments, the sum is never written (K = 256, any N the kernels serve: all passes over N run from one read of the row)."""
    N, K = weight.shape
    x2, ldx = _x3_rows(x, "x")
    buf, wexp = x3_pack(weight)
    out = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)
    if act is not None or residual is not None:
        if xadd is not None:
            raise RuntimeError("x3_linear: xadd and act / residual are served by different entry points")
        code = {None: int(bool(relu)), "relu": 1, "gelu": 2}[act]
        rptr, ldr = None, 0
        if residual is not None:
            if residual.shape != out.shape or residual.dtype != torch.float32 or not residual.is_cuda:
                raise RuntimeError("x3_linear: residual must be a float32 GPU tensor of the output's shape")
            r2, ldr = _x3_rows(residual, "residual")
            rptr = ctypes.c_void_p(r2.data_ptr())
        with torch.cuda.device(x.device):
            native.check(native.lib().dvis_x3_linear_res(
                ctypes.c_void_p(x2.data_ptr()), ldx, x2.shape[0], K, ctypes.c_void_p(buf.data_ptr()), N,
                X3_XEXP if xexp is None else xexp, wexp, None if bias is None else native.dev_ptr(bias.detach(), "bias"), code,
                rptr, ldr, ctypes.c_void_p(out.data_ptr()), N, native.stream_ptr(x.device)), "dvis_x3_linear_res")
        return out
    if xadd is not None:
        if x.dim() != 3 or xadd.numel() != x.shape[1] * K or xadd.dtype != torch.float32 or not xadd.is_cuda:
            raise RuntimeError("x3_linear: xadd must be a float32 GPU (S, K) embedding for x of shape (B, S, K)")
        xadd = xadd.contiguous()
        with torch.cuda.device(x.device):
            native.check(native.lib().dvis_x3_linear_add(
                ctypes.c_void_p(x2.data_ptr()), ldx, x2.shape[0], K, ctypes.c_void_p(buf.data_ptr()), N,
                X3_XEXP if xexp is None else xexp, wexp, ctypes.c_void_p(xadd.data_ptr()), x.shape[1],
                None if bias is None else native.dev_ptr(bias.detach(), "bias"), int(relu), ctypes.c_void_p(out.data_ptr()), N,
                native.stream_ptr(x.device)), "dvis_x3_linear_add")
        return out
    with torch.cuda.device(x.device):
        native.check(native.lib().dvis_x3_linear(
            ctypes.c_void_p(x2.data_ptr()), ldx, x2.shape[0], K, ctypes.c_void_p(buf.data_ptr()), N,
            X3_XEXP if xexp is None else xexp, wexp, None if bias is None else native.dev_ptr(bias.detach(), "bias"), int(relu),
            ctypes.c_void_p(out.data_ptr()), N, native.stream_ptr(x.device)), "dvis_x3_linear")
    return out


def _x3_pos(pos, x, N):
    if pos is None:
        return None, 0, None
    if x.dim() != 3 or pos.numel() != x.shape[1] * N or pos.dtype != torch.float32 or not pos.is_cuda:
        raise RuntimeError("x3: pos must be a float32 GPU (S, C) embedding for x of shape (N, S, C)")
    return pos.contiguous(), x.shape[1], torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)


def x3_linear_ln(x, weight, bias, res, norm, pos=None, xexp=None):
    """``norm(res + x @ weight.T + bias)`` (+ second output ``out + pos``) in one kernel (dvis_x3_linear_ln)."""
    N, K = weight.shape
    x2, ldx = _x3_rows(x, "x")
    r2, ldr = _x3_rows(res, "res")
    buf, wexp = x3_pack(weight)
    out = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)
    pos, pos_rows, out2 = _x3_pos(pos, x, N)
    with torch.cuda.device(x.device):
        native.check(native.lib().dvis_x3_linear_ln(
            ctypes.c_void_p(x2.data_ptr()), ldx, x2.shape[0], K, ctypes.c_void_p(buf.data_ptr()), N,
            X3_XEXP if xexp is None else xexp, wexp, native.dev_ptr(bias.detach(), "bias"), ctypes.c_void_p(r2.data_ptr()), ldr,
            native.dev_ptr(norm.weight.detach(), "gamma"), native.dev_ptr(norm.bias.detach(), "beta"), float(norm.eps),
            None if pos is None else ctypes.c_void_p(pos.data_ptr()), pos_rows, ctypes.c_void_p(out.data_ptr()),
            None if out2 is None else ctypes.c_void_p(out2.data_ptr()), N, native.stream_ptr(x.device)), "dvis_x3_linear_ln")
    return out if out2 is None else (out, out2)


def x3_ffn_ok(x, lin1, lin2):
    K, H, N = lin1.weight.shape[1], lin1.weight.shape[0], lin2.weight.shape[0]
    return (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and x.shape[-1] == K
            and lin2.weight.shape[1] == H and lin1.bias is not None and lin2.bias is not None
            and native.lib().dvis_x3_ffn_packed_bytes(K, H, N) > 0)


def x3_ffn_ln(x, lin1, lin2, norm, pos=None, xexp=None, hexp=None):
    """``norm(x + lin2(relu(lin1(x))))`` (+ ``out + pos``) in one kernel (dvis_x3_ffn_ln): the hidden tensor stays on chip."""
    w1, w2 = lin1.weight, lin2.weight
    H, K = w1.shape
    N = w2.shape[0]
    x2, ldx = _x3_rows(x, "x")

    def make():
        a, b = w1.detach().contiguous(), w2.detach().contiguous()
        e1, e2 = _x3_exp(a), _x3_exp(b)
        buf = torch.empty(native.lib().dvis_x3_ffn_packed_bytes(K, H, N), dtype=torch.uint8, device=a.device)
        with torch.cuda.device(a.device):
            native.check(native.lib().dvis_x3_ffn_pack(ctypes.c_void_p(a.data_ptr()), K, ctypes.c_void_p(b.data_ptr()), H, K, H,
                                                       N, e1, e2, ctypes.c_void_p(buf.data_ptr()),
                                                       native.stream_ptr(a.device)), "dvis_x3_ffn_pack")
        return buf, e1, e2
    buf, e1, e2 = _x3_cache(w1, (w1._version, w2._version, w1.data_ptr(), w2.data_ptr(), w1.device), make, kind="ffn")
    out = torch.empty((*x.shape[:-1], N), dtype=torch.float32, device=x.device)
    pos, pos_rows, out2 = _x3_pos(pos, x, N)
    with torch.cuda.device(x.device):
        native.check(native.lib().dvis_x3_ffn_ln(
            ctypes.c_void_p(x2.data_ptr()), ldx, x2.shape[0], K, H, N, ctypes.c_void_p(buf.data_ptr()),
            X3_XEXP if xexp is None else xexp, e1, X3_XEXP if hexp is None else hexp, e2,
            native.dev_ptr(lin1.bias.detach(), "b1"), native.dev_ptr(lin2.bias.detach(), "b2"),
            native.dev_ptr(norm.weight.detach(), "gamma"), native.dev_ptr(norm.bias.detach(), "beta"), float(norm.eps),
            None if pos is None else ctypes.c_void_p(pos.data_ptr()), pos_rows, ctypes.c_void_p(out.data_ptr()),
            None if out2 is None else ctypes.c_void_p(out2.data_ptr()), N, native.stream_ptr(x.device)), "dvis_x3_ffn_ln")
    return out if out2 is None else (out, out2)


def maps_to_tokens(maps, affines=None, pos=None):
    """[(N, C, h_l, w_l)] -> (N, sum h_l*w_l, C): ``torch.cat([m.flatten(2).transpose(1, 2) for m in maps], 1)`` with one
    tiled transpose per level instead of torch's strided copy.  CPU / non-fp32 / non-contiguous inputs use the torch ops.
    affines: per level None or (scale, shift) from ``group_norm_affine`` — the map is read as m * scale + shift.
    pos: optional (1, S, C) / (S, C) token-major embedding: returns (tokens, tokens + pos), both written in the one pass."""
    affines = affines or [None] * len(maps)
    N, C = maps[0].shape[:2]
    if not all(m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() for m in maps) or torch.is_grad_enabled():
        _torch_path("maps_to_tokens", maps[0], "needs contiguous fp32 NCHW maps")
        maps = [m if a is None else m * a[0].view(N, C, 1, 1) + a[1].view(N, C, 1, 1) for m, a in zip(maps, affines)]
        out = torch.cat([m.flatten(2).transpose(1, 2) for m in maps], 1)
        return out if pos is None else (out, out + pos.reshape(1, -1, C))
    S = sum(m.shape[2] * m.shape[3] for m in maps)
    out = torch.empty((N, S, C), dtype=torch.float32, device=maps[0].device)
    out_pos = pp = None
    if pos is not None:
        if pos.numel() != S * C or pos.dtype != torch.float32 or not pos.is_cuda:
            raise RuntimeError("maps_to_tokens: pos must be a float32 GPU (S, C) embedding")
        pos = pos.contiguous()
        out_pos = torch.empty_like(out)
        pp = native.dev_ptr(pos, "pos")
    row0 = 0
    with torch.cuda.device(out.device):
        for m, a in zip(maps, affines):
            HW = m.shape[2] * m.shape[3]
            rc = native.lib().dvis_nchw_to_tokens_affine(
                ctypes.c_void_p(m.data_ptr()), None if a is None else native.dev_ptr(a[0], "scale"),
                None if a is None else native.dev_ptr(a[1], "shift"), pp, ctypes.c_void_p(out.data_ptr()),
                None if out_pos is None else ctypes.c_void_p(out_pos.data_ptr()), N, C, HW, S, row0,
                native.stream_ptr(out.device))
            native.check(rc, "dvis_nchw_to_tokens_affine")
            row0 += HW
    return out if pos is None else (out, out_pos)


def normalize_pad_ok(x, mean):
    return bool(x.is_cuda and x.dim() == 4 and x.dtype in (torch.uint8, torch.float32) and x.is_contiguous()
                and mean.numel() == x.shape[1] and mean.dtype == torch.float32 and not torch.is_grad_enabled())


def normalize_pad(x, mean, std, Hp, Wp):
    """zero-padded ((x - mean[c]) / std[c]) of (N, C, H, W) uint8 / fp32 frames as ONE pass: conversion, normalisation and
    ImageList.from_tensors' padding (dvis_Plus/meta_architecture.py:1310-1311).  The same fp32 subtract and divide: same bits."""
    N, C, H, W = x.shape
    out = torch.empty((N, C, Hp, Wp), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        native.check(native.lib().dvis_normalize_pad(
            native.dev_ptr(x, "frames"), int(x.dtype == torch.uint8), native.dev_ptr(out, "out"), N * C, C, H, W, Hp, Wp,
            native.dev_ptr(mean.detach().reshape(-1), "pixel_mean"), native.dev_ptr(std.detach().reshape(-1), "pixel_std"),
            native.stream_ptr(x.device)), "dvis_normalize_pad")
    return out


def tokens_to_map(tokens, row0, h, w):
    """``tokens[:, row0 : row0 + h * w].transpose(1, 2).reshape(N, C, h, w)`` as a CONTIGUOUS map (tokens: (N, S, C) float32):
    one tiled transpose (dvis_tokens_to_nchw) instead of the strided view whose consumer copies it with torch's generic kernel
    (452 MB at the finest encoder level of 30 frames: 0.75 ms).  CPU / other dtypes / autograd: the view."""
    N, S, C = tokens.shape
    if not (tokens.is_cuda and tokens.dtype == torch.float32 and tokens.is_contiguous() and not torch.is_grad_enabled()):
        return tokens[:, row0:row0 + h * w].transpose(1, 2).reshape(N, C, h, w)
    out = torch.empty((N, C, h, w), dtype=torch.float32, device=tokens.device)
    with torch.cuda.device(tokens.device):
        native.check(native.lib().dvis_tokens_to_nchw(ctypes.c_void_p(tokens.data_ptr()), ctypes.c_void_p(out.data_ptr()), N, C, h * w,
                                                      S, row0, native.stream_ptr(tokens.device)), "dvis_tokens_to_nchw")
    return out


def conv1x1(x, weight, bias=None):
    """1x1 stride-1 convolution on NCHW.  On the GPU a channel-REDUCING 1x1 conv (Ci >= Co) is issued as the batched
    library GEMM W (Co,Ci) @ X (N,Ci,HW) instead of a MIOpen convolution: same contraction, same layout, measured on
    MI355X at the R50 / pixel-decoder shapes (tools/conv1x1_probe.py): 256->64 @184x320 1.10 -> 0.56 ms, 64->64
    0.55 -> 0.24 ms, 2048->256 0.32 -> 0.25 ms; channel-expanding ones are faster through MIOpen and stay there."""
    Co, Ci = weight.shape[:2]
    if conv1x1_x3_ok(x, weight):
        return conv1x1_x3(x, weight, bias)            # split-f16 matrix-core kernel (lateral / input / mask-feature projections)
    if bias is not None and CONV1X1_MFMA and x.is_cuda and x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32 \
            and weight.dtype == torch.float32 and not torch.is_grad_enabled() \
            and native.lib().dvis_conv1x1_mfma_supported(Ci, Co, x.shape[2] * x.shape[3]):
        return conv1x1_mfma(x, weight, bias)          # contraction + bias in one kernel (the pixel decoder's input projections)
    if x.is_cuda and Ci >= Co and x.is_contiguous() and x.dim() == 4 and x.dtype == weight.dtype \
            and not torch.is_grad_enabled():
        N, _, H, W = x.shape
        # bmm with a stride-0 batch of W, not torch.matmul: matmul folds the batch (transpose + copy of X, 4x slower)
        # whenever the 2-D operand is a Parameter that requires grad, even under no_grad
        y = torch.bmm(weight.detach().view(1, Co, Ci).expand(N, Co, Ci), x.view(N, Ci, H * W)).view(N, Co, H, W)
        return y if bias is None else bias_act_(y, bias.detach(), None, relu=False)   # in place, at the HBM stream rate
    # channel-expanding, strided (e.g. channels-last) or autograd inputs: MIOpen — a library call either way
    return torch.nn.functional.conv2d(x, weight, bias)


def conv1x1_bias_act(x, weight, bias=None, res=None, relu=False):
    """relu?(conv1x1(x, weight) + bias[c] + res) on NCHW.  Shapes dvis_conv1x1_bias_act serves (the memory-bound 1x1
    layers of the bottleneck at the large maps) run as ONE kernel — contraction and epilogue, no second pass over the
    output; every other shape is the library contraction followed by the in-place ``bias_act_`` pass."""
    Co, Ci = weight.shape[:2]
    if x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.is_contiguous() \
            and not torch.is_grad_enabled() and (res is None or (res.is_contiguous() and res.dtype == torch.float32)):
        N, _, H, W = x.shape
        if Ci >= X3_CONV1X1_MIN_CI and conv1x1_x3_ok(x, weight, 1, res):
            return conv1x1_x3(x, weight, bias, res, relu)
        if native.lib().dvis_conv1x1_supported(Ci, Co, H * W) and (res is None or res.shape == (N, Co, H, W)):
            w2 = weight.detach().reshape(Co, Ci)
            if not w2.is_contiguous():
                w2 = w2.contiguous()
            out = torch.empty((N, Co, H, W), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                rc = native.lib().dvis_conv1x1_bias_act(
                    native.dev_ptr(x, "x"), native.dev_ptr(w2, "weight"),
                    None if bias is None else native.dev_ptr(bias.detach(), "bias"),
                    None if res is None else native.dev_ptr(res, "res"), native.dev_ptr(out, "out"), N, Ci, Co, H * W,
                    1 if relu else 0, native.stream_ptr(x.device))
            native.check(rc, "dvis_conv1x1_bias_act")
            return out
        if CONV1X1_MFMA and native.lib().dvis_conv1x1_mfma_supported(Ci, Co, H * W) and (res is None or res.shape == (N, Co, H, W)):
            return conv1x1_mfma(x, weight, bias, res, relu)
    return bias_act_(conv1x1(x, weight), None if bias is None else bias.detach(), res, relu)


def conv1x1_x3_ok(x, weight, stride=1, res=None):
    """Does csrc/conv1x1_x3.hip (split-f16 matrix-core arithmetic, see csrc/gemm_x3.hip) serve this 1x1 convolution?"""
    if not (x3_on() and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.is_contiguous()
            and not torch.is_grad_enabled()):
        return False
    N, Ci, H, W = x.shape
    Co = weight.shape[0]
    OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
    if res is not None and not (res.is_contiguous() and res.dtype == torch.float32 and tuple(res.shape) == (N, Co, OH, OW)):
        return False
    # (per IMAGE: a batch whose tensors exceed the kernel's 2 GiB of 32-bit offsets is launched in chunks of images by
    # _conv_x3_chunks — the dispatch, and with it a frame's bits, must not depend on how many frames share the call)
    return N == 0 or bool(native.lib().dvis_conv1x1_x3_supported(Ci, Co, 1, H * W, OH * OW))


def _conv_x3_chunks(N, Ci, Co, HW_in, HW_out):
    """Images per launch of the split-f16 convolution kernels: both tensors of a launch stay below 2 GiB."""
    per = max(Ci * HW_in, Co * HW_out) * 4
    return max(1, min(N, (2 ** 31 - 1) // per))


def conv1x1_x3(x, weight, bias=None, res=None, relu=False, stride=1, xexp=None):
    """relu?(conv1x1(x, weight)[:, :, ::stride, ::stride] + bias[c] + res) through dvis_conv1x1_x3."""
    N, Ci, H, W = x.shape
    Co = weight.shape[0]

    def make():
        w2 = weight.detach().reshape(Co, Ci).contiguous()
        e = _x3_exp(w2)
        buf = torch.empty(native.lib().dvis_conv1x1_x3_packed_bytes(Ci, Co), dtype=torch.uint8, device=w2.device)
        with torch.cuda.device(w2.device):
            native.check(native.lib().dvis_conv1x1_x3_pack(native.dev_ptr(w2, "weight"), Co, Ci, e, ctypes.c_void_p(buf.data_ptr()),
                                                           native.stream_ptr(w2.device)), "dvis_conv1x1_x3_pack")
        return buf, e
    buf, wexp = _x3_cache(weight, (weight._version, weight.data_ptr(), weight.device), make, kind="conv1x1")
    OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
    out = torch.empty((N, Co, OH, OW), dtype=torch.float32, device=x.device)
    step = _conv_x3_chunks(N, Ci, Co, H * W, OH * OW)
    with torch.cuda.device(x.device):
        for i in range(0, N, step):
            n = min(step, N - i)
            native.check(native.lib().dvis_conv1x1_x3(
                native.dev_ptr(x[i:i + n], "x"), ctypes.c_void_p(buf.data_ptr()),
                None if bias is None else native.dev_ptr(bias.detach(), "bias"),
                None if res is None else native.dev_ptr(res[i:i + n], "res"), native.dev_ptr(out[i:i + n], "out"), n, Ci, Co, H, W,
                stride, X3_CONV_XEXP if xexp is None else xexp, wexp, 1 if relu else 0, native.stream_ptr(x.device)),
                "dvis_conv1x1_x3")
    return out


X3_DUAL = os.environ.get("DVIS_X3_DUAL", "1") != "0"


def conv1x1_x3_dual_ok(a, w3, x, ws, stride2):
    """conv3(a) + shortcut(x) of a bottleneck with a projection shortcut as ONE launch of the split-f16 kernel?"""
    if not (X3_DUAL and conv1x1_x3_ok(a, w3) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()
            and ws.dtype == torch.float32 and ws.shape[0] == w3.shape[0] and stride2 in (1, 2)):
        return False
    N, C, H, W = a.shape
    N2, C2, H2, W2 = x.shape
    Co = w3.shape[0]
    return (N2 == N and C2 % 64 == 0 and 64 <= C2 <= 4096 and (H2 + stride2 - 1) // stride2 == H and (W2 + stride2 - 1) // stride2 == W
            and (C + C2) <= 4096 and bool(native.lib().dvis_conv1x1_x3_supported(C + C2, Co, 1, H * W, H * W))
            and C2 * H2 * W2 * 4 < 2 ** 31)


def conv1x1_x3_dual(a, w3, b3, x, ws, bs, relu=True, stride2=1, xexp=None):
    """relu?(conv1x1(a, w3) + b3 + conv1x1(x, ws)[:, :, ::stride2, ::stride2] + bs) through dvis_conv1x1_x3_dual."""
    N, C, H, W = a.shape
    _, C2, H2, W2 = x.shape
    Co = w3.shape[0]

    def make():
        wcat = torch.cat([w3.detach().reshape(Co, C), ws.detach().reshape(Co, C2)], 1).contiguous()
        e = _x3_exp(wcat)
        buf = torch.empty(native.lib().dvis_conv1x1_x3_packed_bytes(C + C2, Co), dtype=torch.uint8, device=wcat.device)
        with torch.cuda.device(wcat.device):
            native.check(native.lib().dvis_conv1x1_x3_pack(native.dev_ptr(wcat, "weight"), Co, C + C2, e, ctypes.c_void_p(buf.data_ptr()),
                                                           native.stream_ptr(wcat.device)), "dvis_conv1x1_x3_pack")
        bias = None
        if b3 is not None or bs is not None:
            bias = (0 if b3 is None else b3.detach()) + (0 if bs is None else bs.detach())
        return buf, e, bias
    buf, wexp, bias = _x3_cache(w3, (w3._version, w3.data_ptr(), ws._version, ws.data_ptr(), w3.device,
                                     None if b3 is None else b3._version, None if bs is None else bs._version), make,
                                kind="conv1x1 + shortcut")
    out = torch.empty((N, Co, H, W), dtype=torch.float32, device=a.device)
    step = min(_conv_x3_chunks(N, C, Co, H * W, H * W), _conv_x3_chunks(N, C2, Co, H2 * W2, H * W))
    with torch.cuda.device(a.device):
        for i in range(0, N, step):
            n = min(step, N - i)
            native.check(native.lib().dvis_conv1x1_x3_dual(
                native.dev_ptr(a[i:i + n], "a"), native.dev_ptr(x[i:i + n], "x"), ctypes.c_void_p(buf.data_ptr()),
                None if bias is None else native.dev_ptr(bias, "bias"), None, native.dev_ptr(out[i:i + n], "out"), n, C, C2, Co, H, W,
                H2, W2, stride2, X3_CONV_XEXP if xexp is None else xexp, wexp, 1 if relu else 0, native.stream_ptr(a.device)),
                "dvis_conv1x1_x3_dual")
    return out


def conv3x3_x3_ok(x, weight, stride=1, res=None):
    """Does the 3x3 / padding 1 form of csrc/conv1x1_x3.hip serve this convolution?"""
    return weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and conv1x1_x3_ok(x, weight, stride, res)


def conv3x3_x3(x, weight, bias=None, res=None, relu=False, stride=1, xexp=None):
    """relu?(conv2d(x, weight, padding=1, stride=stride) + bias[c] + res) through dvis_conv3x3_x3."""
    N, Ci, H, W = x.shape
    Co = weight.shape[0]

    def make():
        w2 = weight.detach().contiguous()
        e = _x3_exp(w2)
        buf = torch.empty(native.lib().dvis_conv3x3_x3_packed_bytes(Ci, Co), dtype=torch.uint8, device=w2.device)
        with torch.cuda.device(w2.device):
            native.check(native.lib().dvis_conv3x3_x3_pack(native.dev_ptr(w2, "weight"), Co, Ci, e, ctypes.c_void_p(buf.data_ptr()),
                                                           native.stream_ptr(w2.device)), "dvis_conv3x3_x3_pack")
        return buf, e
    buf, wexp = _x3_cache(weight, (weight._version, weight.data_ptr(), weight.device), make, kind="conv3x3")
    OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
    out = torch.empty((N, Co, OH, OW), dtype=torch.float32, device=x.device)
    step = _conv_x3_chunks(N, Ci, Co, H * W, OH * OW)
    with torch.cuda.device(x.device):
        for i in range(0, N, step):
            n = min(step, N - i)
            native.check(native.lib().dvis_conv3x3_x3(
                native.dev_ptr(x[i:i + n], "x"), ctypes.c_void_p(buf.data_ptr()),
                None if bias is None else native.dev_ptr(bias.detach(), "bias"),
                None if res is None else native.dev_ptr(res[i:i + n], "res"), native.dev_ptr(out[i:i + n], "out"), n, Ci, Co, H, W,
                stride, X3_CONV_XEXP if xexp is None else xexp, wexp, 1 if relu else 0, native.stream_ptr(x.device)),
                "dvis_conv3x3_x3")
    return out


X3_IMAGES = os.environ.get("DVIS_X3_IMAGES", "1") != "0"


class OperandImage:
    """A C-channel map as csrc/conv1x1_x3.hip's operand image (include/dvis_hip.h: dvis_conv_x3_image): per 32-pixel group C / 64
    chunks of 8 KB of pre-split f16 terms.  `exp`: the power of two the values were scaled with before the split."""

    def __init__(self, data, N, C, H, W, exp):
        self.data, self.N, self.C, self.H, self.W, self.exp = data, N, C, H, W, exp

    @property
    def device(self):
        return self.data.device

    def images(self, i, n):
        per = self.data.numel() // self.N
        return self.data[i * per:(i + n) * per]


def x3_images_ok(N, C, K, H, W, device, taps=1, stride=1):
    """Operand images between two launches of csrc/conv1x1_x3.hip: split-f16 kernels on, a GPU, channel counts the image forms serve."""
    if not (X3_IMAGES and x3_on() and device.type == "cuda" and not torch.is_grad_enabled()):
        return False
    if C % 64 != 0 or not (K == 128 or K % 256 == 0) or N <= 0:
        return False
    OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
    return bool(native.lib().dvis_conv1x1_x3_supported(C, K, 1, H * W, OH * OW))


def _image_chunks(N, C, K, H, W, OH, OW):
    """Images per launch: every tensor of the launch below 2 GiB — fp32 maps, and operand images (rows padded to 32 pixels)."""
    per = max(C * H * ((W + 31) // 32 * 32), K * OH * ((OW + 31) // 32 * 32)) * 4
    return max(1, min(N, (2 ** 31 - 1) // per))


def upsample_add_image(lateral, top, lat_affine=None, oexp=None):
    """`upsample_add` with the sum written as an operand image (the 3x3 output convolution behind it reads fragments)."""
    N, C, H, W = lateral.shape
    lib = native.lib()
    oe = X3_CONV_XEXP if oexp is None else oexp
    top = top.contiguous()
    data = torch.empty(lib.dvis_conv_x3_image_bytes(N, C, H, W), dtype=torch.uint8, device=lateral.device)
    img = OperandImage(data, N, C, H, W, oe)
    step = _image_chunks(N, C, C, H, W, H, W)
    with torch.cuda.device(lateral.device):
        for i in range(0, N, step):
            n = min(step, N - i)
            sc = None if lat_affine is None else native.dev_ptr(lat_affine[0].reshape(N, C)[i:i + n], "scale")
            sh = None if lat_affine is None else native.dev_ptr(lat_affine[1].reshape(N, C)[i:i + n], "shift")
            native.check(lib.dvis_upsample_add_image(native.dev_ptr(lateral[i:i + n], "lateral"), sc, sh, native.dev_ptr(top[i:i + n], "top"),
                                                     ctypes.c_void_p(img.images(i, n).data_ptr()), n, C, H, W, top.shape[-2], top.shape[-1], oe,
                                                     native.stream_ptr(lateral.device)), "dvis_upsample_add_image")
    return img


def conv_x3_image(src, weight, bias=None, res=None, relu=False, stride=1, out_image=False, oexp=None, xexp=None):
    """relu?(conv2d(src, weight, stride, padding = k // 2) + bias + res) through dvis_conv_x3_image: `src` an OperandImage or an
    (N, C, H, W) fp32 map (then out_image must be set), weight (K, C, 1, 1) or (K, C, 3, 3); returns an OperandImage (out_image) or
    the (N, K, OH, OW) fp32 map."""
    lib = native.lib()
    from_img = isinstance(src, OperandImage)
    if not (from_img or out_image):
        raise ValueError("conv_x3_image: neither side is an operand image")
    N, C, H, W = (src.N, src.C, src.H, src.W) if from_img else src.shape
    K, taps = weight.shape[0], weight.shape[2] * weight.shape[3]
    dev = weight.device

    def make():
        w2 = weight.detach().contiguous()
        e = _x3_exp(w2)
        nbytes = lib.dvis_conv1x1_x3_packed_bytes(C, K) * taps
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            if from_img:
                native.check(lib.dvis_conv_x3_pack_image(native.dev_ptr(w2, "weight"), K, C, taps, e, ctypes.c_void_p(buf.data_ptr()),
                                                         native.stream_ptr(dev)), "dvis_conv_x3_pack_image")
            elif taps == 1:
                native.check(lib.dvis_conv1x1_x3_pack(native.dev_ptr(w2.reshape(K, C), "weight"), K, C, e, ctypes.c_void_p(buf.data_ptr()),
                                                      native.stream_ptr(dev)), "dvis_conv1x1_x3_pack")
            else:
                native.check(lib.dvis_conv3x3_x3_pack(native.dev_ptr(w2, "weight"), K, C, e, ctypes.c_void_p(buf.data_ptr()),
                                                      native.stream_ptr(dev)), "dvis_conv3x3_x3_pack")
        return buf, e
    kind = ("conv%dx%d" % ((1, 1) if taps == 1 else (3, 3))) + (" image-in" if from_img else "")
    buf, wexp = _x3_cache(weight, (weight._version, weight.data_ptr(), weight.device), make, kind=kind)
    OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
    xe = src.exp if from_img else (X3_CONV_XEXP if xexp is None else xexp)
    oe = X3_CONV_XEXP if oexp is None else oexp
    if out_image:
        out = OperandImage(torch.empty(lib.dvis_conv_x3_image_bytes(N, K, OH, OW), dtype=torch.uint8, device=dev), N, K, OH, OW, oe)
    else:
        out = torch.empty((N, K, OH, OW), dtype=torch.float32, device=dev)
    step = _image_chunks(N, C, K, H, W, OH, OW)
    with torch.cuda.device(dev):
        for i in range(0, N, step):
            n = min(step, N - i)
            native.check(lib.dvis_conv_x3_image(
                ctypes.c_void_p(src.images(i, n).data_ptr()) if from_img else None, None if from_img else native.dev_ptr(src[i:i + n], "x"),
                ctypes.c_void_p(buf.data_ptr()), None if bias is None else native.dev_ptr(bias.detach(), "bias"),
                None if res is None else native.dev_ptr(res[i:i + n], "res"), None if out_image else native.dev_ptr(out[i:i + n], "out"),
                ctypes.c_void_p(out.images(i, n).data_ptr()) if out_image else None, n, C, K, H, W, stride, taps, xe, wexp, oe, 1 if relu else 0,
                native.stream_ptr(dev)), "dvis_conv_x3_image")
    return out


X3_BNECK = os.environ.get("DVIS_X3_BNECK", "1") != "0"


def bneck_stage_x3_ok(x, blocks):
    """Does csrc/bneck_x3.hip serve this stage?  `blocks`: per bottleneck a dict of FOLDED weights / shifts
    {w1, b1, w2, b2, w3, b3, ws, bs} (ws / bs None for an identity shortcut) — the res2 stage of the R50: 64 -> (64, 3x3 64, 256),
    stride 1, a projection shortcut in the first block only."""
    if not (X3_BNECK and x3_on() and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()
            and not torch.is_grad_enabled() and len(blocks) >= 2 and x.shape[1] == 64 and x.shape[0] > 0):
        return False
    for i, b in enumerate(blocks):
        cin = 64 if i == 0 else 256
        if tuple(b["w1"].shape) != (64, cin, 1, 1) or tuple(b["w2"].shape) != (64, 64, 3, 3) or tuple(b["w3"].shape) != (256, 64, 1, 1):
            return False
        if (b["ws"] is not None) != (i == 0) or (i == 0 and tuple(b["ws"].shape) != (256, 64, 1, 1)):
            return False
        if any(b[k] is not None and (b[k].dtype != torch.float32 or b[k].device != x.device) for k in b):
            return False
    return bool(native.lib().dvis_bneck_x3_supported(1, x.shape[2], x.shape[3]))


def bneck_stage_x3(x, blocks, xexp=None):
    """The whole stage through dvis_conv1x1_x3_image (first conv1) + one dvis_bneck_x3 per block (conv2 -> conv3 + shortcut ->
    the next block's conv1); the 64-channel maps between the launches are operand images.  Returns the stage's output."""
    N, _, H, W = x.shape
    lib = native.lib()
    xe = X3_CONV_XEXP if xexp is None else xexp
    dev = x.device

    def packed_conv1():
        w = blocks[0]["w1"]

        def make():
            w2d = w.detach().reshape(64, 64).contiguous()
            e = _x3_exp(w2d)
            buf = torch.empty(lib.dvis_conv1x1_x3_packed_bytes(64, 64), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                native.check(lib.dvis_conv1x1_x3_pack(native.dev_ptr(w2d, "weight"), 64, 64, e, ctypes.c_void_p(buf.data_ptr()),
                                                      native.stream_ptr(dev)), "dvis_conv1x1_x3_pack")
            return buf, e
        return _x3_cache(w, (w._version, w.data_ptr(), w.device), make, kind="conv1x1")

    def packed_block(i):
        b, nxt = blocks[i], (blocks[i + 1] if i + 1 < len(blocks) else None)
        ws, w1n = b["ws"], (None if nxt is None else nxt["w1"])

        def make():
            w2 = b["w2"].detach().contiguous()
            w3 = b["w3"].detach().reshape(256, 64).contiguous()
            wsd = None if ws is None else ws.detach().reshape(256, 64).contiguous()
            w1d = None if w1n is None else w1n.detach().reshape(64, 256).contiguous()
            e2, e1 = _x3_exp(w2), (0 if w1d is None else _x3_exp(w1d))
            e3 = _x3_exp(w3 if wsd is None else torch.cat([w3, wsd], 1))
            buf = torch.empty(lib.dvis_bneck_x3_packed_bytes(0 if w1d is None else 1, 0 if wsd is None else 1), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                native.check(lib.dvis_bneck_x3_pack(native.dev_ptr(w2, "w2"), native.dev_ptr(w3, "w3"),
                                                    None if wsd is None else native.dev_ptr(wsd, "ws"),
                                                    None if w1d is None else native.dev_ptr(w1d, "w1"), e2, e3, e1,
                                                    ctypes.c_void_p(buf.data_ptr()), native.stream_ptr(dev)), "dvis_bneck_x3_pack")
            b3 = b["b3"]
            if b["bs"] is not None:
                b3 = b["bs"].detach() if b3 is None else b3.detach() + b["bs"].detach()
            return buf, e2, e3, e1, (None if b3 is None else b3.detach().contiguous())
        key = tuple((t._version, t.data_ptr()) if t is not None else None for t in (b["w2"], b["w3"], ws, w1n, b["b3"], b["bs"])) + (dev,)
        return _x3_cache(b["w2"], key, make, kind="bottleneck chain")

    img_bytes = lib.dvis_bneck_x3_image_bytes(N, H, W)
    imgs = [torch.empty(img_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
    ys = [torch.empty((N, 256, H, W), dtype=torch.float32, device=dev) for _ in range(min(2, len(blocks)))]
    step = max(1, min(N, (2 ** 31 - 1) // (256 * H * W * 4)))       # images per launch: every tensor below 2 GiB
    per_img = img_bytes // N

    def ptr(t):
        return None if t is None else native.dev_ptr(t.detach(), "shift")

    with torch.cuda.device(dev):
        for i0 in range(0, N, step):
            n = min(step, N - i0)
            isl = slice(i0 * per_img, (i0 + n) * per_img)
            buf1, e1 = packed_conv1()
            native.check(lib.dvis_conv1x1_x3_image(native.dev_ptr(x[i0:i0 + n], "x"), ctypes.c_void_p(buf1.data_ptr()), ptr(blocks[0]["b1"]),
                                                   ctypes.c_void_p(imgs[0][isl].data_ptr()), n, 64, H, W, xe, e1, xe, 1, native.stream_ptr(dev)),
                         "dvis_conv1x1_x3_image")
            for i, b in enumerate(blocks):
                buf, e2, e3, e1n, b3 = packed_block(i)
                last = i + 1 == len(blocks)
                y, yprev = ys[i % 2], ys[(i + 1) % 2]
                native.check(lib.dvis_bneck_x3(
                    ctypes.c_void_p(imgs[i % 2][isl].data_ptr()), None if i == 0 else native.dev_ptr(yprev[i0:i0 + n], "res"),
                    native.dev_ptr(x[i0:i0 + n], "x") if i == 0 else None, ctypes.c_void_p(buf.data_ptr()), ptr(b["b2"]), ptr(b3),
                    None if last else ptr(blocks[i + 1]["b1"]), native.dev_ptr(y[i0:i0 + n], "y"),
                    None if last else ctypes.c_void_p(imgs[(i + 1) % 2][isl].data_ptr()), n, H, W, xe, e2, e3, e1n, native.stream_ptr(dev)),
                    "dvis_bneck_x3")
    return ys[(len(blocks) - 1) % 2]


CONV1X1_MFMA = os.environ.get("DVIS_CONV1X1_MFMA", "1") != "0"
_C1_PACKED = {}


def conv1x1s2_supported(x, weight):
    """Is the stride-2 1x1 convolution of x (the shortcut of a down-sampling bottleneck) served by csrc/conv1x1_mfma.hip?"""
    N, Ci, H, W = x.shape
    return CONV1X1_MFMA and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not torch.is_grad_enabled() \
        and bool(native.lib().dvis_conv1x1_mfma_supported(Ci, weight.shape[0], ((H + 1) // 2) * ((W + 1) // 2))) \
        and 2 * Ci * H * W * 4 < 2 ** 31


def conv1x1_mfma(x, weight, bias=None, res=None, relu=False, stride=1):
    """relu?(conv1x1(x, weight) + bias[c] + res) for the compute-bound 1x1 layers (csrc/conv1x1_mfma.hip: C % 128 == 0, K % 64 == 0):
    contraction, folded-BN shift, shortcut add and ReLU in one kernel instead of the library's batched GEMM + ``bias_act_``.
    stride=2: the same on x[:, :, ::2, ::2] (read in place)."""
    N, Ci, H, W = x.shape
    Co = weight.shape[0]
    key = (weight._version, weight.data_ptr(), weight.device)
    ent = _C1_PACKED.get(id(weight))
    if ent is None or ent[0] != key:
        w2 = weight.detach().reshape(Co, Ci).contiguous()
        uf = ent[1] if ent is not None and ent[1].device == w2.device and ent[1].numel() == Co * Ci else torch.empty_like(w2)
        with torch.cuda.device(w2.device):
            native.check(native.lib().dvis_conv1x1_mfma_pack(native.dev_ptr(w2, "weight"), native.dev_ptr(uf, "uf"), Co, Ci,
                                                             native.stream_ptr(w2.device)), "dvis_conv1x1_mfma_pack")
        if len(_C1_PACKED) > 512:
            _C1_PACKED.clear()
        _C1_PACKED[id(weight)] = ent = (key, uf, weight)
    if stride == 2:
        out = torch.empty((N, Co, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = native.lib().dvis_conv1x1s2_mfma(
                native.dev_ptr(x, "x"), native.dev_ptr(ent[1], "uf"), None if bias is None else native.dev_ptr(bias.detach(), "bias"),
                None if res is None else native.dev_ptr(res, "res"), native.dev_ptr(out, "out"), N, Ci, Co, H, W, 1 if relu else 0,
                native.stream_ptr(x.device))
        native.check(rc, "dvis_conv1x1s2_mfma")
        return out
    out = torch.empty((N, Co, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = native.lib().dvis_conv1x1_mfma(
            native.dev_ptr(x, "x"), native.dev_ptr(ent[1], "uf"), None if bias is None else native.dev_ptr(bias.detach(), "bias"),
            None if res is None else native.dev_ptr(res, "res"), native.dev_ptr(out, "out"), N, Ci, Co, H * W, 1 if relu else 0,
            native.stream_ptr(x.device))
    native.check(rc, "dvis_conv1x1_mfma")
    return out


WINOGRAD_DEFAULT = os.environ.get("DVIS_WINOGRAD", "1") != "0"
_WINOGRAD_PACKED = {}     # id(weight) -> ((version, data_ptr, device), packed): transformed weights, made once per weight


def _winograd_weights(weight):
    key = (weight._version, weight.data_ptr(), weight.device)
    ent = _WINOGRAD_PACKED.get(id(weight))
    if ent is None or ent[0] != key:
        K, C = weight.shape[:2]
        w = weight.detach().contiguous()
        uf = ent[1] if ent is not None and ent[1].device == w.device and ent[1].numel() == 16 * K * C else \
            torch.empty(16 * K * C, dtype=torch.float32, device=w.device)       # refreshed in place (captured graphs)
        with torch.cuda.device(w.device):
            native.check(native.lib().dvis_conv3x3_winograd_pack(native.dev_ptr(w, "weight"), native.dev_ptr(uf, "uf"), K, C,
                                                                 native.stream_ptr(w.device)), "dvis_conv3x3_winograd_pack")
        if len(_WINOGRAD_PACKED) > 512:       # (weights that were replaced, e.g. re-folded FrozenBN: drop their packs)
            _WINOGRAD_PACKED.clear()
        _WINOGRAD_PACKED[id(weight)] = ent = (key, uf, weight)     # (holds the weight: id() stays unique)
    return ent[1]


_S2_PACKED = {}
S2_DEFAULT = os.environ.get("DVIS_CONV3X3S2", "1") != "0"


def _s2_weights(weight):
    key = (weight._version, weight.data_ptr(), weight.device)
    ent = _S2_PACKED.get(id(weight))
    if ent is None or ent[0] != key:
        K, C = weight.shape[:2]
        w = weight.detach().contiguous()
        uf = ent[1] if ent is not None and ent[1].device == w.device and ent[1].numel() == 12 * K * C else \
            torch.empty(12 * K * C, dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            native.check(native.lib().dvis_conv3x3s2_pack(native.dev_ptr(w, "weight"), native.dev_ptr(uf, "uf"), K, C,
                                                          native.stream_ptr(w.device)), "dvis_conv3x3s2_pack")
        if len(_S2_PACKED) > 512:
            _S2_PACKED.clear()
        _S2_PACKED[id(weight)] = ent = (key, uf, weight)
    return ent[1]


def conv3x3s2_bias_act(x, weight, bias=None, relu=False, own=None):
    """relu?(conv2d(x, weight (K, C, 3, 3), stride 2, padding 1) + bias[k]) on NCHW.  own=True: the direct fp32-MFMA kernel
    (csrc/conv3x3s2.hip: 119 - 131 TFLOP/s at the R50 shapes, 1.4 - 1.65x the library + epilogue pass), raising when the
    shape is not served; own=None: that kernel where the shape is served; otherwise the library convolution followed by the
    in-place ``bias_act_`` pass."""
    N, C, H, W = x.shape
    K = weight.shape[0]
    if own is None and conv3x3_x3_ok(x, weight, 2):
        return conv3x3_x3(x, weight, bias, None, relu, 2)      # nine-tap split-f16 matrix-core kernel: 1.6 - 2.2x the one below
    ok = x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and tuple(weight.shape[1:]) == (C, 3, 3) \
        and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)) \
        and bool(native.lib().dvis_conv3x3s2_supported(C, K, H, W))
    if own and not ok:
        raise RuntimeError(f"conv3x3s2_bias_act(own=True): C={C} K={K} H={H} W={W} is not served (dvis_conv3x3s2_supported)")
    if ok and (own or (own is None and WINOGRAD_DEFAULT and S2_DEFAULT)):
        x = x if x.is_contiguous() else x.contiguous()
        uf = _s2_weights(weight)
        out = torch.empty((N, K, (H + 1) // 2, W // 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = native.lib().dvis_conv3x3s2(
                native.dev_ptr(x, "x"), native.dev_ptr(uf, "uf"), None if bias is None else native.dev_ptr(bias.detach(), "bias"),
                native.dev_ptr(out, "out"), N, C, K, H, W, 1 if relu else 0, native.stream_ptr(x.device))
        native.check(rc, "dvis_conv3x3s2")
        return out
    y = torch.nn.functional.conv2d(x, weight, None, 2, 1)
    if bias is None and not relu:
        return y
    return bias_act_(y, None if bias is None else bias.detach(), None, relu)


_STEM_PACKED = {}


def conv7x7s2_stem(x, weight, bias=None, relu=False, own=None):
    """relu?(conv2d(x (N, 3, H, W), weight (64, 3, 7, 7), stride 2, padding 3) + bias[k]): the ResNet stem convolution on the
    direct fp32-MFMA kernel (csrc/conv7x7s2.hip) where the shape is served (own=True: raise otherwise), else the library."""
    N, C, H, W = x.shape
    K = weight.shape[0]
    ok = x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and tuple(weight.shape) == (64, 3, 7, 7) \
        and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)) \
        and bool(native.lib().dvis_conv7x7s2_supported(C, K, H, W))
    if own and not ok:
        raise RuntimeError(f"conv7x7s2_stem(own=True): C={C} K={K} H={H} W={W} is not served (dvis_conv7x7s2_supported)")
    if ok and (own or (own is None and WINOGRAD_DEFAULT and S2_DEFAULT)):
        key = (weight._version, weight.data_ptr(), weight.device)
        ent = _STEM_PACKED.get(id(weight))
        if ent is None or ent[0] != key:
            w = weight.detach().contiguous()
            uf = ent[1] if ent is not None and ent[1].device == w.device else torch.empty(10240, dtype=torch.float32, device=w.device)
            with torch.cuda.device(w.device):
                native.check(native.lib().dvis_conv7x7s2_pack(native.dev_ptr(w, "weight"), native.dev_ptr(uf, "uf"),
                                                              native.stream_ptr(w.device)), "dvis_conv7x7s2_pack")
            if len(_STEM_PACKED) > 64:
                _STEM_PACKED.clear()
            _STEM_PACKED[id(weight)] = ent = (key, uf, weight)
        x = x if x.is_contiguous() else x.contiguous()
        out = torch.empty((N, 64, H // 2, W // 2), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = native.lib().dvis_conv7x7s2(native.dev_ptr(x, "x"), native.dev_ptr(ent[1], "uf"),
                                             None if bias is None else native.dev_ptr(bias.detach(), "bias"),
                                             native.dev_ptr(out, "out"), N, H, W, 1 if relu else 0, native.stream_ptr(x.device))
        native.check(rc, "dvis_conv7x7s2")
        return out
    y = torch.nn.functional.conv2d(x, weight, None, 2, 3)
    if bias is None and not relu:
        return y
    return bias_act_(y, None if bias is None else bias.detach(), None, relu)


def conv3x3_bias_act(x, weight, bias=None, relu=False, winograd=None):
    """relu?(conv2d(x, weight (K, C, 3, 3), stride 1, padding 1) + bias[k]) on NCHW.  Shapes dvis_conv3x3_winograd serves run
    as ONE own kernel — Winograd F(2x2, 3x3) on the fp32 matrix cores, bias / ReLU in its epilogue (csrc/winograd_conv.hip:
    the FPN output convolution and conv2 of the R50 bottlenecks); other shapes, and `winograd=False` / DVIS_WINOGRAD=0, are
    the library convolution followed by the in-place ``bias_act_`` pass."""
    if winograd is None and weight.shape[1] >= 128 and conv3x3_x3_ok(x, weight, 1):
        # nine taps on the f16 matrix cores with split operands: 27 / 16 of an fp32 matrix-core product per output, the fp32
        # Winograd kernel below 4 — 1.2 - 1.5x faster from 128 input channels on (64: the Winograd kernel wins)
        return conv3x3_x3(x, weight, bias, None, relu, 1)
    use = WINOGRAD_DEFAULT if winograd is None else winograd
    if use and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and weight.dtype == torch.float32 \
            and tuple(weight.shape[2:]) == (3, 3) and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)):
        N, C, H, W = x.shape
        K = weight.shape[0]
        served = weight.shape[1] == C and native.lib().dvis_conv3x3_winograd_supported(C, K, H, W)
        if winograd and not served:
            raise RuntimeError(f"conv3x3_bias_act(winograd=True): shape C={C} K={K} H={H} W={W} is not served "
                               "(dvis_conv3x3_winograd_supported)")
        if served:
            x = x if x.is_contiguous() else x.contiguous()
            uf = _winograd_weights(weight)
            out = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                rc = native.lib().dvis_conv3x3_winograd(
                    native.dev_ptr(x, "x"), native.dev_ptr(uf, "uf"),
                    None if bias is None else native.dev_ptr(bias.detach(), "bias"), native.dev_ptr(out, "out"), N, C, K, H, W,
                    1 if relu else 0, native.stream_ptr(x.device))
            native.check(rc, "dvis_conv3x3_winograd")
            return out
    y = torch.nn.functional.conv2d(x, weight, None, 1, 1)
    if bias is None and not relu:
        return y
    return bias_act_(y, None if bias is None else bias.detach(), None, relu)


def bias_act_(x, bias=None, res=None, relu=True):
    """In place: x = relu?(x + bias[c] + res) on an NCHW float32 tensor — one pass instead of torch's three kernels
    (conv bias add, residual add, ReLU).  Non-GPU / odd shapes use the torch ops."""
    N, C, H, W = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
            and (res is None or (res.is_contiguous() and res.shape == x.shape and res.dtype == torch.float32))
            and not torch.is_grad_enabled()):
        _torch_path("bias_act_", x, "needs contiguous fp32 NCHW (and a matching residual)")
        if bias is not None:
            x = x + bias.view(1, -1, 1, 1)
        if res is not None:
            x = x + res
        return torch.relu_(x) if relu else x
    with torch.cuda.device(x.device):
        rc = native.lib().dvis_bias_act(native.dev_ptr(x, "x"), None if bias is None else native.dev_ptr(bias, "bias"),
                                        None if res is None else native.dev_ptr(res, "res"), N * C, C, H * W,
                                        1 if relu else 0, native.stream_ptr(x.device))
    native.check(rc, "dvis_bias_act")
    return x
