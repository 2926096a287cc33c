"""ORACLE (test infrastructure): ctypes front-end of ``msda_oracle.c`` + a torch fp restatement.

``msda_forward`` / ``msda_backward`` run the plain-C loops (fp32 or fp64) on numpy/torch CPU
arrays.  ``msda_forward_torch`` restates the reference's own CPU formulation
(ops/functions/ms_deform_attn_func.py:52-72: per-level ``F.grid_sample`` on ``2*loc-1`` with
bilinear / zeros / align_corners=False, then the weighted sum) and is differentiable; the two
are cross-checked against each other and against tests/golden/g1_msda_*.npz.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmsda_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libmsda_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _as_np(x, dt):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=dt)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _dims(value, loc):
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    return N, S, M, D, L, Lq, P


def msda_forward(value, shapes, level_start, loc, w):
    """C oracle forward.  Returns a numpy array (N, Lq, M*D) of value's dtype (float32/float64)."""
    dt = np.float64 if _as_np(value, None).dtype == np.float64 else np.float32
    value, loc, w = _as_np(value, dt), _as_np(loc, dt), _as_np(w, dt)
    shapes, level_start = _as_np(shapes, np.int64), _as_np(level_start, np.int64)
    N, S, M, D, L, Lq, P = _dims(value, loc)
    out = np.empty((N, Lq, M * D), dtype=dt)
    fn = lib().msda_oracle_forward_f64 if dt == np.float64 else lib().msda_oracle_forward_f32
    fn(_ptr(value), _ptr(shapes), _ptr(level_start), _ptr(loc), _ptr(w),
       *(ctypes.c_int(v) for v in (N, S, M, D, L, Lq, P)), _ptr(out))
    return out


def msda_backward(value, shapes, level_start, loc, w, grad_out):
    dt = np.float64 if _as_np(value, None).dtype == np.float64 else np.float32
    value, loc, w, grad_out = (_as_np(a, dt) for a in (value, loc, w, grad_out))
    shapes, level_start = _as_np(shapes, np.int64), _as_np(level_start, np.int64)
    N, S, M, D, L, Lq, P = _dims(value, loc)
    gv, gl, gw = np.zeros_like(value), np.empty_like(loc), np.empty_like(w)
    fn = lib().msda_oracle_backward_f64 if dt == np.float64 else lib().msda_oracle_backward_f32
    fn(_ptr(value), _ptr(shapes), _ptr(level_start), _ptr(loc), _ptr(w), _ptr(grad_out),
       *(ctypes.c_int(v) for v in (N, S, M, D, L, Lq, P)), _ptr(gv), _ptr(gl), _ptr(gw))
    return gv, gl, gw


def msda_forward_torch(value, shapes, loc, w):
    """Differentiable torch restatement (grid_sample formulation), any float dtype, CPU."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    hw = [(int(h), int(wd)) for h, wd in shapes.tolist()]
    per_level = value.split([h * wd for h, wd in hw], dim=1)
    grid = 2 * loc - 1
    sampled = []
    for lvl, (h, wd) in enumerate(hw):
        v = per_level[lvl].permute(0, 2, 3, 1).reshape(N * M, D, h, wd)          # (N*M, D, H, W)
        g = grid[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(N * M, Lq, P, 2)   # (N*M, Lq, P, 2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    samp = torch.stack(sampled, dim=-2).flatten(-2)                                # (N*M, D, Lq, L*P)
    aw = w.permute(0, 2, 1, 3, 4).reshape(N * M, 1, Lq, L * P)
    out = (samp * aw).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()
