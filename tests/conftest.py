import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are skipped automatically when no GPU is visible (never silently passed).  The product is strict by
    default (dvis_plus_amd.functions._torch_path: a fused glue op that would quietly take its torch formulation on a GPU
    tensor raises unless DVIS_STRICT=0), so every GPU test proves the HIP path is the one that ran — nothing to set here."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """npz fixture written by tests/golden/gen_golden.py: in/…, out/…, sd/… arrays."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def _sub(self, prefix, as_torch=True):
        d = {}
        for k in self.z.files:
            if k.startswith(prefix + "/"):
                a = self.z[k]
                d[k[len(prefix) + 1:]] = torch.from_numpy(a.copy()) if as_torch else a
        return d

    @property
    def ins(self):
        return self._sub("in")

    @property
    def outs(self):
        return self._sub("out")

    @property
    def sd(self):
        return self._sub("sd")

    @property
    def meta(self):
        return eval(str(self.z["meta"]))  # repr() of a plain dict written by gen_golden.py


@pytest.fixture
def golden():
    return Golden


def level_tensors(shapes, device="cpu"):
    s = torch.as_tensor(shapes, dtype=torch.long, device=device)
    lsi = torch.cat((s.new_zeros((1,)), s.prod(1).cumsum(0)[:-1]))
    return s, lsi


def make_msda_inputs(N, M, D, shapes, Lq, P, dtype=torch.float32, seed=0, spread=1.4):
    """Seeded random op inputs; locations cover [-0.2, 1.2] so borders / outside samples occur."""
    g = torch.Generator().manual_seed(seed)
    s, lsi = level_tensors(shapes)
    L = len(shapes)
    S = int(s.prod(1).sum())
    value = torch.randn(N, S, M, D, generator=g, dtype=torch.float64).to(dtype)
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=g, dtype=torch.float64) * spread - (spread - 1) / 2).to(dtype)
    w = torch.rand(N, Lq, M, L, P, generator=g, dtype=torch.float64)
    w = (w / w.flatten(-2).sum(-1)[..., None, None]).to(dtype)
    return value, s, lsi, loc, w


# ---------------------------------------------------------------------------------------------------------------
# CPU stand-ins for the HIP ops, built from the ORACLE — test infrastructure only.  They let the `-m "not gpu"`
# tests exercise the product's HOST logic (module wiring, state_dict keys, batching, clip sharding) on CPU
# tensors.  The product itself has no such path: without this fixture the ops raise on CPU tensors.
# ---------------------------------------------------------------------------------------------------------------
def _o_attention(q, k, v, nheads, mask=None, allowed_count=None, out=None, short=False):
    Lq, B, C = q.shape
    Lk, d = k.shape[0], C // nheads
    qh = q.reshape(Lq, B, nheads, d).permute(1, 2, 0, 3)
    kh = k.reshape(Lk, B, nheads, d).permute(1, 2, 0, 3)
    vh = v.reshape(Lk, B, nheads, d).permute(1, 2, 0, 3)
    s = (qh * (1.0 / d ** 0.5)) @ kh.transpose(-1, -2)
    if mask is not None:
        m = mask.bool().clone()
        if allowed_count is not None:
            m[allowed_count == 0] = False
        s = s.masked_fill(m[:, None], float("-inf"))
    o = (torch.softmax(s, -1) @ vh).permute(2, 0, 1, 3).reshape(Lq, B, C)
    if out is not None:
        out.copy_(o)
        return out
    return o


def _o_attn_mask(mask_embed, mask_features, target_size):
    import torch.nn.functional as F
    logits = torch.einsum("bqc,bchw->bqhw", mask_embed, mask_features)
    small = F.interpolate(logits, size=tuple(target_size), mode="bilinear", align_corners=False)
    mask = (small.sigmoid().flatten(2) < 0.5)
    return mask.to(torch.uint8), (~mask).sum(-1).to(torch.int32)


def _o_mask_logits(mask_embed, mask_features):
    return torch.einsum("bqc,bchw->bqhw", mask_embed, mask_features)


def _o_msda_fused(value, spatial_shapes, level_start_index, reference_points, offsets, logits, n_levels, n_points,
                  shapes_host=None, pos_offsets=None, pos_logits=None, head_stride=0, value_head_major=False):
    from oracle.msda import msda_forward_torch
    if value_head_major:
        value = value.permute(1, 2, 0, 3).contiguous()
    N, S, M, D = value.shape
    Lq = reference_points.shape[1]
    L, P = n_levels, n_points

    def heads(rows, width):          # (rows, >= ...) -> (rows, M, width): head m's run starts m * head_stride floats in
        if not head_stride:
            return rows[:, :M * width].reshape(rows.shape[0], M, width)
        return torch.stack([rows[:, m * head_stride:m * head_stride + width] for m in range(M)], 1)
    off = heads(offsets, L * P * 2).reshape(N, Lq, M, L, P, 2)
    lg = heads(logits, L * P).reshape(N, Lq, M, L * P)
    if pos_offsets is not None:
        off = off + heads(pos_offsets, L * P * 2).reshape(1, Lq, M, L, P, 2)
        lg = lg + heads(pos_logits, L * P).reshape(1, Lq, M, L * P)
    w = torch.softmax(lg, -1).reshape(N, Lq, M, L, P)
    norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
    loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    return msda_forward_torch(value, spatial_shapes, loc.expand(N, -1, -1, -1, -1, -1), w)


class _OMSDAFunction:
    @staticmethod
    def apply(value, shapes, level_start, loc, w, im2col_step):
        from oracle.msda import msda_forward_torch
        return msda_forward_torch(value, shapes, loc, w)


@pytest.fixture
def oracle_ops(monkeypatch):
    from dvis_plus_amd import functions as Fn
    monkeypatch.setattr(Fn, "attention", _o_attention)
    monkeypatch.setattr(Fn, "attn_mask", _o_attn_mask)
    monkeypatch.setattr(Fn, "mask_logits", _o_mask_logits)
    monkeypatch.setattr(Fn, "msda_fused_forward", _o_msda_fused)
    monkeypatch.setattr(Fn, "MSDeformAttnFunction", _OMSDAFunction)
    return Fn
