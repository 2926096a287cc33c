"""Comparison of integer / boolean outputs (BASELINE.json: "bit-exact for index/argmax assignment").

Two regimes:
  * same floats in, decision only (post-processing kernels fed the same logits as torch): `exact()` — torch.equal, the
    differing-pixel count is reported even when it is zero;
  * floats that come out of different fp32 summation orders (GPU pipeline vs CPU oracle, logits agree to ~1e-5): a pixel
    may legitimately flip only if the ORACLE's own value is within `tol` of the decision boundary.  `near_boundary()`
    counts the differing pixels, measures how far from the boundary the worst one is (fp64 where the caller has it),
    asserts that distance <= tol and bounds the COUNT by the number of oracle pixels that close to the boundary — a
    count, not a fraction.
Every call appends one line to $DVIS_PARITY_REPORT (when set) and prints it, so the numbers are on record.
"""
import os

import torch


def _report(line):
    print("[intcmp] " + line)
    path = os.environ.get("DVIS_PARITY_REPORT")
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "a") as f:
            f.write(line + "\n")


def exact(got, want, what):
    got, want = got.cpu(), want.cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    n = int((got != want).sum())
    _report(f"{what}: {n} of {got.numel()} elements differ (exact comparison)")
    assert n == 0, f"{what}: {n} of {got.numel()} elements differ"


def near_boundary(got, want, distance, tol, what, max_count=None):
    """got / want: integer or bool tensors; distance: the oracle's |value - decision boundary| per element (any float
    dtype, same shape): a differing element must have distance <= tol."""
    got, want, distance = got.cpu(), want.cpu(), distance.cpu()
    assert got.shape == want.shape == distance.shape, f"{what}: shapes {got.shape} {want.shape} {distance.shape}"
    diff = got != want
    n = int(diff.sum())
    n_close = int((distance <= tol).sum())
    worst = float(distance[diff].max()) if n else 0.0
    _report(f"{what}: {n} of {got.numel()} elements differ; largest oracle distance to the decision boundary among them "
            f"{worst:.3e} (tol {tol:.1e}); the oracle has {n_close} elements within tol")
    assert worst <= tol, f"{what}: an element {worst:.3e} away from the decision boundary differs (tol {tol})"
    if max_count is not None:
        assert n <= max_count, f"{what}: {n} differing elements > bound {max_count}"
    return n


def argmax_margin(values, dim=0):
    """Distance of an arg-max decision to its boundary: top-1 minus top-2 along `dim` (inf for a single candidate)."""
    if values.shape[dim] < 2:
        return torch.full_like(values.select(dim, 0), float("inf"))
    top2 = values.topk(2, dim=dim)[0]
    return top2.select(dim, 0) - top2.select(dim, 1)
