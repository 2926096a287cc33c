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
    """`-m gpu` tests are skipped automatically when no GPU is visible (never silently passed)."""
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
