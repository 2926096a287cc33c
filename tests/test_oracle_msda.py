"""CPU: the oracle (plain C + torch restatement) against the golden vectors from the imported reference."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, Golden, make_msda_inputs
from oracle import msda as omsda

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g1_msda_*.npz")))


def _tol(dtype):
    return dict(rtol=1e-9, atol=1e-12) if dtype == np.float64 else dict(rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_forward_matches_golden(name):
    g = Golden(name)
    i, o = g._sub("in", False), g._sub("out", False)
    out = omsda.msda_forward(i["value"], i["shapes"], i["level_start"], i["loc"], i["w"])
    assert out.dtype == o["out"].dtype
    np.testing.assert_allclose(out, o["out"], **_tol(out.dtype))


@pytest.mark.parametrize("name", [c for c in CASES if "reftest" not in c])
def test_c_oracle_backward_matches_golden(name):
    g = Golden(name)
    i, o = g._sub("in", False), g._sub("out", False)
    gv, gl, gw = omsda.msda_backward(i["value"], i["shapes"], i["level_start"], i["loc"], i["w"], i["grad_out"])
    tol = _tol(gv.dtype)
    if gv.dtype == np.float32:   # grad_loc carries a factor W_l / H_l and a sum over D channels
        tol = dict(rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(gv, o["grad_value"], **tol)
    np.testing.assert_allclose(gl, o["grad_loc"], **tol)
    np.testing.assert_allclose(gw, o["grad_w"], **tol)


@pytest.mark.parametrize("name", CASES)
def test_torch_restatement_matches_golden(name):
    g = Golden(name)
    i, o = g.ins, g.outs
    out = omsda.msda_forward_torch(i["value"], i["shapes"], i["loc"], i["w"])
    assert torch.equal(out, o["out"])  # same ops, same order as the reference's CPU path -> bit-exact


def test_reference_tolerances_of_its_own_test():
    """ops/test.py:43,59: fp64 allclose defaults, fp32 rtol=1e-2 atol=1e-3 — the oracle is far inside both."""
    for tag, kw in (("f64", {}), ("f32", dict(rtol=1e-2, atol=1e-3))):
        g = Golden(f"g1_msda_reftest_{tag}")
        i, o = g._sub("in", False), g._sub("out", False)
        out = omsda.msda_forward(i["value"], i["shapes"], i["level_start"], i["loc"], i["w"])
        assert np.allclose(out, o["out"], **kw)


def test_c_vs_torch_random_and_edges():
    for seed, (N, M, D, shapes, Lq, P) in enumerate([(2, 3, 5, [(3, 4), (1, 1)], 11, 3), (1, 1, 1, [(1, 7)], 4, 1),
                                                     (3, 8, 32, [(5, 6), (9, 11), (2, 2)], 70, 4)]):
        value, s, lsi, loc, w = make_msda_inputs(N, M, D, shapes, Lq, P, torch.float64, seed, spread=2.0)
        a = omsda.msda_forward(value, s, lsi, loc, w)
        b = omsda.msda_forward_torch(value, s, loc, w).numpy()
        np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12)


def test_c_oracle_empty_and_outside():
    value, s, lsi, loc, w = make_msda_inputs(1, 2, 4, [(3, 3)], 5, 2, torch.float32, 1)
    loc[:] = 5.0                                 # everything outside -> exact zeros
    assert np.count_nonzero(omsda.msda_forward(value, s, lsi, loc, w)) == 0
    out = omsda.msda_forward(value[:0], s, lsi, loc[:0], w[:0])
    assert out.shape == (0, 5, 8)
