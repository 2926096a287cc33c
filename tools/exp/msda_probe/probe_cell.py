"""Cell kernel (msda_cell.hip) vs the tile kernel of the product on the probe's inputs: max |difference| and time."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.argv = sys.argv[:1]
import probe as P   # builds the inputs, runs the default experiments

import torch
lib = P.lib
lib.msda_cell.restype = ctypes.c_int
out2 = torch.full_like(P.out, float("nan"))
H2, W2 = P.shapes_py[2]


def run_cell(exp=0):
    rc = lib.msda_cell(P.p(P.value), P.p(P.shapes), P.p(P.lsi), P.p(P.ref), 1, P.p(P.off), ctypes.c_int64(P.off.stride(0)),
                       P.p(P.lg), ctypes.c_int64(P.lg.stride(0)), P.N, P.S, P.M, P.Lq, H2, W2, P.p(out2), P.st, exp)
    assert rc == 0, rc


P.run(0)
run_cell()
torch.cuda.synchronize()
d = (out2 - P.out).abs()
print(f"cell vs tile: max |d| = {float(d.max()):.3e}  (nan: {int(torch.isnan(out2).sum())}), max |out| {float(P.out.abs().max()):.2f}")
names = {0: "full", 1: "no box fill", 2: "no LDS corner reads", 3: "no fill, no LDS reads", 4: "no global gathers (levels 0, 1)",
         6: "no LDS reads, no global gathers", 7: "set-up + stores only"}
for exp in (0, 1, 2, 3, 4, 6, 7):
    for _ in range(3):
        run_cell(exp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run_cell(exp)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"cell kernel, {names[exp]:34s}: {us:8.1f} us/launch = {us / P.N:6.2f} us/frame-layer")
