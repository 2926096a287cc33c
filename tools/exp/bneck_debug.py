"""Where does the bottleneck chain differ from the fp64 reference?  (development aid for csrc/bneck_x3.hip)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_bneck_x3_gpu as T      # noqa: E402
from dvis_plus_amd import functions as Fn      # noqa: E402


def report(tag, got, ref):
    bad = (got.double() != ref)
    n = int(bad.sum())
    print(f"{tag}: {n} of {bad.numel()} differ")
    if n == 0:
        return
    idx = bad.nonzero()
    for d, name in enumerate(("n", "c", "y", "x")):
        vals, cnt = idx[:, d].unique(return_counts=True)
        print(f"   {name}: {len(vals)} distinct; first {vals[:16].tolist()} counts {cnt[:16].tolist()}")
    cb = (idx[:, 1] // 32).unique(return_counts=True)
    print("   channel block:", cb[0].tolist(), cb[1].tolist())
    xm = (idx[:, 3] % 32).unique(return_counts=True)
    print("   x % 32:", xm[0].tolist()[:32], xm[1].tolist()[:32])
    i = idx[0].tolist()
    print("   first:", i, float(got[tuple(i)]), float(ref[tuple(i)]))


with torch.no_grad():
    if os.environ.get("BNECK_NOFLAG"):
        Fn.X3_GUARD.word(torch.device(T.DEV))
        from dvis_plus_amd import native
        native.lib().dvis_x3_set_range_flag(None)
        print("range flag unregistered")
    for (N, H, W, nb) in [(3, 5, 64, 3), (3, 5, 64, 2), (2, 9, 40, 2), (4, 30, 96, 3), (4, 30, 96, 2), (8, 64, 320, 3)]:
        g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
        blocks = T._blocks(3, g, integer=True)[:nb]
        x = (torch.rand(N, 64, H, W, generator=g) < 0.3).float().to(T.DEV) * torch.randint(1, 4, (N, 64, H, W), generator=g).float().to(T.DEV)
        refs, amax = T._reference(x, blocks)
        for rep in range(2):
            got = Fn.bneck_stage_x3(x, blocks)
            torch.cuda.synchronize()
            report(f"({N},{H},{W}) blocks {nb} rep {rep} amax {amax}", got, refs[-1])
