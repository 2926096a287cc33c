import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch, torch.nn.functional as F
from dvis_plus_amd.functions import conv1x1_bias_act
dev = "cuda:0"
N, K, M, H, W = 2, 256, 64, 24, 40
g = torch.Generator().manual_seed(1)
x = torch.randn(N, K, H, W, generator=g).to(dev); w = (torch.randn(M, K, 1, 1, generator=g) / 16).to(dev); b = torch.randn(M, generator=g).to(dev)
with torch.no_grad():
    ref = F.conv2d(x.double(), w.double(), b.double()).float()
    bad = 0
    for it in range(200):
        out = conv1x1_bias_act(x, w, b, None, False)
        d = (out - ref).abs() > 1e-3
        if d.any():
            bad += 1
            idx = d.nonzero()
            if bad <= 6:
                print("iter", it, "n mismatches", len(idx), "frames", idx[:, 0].unique().tolist(), "rows", idx[:, 1].unique().tolist(),
                      "pixels", (idx[:, 2] * W + idx[:, 3]).unique().tolist())
                n0, m0, y0, x0 = idx[0].tolist()
                print("   out", float(out[n0, m0, y0, x0]), "ref", float(ref[n0, m0, y0, x0]), "bias", float(b[m0]))
    print("bad iterations:", bad, "of 200")
