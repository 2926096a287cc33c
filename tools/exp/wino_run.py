"""A few launches of the Winograd convolution at the FPN shape (profiling target: tools/prof.sh wino python tools/exp/wino_run.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn  # noqa: E402

C, K, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 256, 184, 320)))
N = int(sys.argv[5]) if len(sys.argv) > 5 else 30
x = torch.randn(N, C, H, W, device="cuda")
w = torch.randn(K, C, 3, 3, device="cuda") * 0.02
for _ in range(4):
    Fn.conv3x3_bias_act(x, w, None, False, winograd=True)
torch.cuda.synchronize()
