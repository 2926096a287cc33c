"""Which (rows, N, K) go to which split-f16 linear kernel in the ViT-Adapter-L configuration (one 30-frame clip)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn
from dvis_plus_amd.meta_architecture import build_dvis_plus
dev = torch.device("cuda:0")
m = build_dvis_plus("offline", task="vps", backbone="vitl", num_queries=200).to(dev).eval()
clip = torch.randint(0, 256, (30, 3, 720, 1280), dtype=torch.uint8, device=dev)
video = {"image": clip, "height": 720, "width": 1280}
seen = collections.Counter()
for name in ("x3_linear", "x3_tile_linear", "x3_linear_ln", "x3_ffn_ln"):
    orig = getattr(Fn, name)
    def wrap(*a, _o=orig, _n=name, **k):
        x = a[0]
        w = a[1] if _n != "x3_ffn_ln" else a[1].weight
        seen[(_n, x.numel() // x.shape[-1], tuple(w.shape), k.get("act"), k.get("residual") is not None, k.get("xadd") is not None)] += 1
        return _o(*a, **k)
    setattr(Fn, name, wrap)
with torch.no_grad():
    m([video])
    seen.clear()
    m([video])
for k, c in sorted(seen.items(), key=lambda kv: -kv[0][1] * kv[0][2][0] * kv[0][2][1] * kv[1]):
    print(c, k)
