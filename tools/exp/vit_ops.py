"""Which aten ops does the ViT-Adapter-L backbone dispatch per call, with what sizes?  (finding the elementwise / copy passes)"""
import os, sys, collections
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd.meta_architecture import build_dvis_plus
dev = torch.device("cuda:0")
m = build_dvis_plus("offline", task="vps", backbone="vitl", num_queries=200).to(dev).eval()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(T, 3, 736, 1280, device=dev)
seen = collections.Counter()
import traceback
where = {}

class Watch(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        n = 0
        for a in list(args) + ([out] if torch.is_tensor(out) else []):
            if torch.is_tensor(a):
                n = max(n, a.numel())
        if n >= T * 3681 * 256 and name not in ("view", "_unsafe_view", "transpose", "slice", "t", "expand", "unsqueeze", "select", "permute", "reshape", "detach", "as_strided", "empty", "empty_like", "empty_strided", "alias", "unbind", "split", "squeeze"):
            key = (name, n // (T * 3681))
            seen[key] += 1
            if key not in where:
                st = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "dvis_plus_amd" in f.filename]
                where[key] = " < ".join(st[-3:])
        return out

with torch.no_grad():
    if len(sys.argv) > 2 and sys.argv[2] == "model":
        clip = torch.randint(0, 256, (T, 3, 720, 1280), dtype=torch.uint8, device=dev)
        video = {"image": clip, "height": 720, "width": 1280}
        m([video])
        with Watch():
            m([video])
    else:
        m.backbone(x)
        with Watch():
            m.backbone(x)
for (name, per_tok), c in sorted(seen.items(), key=lambda kv: -kv[1] * kv[0][1]):
    print(f"{c:5d} x {name:28s} {per_tok:6d} elements per (frame, token)   {where[(name, per_tok)]}")
