"""Development: where do the hipGraph replay of the online segmenter and the eager launches differ?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

dev = "cuda:0"
m = build_dvis_plus_r50("online", task="vps", object_mask_threshold=0.008).to(dev)
g = torch.Generator().manual_seed(11)
frames = torch.randint(0, 256, (5, 3, 360, 640), generator=g, dtype=torch.uint8).to(dev)
with torch.no_grad():
    images, _ = m.preprocess(frames)
    e1 = [t.clone() for t in m.segment(images)]
    e2 = [t.clone() for t in m.segment(images)]
    print("eager vs eager:", [bool(torch.equal(a, b)) for a, b in zip(e1, e2)])
    assert m._segmenter_graph_ok(images)
    for k in range(3):
        gr = [t.clone() for t in m._seg_graph(("dbg",), images)]
        print(f"graph call {k} vs eager:", [bool(torch.equal(a, b)) for a, b in zip(e1, gr)],
              [float((a - b).abs().max()) for a, b in zip(e1, gr)])
    # whole forward
    v = {"image": frames, "height": 360, "width": 640}
    os.environ["DVIS_SEGMENTER_GRAPH"] = "0"
    a = m([v])
    a = {k: (x.clone() if torch.is_tensor(x) else x) for k, x in a.items()}
    b = m([v])
    print("forward eager vs eager:", bool(torch.equal(a["pred_masks"], b["pred_masks"])), a["segments_infos"] == b["segments_infos"])
    os.environ["DVIS_SEGMENTER_GRAPH"] = "8"
    for k in range(3):
        c = m([v])
        print(f"forward graph {k} vs eager:", bool(torch.equal(a["pred_masks"], c["pred_masks"])), a["segments_infos"] == c["segments_infos"],
              float((a["pred_masks"] != c["pred_masks"]).float().mean()))
