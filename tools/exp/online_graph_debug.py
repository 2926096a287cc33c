"""Development (round 5, UNRESOLVED — the feature was removed): a hipGraph of the online mode's segmenter (GraphRunner around
model.segment for windows of <= 8 frames; config #2 spends 17 % of its wall waiting for ~450 host launches per window).
Needs `model._seg_graph = GraphRunner(lambda im: tuple(model.segment(im)))` and `_segmenter_graph_ok` back in
meta_architecture.py to run.  Findings: replays are bit-equal to the eager launches — until the referring tracker runs once
(hipGraph replay OR eager): from then on every captured segmenter graph returns wrong DECODER outputs (mask_features stay right),
an eager segmenter pass (or a fresh capture) makes the old graphs right again.  Not the weights caches (pointers and keys
unchanged), not the x3 / pyramid / own-GEMM / short-attention choices (all four switched: same picture)."""
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
    pred = m.sem_seg_head.predictor
    def state():
        kv = pred._kv_cache
        ptrs = [t.data_ptr() for lvl in kv[1] if lvl is not None for t in lvl[1:]]
        packs = {k: (v[1][0].data_ptr(), v[3]) for k, v in __import__("dvis_plus_amd.functions", fromlist=["x"])._X3_PACKED.d.items()}
        return kv[0], ptrs, packs
    s0 = state()
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

s1 = state()
print("kv cache key same:", s0[0] == s1[0], "kv tensors same:", s0[1] == s1[1], "packs: before", len(s0[2]), "after", len(s1[2]),
      "common packs unchanged:", all(s1[2][k] == v for k, v in s0[2].items() if k in s1[2]), "dropped:", [k for k in s0[2] if k not in s1[2]][:5])
print("---- proxy experiment")
real = m._seg_graph


class Proxy:
    def __init__(self):
        self._cache = real._cache

    def __call__(self, key, *t):
        out = real(key, *t)
        torch.cuda.synchronize()
        print("   seg graph inside forward vs eager:", [bool(torch.equal(a, b)) for a, b in zip(e1, out)], "cache entries", len(real._cache))
        self.last = out
        return out


m._seg_graph = Proxy()
with torch.no_grad():
    for k in range(3):
        c = m([v])
        torch.cuda.synchronize()
        print(f"forward graph {k} vs eager:", bool(torch.equal(a["pred_masks"], c["pred_masks"])),
              "static outputs after the forward still equal eager:", [bool(torch.equal(x, y)) for x, y in zip(e1, m._seg_graph.last)])
    # clone the outputs before use
    class CloneProxy(Proxy):
        def __call__(self, key, *t):
            return tuple(x.clone() for x in real(key, *t))
    m._seg_graph = CloneProxy()
    for k in range(2):
        c = m([v])
        print(f"forward graph (outputs cloned) {k} vs eager:", bool(torch.equal(a["pred_masks"], c["pred_masks"])))

print("---- sequence experiment")
m._seg_graph = real
with torch.no_grad():
    def chk(tag, key):
        out = [t.clone() for t in real(key, images)]
        print(f"   {tag}:", [bool(torch.equal(x, y)) for x, y in zip(e1, out)])
    key2 = [k for k in real._cache if k[0] != ("dbg",)][0][0]
    chk("entry dbg after the forwards", ("dbg",))
    chk("entry #2", key2)
    chk("entry #2 again", key2)
    chk("entry dbg again", ("dbg",))
    real(("fresh",), images)
    chk("fresh entry captured now (1st replay after capture)", ("fresh",))
    chk("fresh entry 2nd", ("fresh",))
    ee = [t.clone() for t in m.segment(images)]
    print("   eager now vs eager at start:", [bool(torch.equal(x, y)) for x, y in zip(e1, ee)])
    chk("fresh entry after an eager segment", ("fresh",))
    print("---- which event breaks a captured segmenter graph?")
    to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
    m.tracker(to_bctq(e1[0]), e1[3].unsqueeze(0), resume=False, frame_embeds_no_norm=to_bctq(e1[1]), need_masks=False)
    chk("fresh after a tracker call (graph replay)", ("fresh",))
    m.tracker.use_graphs = False
    m.tracker(to_bctq(e1[0]), e1[3].unsqueeze(0), resume=False, frame_embeds_no_norm=to_bctq(e1[1]), need_masks=False)
    m.tracker.use_graphs = True
    chk("fresh after an eager tracker call", ("fresh",))
    real(("fresh2",), images)
    chk("fresh after capturing fresh2", ("fresh",))
    proj = m.tracker.project_mask_features(e1[3])
    chk("fresh after project_mask_features", ("fresh",))
    c = m([v])
    chk("fresh after a whole forward", ("fresh",))
    chk("fresh2 after a whole forward", ("fresh2",))
