import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
dev = "cuda:0"
m = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.008).to(dev)
def clip(T, seed):
    g = torch.Generator().manual_seed(seed)
    return {"image": torch.randint(0, 256, (T, 3, 360, 640), generator=g, dtype=torch.uint8).to(dev), "height": 360, "width": 640}
clips = [clip(7, 20), clip(7, 21)]
with torch.no_grad():
    m([clips[0]])
    sts = [m._segment_phase(c) for c in clips]
    def run_single(st):
        m.debug_stages = {}
        out = m._track_phase(dict(st))
        d = dict(m.debug_stages); d["masks_all"] = d["mask_fn"](None).clone(); d["pan"] = out["pred_masks"].clone()
        return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items() if k != "mask_fn"}
    a0, a1 = run_single(sts[0]), run_single(sts[1])
    # batched
    stash = []
    fin = m._finish_phase
    def spy(st, mask_embed, cls, aux):
        stash.append((mask_embed.clone(), cls.clone(), aux.clone()))
        return fin(st, mask_embed, cls, aux)
    m._finish_phase = spy
    m.debug_stages = None
    outs = m._track_phase_batched([dict(s) for s in sts])
    m._finish_phase = fin
    for j, a in enumerate((a0, a1)):
        me, cls, aux = stash[j]
        print(j, "refiner_mask_embed", torch.equal(me, a["refiner_mask_embed"]), "cls", torch.equal(cls, a["cls"]), "aux", torch.equal(aux, a["aux"]),
              "pan", torch.equal(outs[j]["pred_masks"], a["pan"]), float((me - a["refiner_mask_embed"]).abs().max()), float((aux - a["aux"]).abs().max()))
