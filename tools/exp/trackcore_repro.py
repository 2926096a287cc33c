"""Is _track_core (tracker + refiner) a pure function of its inputs?  Records (inputs, outputs) of every in-stream call
during stream(), then recomputes from the recorded inputs afterwards."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
import pipeline_parity as PPar
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0)
if os.environ.get("SHARPEN", "1") == "1":
    PPar.perturb_msda(model.sem_seg_head.pixel_decoder)
    PPar.sharpen_masks(model, PPar.SHARPEN)
model = model.to(dev).eval()
model.tracker.fused_chain = os.environ.get("FUSED", "1") == "1"
g = torch.Generator().manual_seed(1)
clips = [{"image": torch.randint(0, 256, (T, 3, 360, 640), generator=g).float().to(dev), "height": 360, "width": 640}
         for T in (5, 4, 5)]
rec = []
core = model._track_core
model.debug_stages = {}
def spy(e, en):
    ein, enin = e.clone(), en.clone()
    out = core(e, en)
    st = {k: v.clone() for k, v in model.debug_stages.items() if torch.is_tensor(v)}
    rec.append(((ein, enin), tuple(o.clone() for o in out), st, (e.clone(), en.clone())))
    return out
model._track_core = spy
for rnd in range(3):
    rec.clear()
    list(model.stream(clips))
    torch.cuda.synchronize()
    for ci, ((e, en), outs, st, (e2, en2)) in enumerate(list(rec)):
        with torch.no_grad():
            again = core(e, en)
        d = [float((a - b).abs().max()) for a, b in zip(again, outs)]
        print(f"round {rnd} clip {ci}: in-stream vs recompute max|d| {d}; inputs mutated by the call: {not (torch.equal(e, e2) and torch.equal(en, en2))}", flush=True)
        if rnd == 2:
            for k, v in st.items():
                w = model.debug_stages.get(k)
                if torch.is_tensor(w) and w.shape == v.shape:
                    print(f"      stage {k}: max|d| {float((v - w).abs().max()):.3e}")
# which part?  tracker alone
trk = model.tracker
(e, en) = rec[0][0]
to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
outs = []
for i in range(3):
    o = trk(to_bctq(e), None, resume=False, frame_embeds_no_norm=to_bctq(en), need_masks=False)
    outs.append({k: v.clone() for k, v in o.items() if torch.is_tensor(v)})
for k in outs[0]:
    print("tracker repeat", k, float((outs[0][k] - outs[1][k]).abs().max()), float((outs[1][k] - outs[2][k]).abs().max()))
