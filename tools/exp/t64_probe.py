"""Where does a 64-frame segmenter batch spend its time?  Forward hooks with a device sync after the main blocks."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
m = build_dvis_plus_r50("offline", task="vps").to(dev).eval()
t0 = time.time()
def hook(name):
    def f(mod, inp, out):
        torch.cuda.synchronize()
        print(f"{time.time() - t0:7.2f} s  after {name}", flush=True)
    return f
bb, pd = m.backbone, m.sem_seg_head.pixel_decoder
for n in ("stem", "res2", "res3", "res4", "res5"):
    getattr(bb, n).register_forward_hook(hook("backbone." + n))
for i, p in enumerate(pd.input_proj):
    p[0].register_forward_hook(hook(f"input_proj[{i}].conv"))
for i, l in enumerate(pd.transformer.encoder.layers):
    l.self_attn.register_forward_hook(hook(f"encoder[{i}].self_attn"))
    l.register_forward_hook(hook(f"encoder[{i}]"))
pd.lateral_convs[0].register_forward_hook(hook("lateral_conv"))
pd.output_convs[0].register_forward_hook(hook("output_conv"))
pd.mask_features.register_forward_hook(hook("mask_features"))
m.sem_seg_head.predictor.register_forward_hook(hook("decoder"))
x = torch.rand(T, 3, 720, 1280, device=dev) * 255
with torch.no_grad():
    images, _ = m.preprocess(x)
    print(f"{time.time() - t0:7.2f} s  preprocessed {tuple(images.shape)}", flush=True)
    m.segment(images)
torch.cuda.synchronize()
print(f"{time.time() - t0:7.2f} s  done (first call: includes MIOpen's solver search)", flush=True)
