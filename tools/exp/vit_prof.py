"""torch.profiler over one ViT-Adapter-L clip: which aten ops own the elementwise / copy kernels (with input shapes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd.meta_architecture import build_dvis_plus
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
m = build_dvis_plus("offline", task="vps", backbone="vitl", num_queries=200).to(dev).eval()
T = 30
clip = torch.randint(0, 256, (T, 3, 720, 1280), dtype=torch.uint8, device=dev)
video = {"image": clip, "height": 720, "width": 1280}
with torch.no_grad():
    m([video]); m([video])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        m([video])
        torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
    if t > 500:
        rows.append((t, e.key, e.count, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
for t, k, c, sh in rows[:45]:
    print(f"{t / 1e3:9.2f} ms  {c:4d} x {k:40s} {sh}")
