"""Which torch ops launch the `Cijk_Ailk_Bljk...UserArgs` library GEMMs (A operand not transposed) in one offline clip?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
m = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0).to(dev)
v = {"image": bench.synthetic_clip(30, dev, seed=1234), "height": 720, "width": 1280}
v["object_mask_threshold"] = bench.calibrate_threshold(m, [v], 20)
with torch.no_grad():
    m([v]); m([v]); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        m([v]); torch.cuda.synchronize()
ev = prof.events()
# map kernel events to their launching CPU op via correlation: use key_averages grouped by input shape for aten::addmm / mm / bmm / linear
rows = []
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA and "Ailk" in e.name and "UserArgs" in e.name:
        rows.append((e.name[:60], e.time_range.start, e.cuda_time if hasattr(e, "cuda_time") else e.device_time))
print(len(rows), "Ailk UserArgs kernels")
ka = prof.key_averages(group_by_input_shape=True)
for k in sorted(ka, key=lambda k: -(k.device_time_total if hasattr(k, "device_time_total") else k.cuda_time_total))[:40]:
    if k.key in ("aten::addmm", "aten::mm", "aten::bmm", "aten::linear", "aten::_addmm_activation", "aten::matmul"):
        tot = k.device_time_total if hasattr(k, "device_time_total") else k.cuda_time_total
        print(f"{k.key:26s} n={k.count:4d} total {tot/1e3:8.2f} ms  shapes {str(k.input_shapes)[:150]}")
