import torch, sys
sys.path.insert(0, '.')
from dvis_plus_amd.functions import upsample_add
dev = torch.device('cuda', 0)
lat = torch.randn(30, 256, 184, 320, device=dev); top = torch.randn(30, 256, 92, 160, device=dev)
torch.set_grad_enabled(False)
upsample_add(lat, top); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): upsample_add(lat, top)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"upsample_add 30x256x184x320: {ms*1e3:.0f} us, {(2*lat.numel()+top.numel())*4/ms/1e9:.2f} TB/s")
