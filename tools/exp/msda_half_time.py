"""Fused MSDeformAttn at the pixel decoder's shape (30 frames of 720p): fp32 vs fp16 / bf16 storage (dvis_msda_fused_forward_h)."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd.functions import msda_fused_forward
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, M, D, L, P = 30, 8, 32, 3, 4
shapes_py = [(23, 40), (46, 80), (92, 160)]
shapes = torch.tensor(shapes_py, dtype=torch.long, device=dev)
lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
S = Lq = int(shapes.prod(1).sum())
value = torch.randn(N, S, M, D, device=dev)
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w,
                                            indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in shapes_py])
ref = ref[None, :, None, :].expand(1, Lq, L, 2).contiguous()
ang = torch.arange(M, device=dev) * (2 * math.pi / M)
d = torch.stack([ang.cos(), ang.sin()], -1)
d = d / d.abs().max(-1, keepdim=True)[0]
bias = (d[:, None, None, :] * torch.arange(1, P + 1, device=dev)[None, None, :, None]).expand(M, L, P, 2)
off = (bias[None] + 0.16 * torch.randn(N * Lq, M, L, P, 2, device=dev)).reshape(N * Lq, -1)
lg = 0.1 * torch.randn(N * Lq, M * L * P, device=dev)
proj = torch.cat([off, lg, torch.zeros(N * Lq, 32, device=dev)], 1).contiguous()        # 320-wide fused projection row
n_off = M * L * P * 2
for dt in (torch.float32, torch.float16, torch.bfloat16):
    v, p = value.to(dt), proj.to(dt)
    run = lambda: msda_fused_forward(v, shapes, lsi, ref, p[:, :n_off], p[:, n_off:], L, P, shapes_host=shapes_py)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    eb = v.element_size()
    alg = eb * (S * M * D + Lq * M * L * P * 3 + Lq * M * D)            # value + raw offsets / logits + output, per frame
    print(f"{str(dt):16s}: {us:8.1f} us/launch = {us / N:6.2f} us/frame-layer, algorithmic {alg / 1e6:.1f} MB/frame-layer -> "
          f"{alg * N / us / 1e6:.2f} TB/s = {alg * N / us / 1e6 / 8:.3f} of HBM peak", flush=True)
