"""Product kernel A/B on the GPU box: 8 x 8 query tiles vs 64 consecutive queries per workgroup, inputs as the
pixel decoder issues them (30 frames, 720p maps, init-rule offsets + a learned part of PROBE_SPREAD pixels)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from dvis_plus_amd.functions import msda_fused_forward  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, M, D, L, P = int(os.environ.get("PROBE_N", "30")), 8, 32, 3, 4
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
for spread in [float(x) for x in os.environ.get("PROBE_SPREAD", "0.16,1.0").split(",")]:
    off = (bias[None] + spread * torch.randn(N * Lq, M, L, P, 2, device=dev)).reshape(N * Lq, -1).contiguous()
    lg = (0.1 * torch.randn(N * Lq, M * L * P, device=dev)).contiguous()
    res = {}
    for knob, name in (("0", "64 consecutive queries per workgroup"), ("2", "8x8 query tiles (shapes known on the host)")):
        run = lambda: msda_fused_forward(value, shapes, lsi, ref, off, lg, L, P, shapes_host=shapes_py if knob == "2" else None)
        for _ in range(3):
            out = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res[knob] = out
        print(f"spread {spread:4.2f} px  {name:48s} {us:8.1f} us/launch = {us / N:6.2f} us/frame-layer "
              f"= {61824000 * N / us / 1e3:7.1f} GB/s algorithmic")
    for k in ("2",):
        d = (res["0"] - res[k]).abs()
        per_level = [float(d[:, a:b].max()) for a, b in ((0, 920), (920, 4600), (4600, 19320))]
        print(f"   knob {k} vs tile kernel: bit-identical {torch.equal(res['0'], res[k])}, max|diff| {float(d.max()):.2e}, "
              f"per query level {per_level}, differing fraction {float((d > 0).float().mean()):.4f}")
