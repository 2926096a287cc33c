"""MSDA forward at the ViT-Adapter extractor shape (value (N,3680,16,64), Lq=19320, L=1, P=4), timed.
Queries are the 3-level pyramid tokens, the single value level is the stride-16 ViT map (46x80).  Dev tool."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd import native  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
shapes = torch.tensor([(46, 80)], dtype=torch.long, device=dev)
lsi = torch.zeros(1, dtype=torch.long, device=dev)
M, D, L, P, S, Lq = 16, 64, 1, 4, 3680, 19320
g = torch.Generator(device=dev).manual_seed(0)
value = torch.randn(N, S, M, D, device=dev, generator=g)
ref = []
for (h, w) in [(92, 160), (46, 80), (23, 40)]:
    ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, device=dev) / h,
                            torch.linspace(0.5, w - 0.5, w, device=dev) / w, indexing="ij")
    ref.append(torch.stack((xs.reshape(-1), ys.reshape(-1)), -1))
ref = torch.cat(ref, 0)[None, :, None, None, None, :]
off = torch.randn(N, Lq, M, L, P, 2, device=dev, generator=g) * 2.5 / torch.tensor([80.0, 46.0], device=dev)
loc = (ref + off).contiguous()
w = torch.softmax(torch.randn(N, Lq, M, L * P, device=dev, generator=g), -1).view(N, Lq, M, L, P).contiguous()
out = torch.empty(N, Lq, M * D, device=dev)
alg = 4 * (S * M * D + Lq * M * L * P * 3 + Lq * M * D) * N
lib = native.lib()
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run():
    assert lib.dvis_msda_forward(0, p(value), p(shapes), p(lsi), p(loc), p(w), N, S, M, D, L, Lq, P, p(out), st) == 0


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"ViT-L extractor op, N={N}: {ms:.3f} ms/launch = {ms / N * 1e3:.1f} us/frame, {alg / ms / 1e6:.0f} GB/s algorithmic "
      f"({alg / N / 1e6:.1f} MB/frame), variant {os.environ.get('DVIS_MSDA_VARIANT', 'default')}")
