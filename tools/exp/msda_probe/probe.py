"""Timing experiments on the MSDA tile kernel (dev tool, GPU box):  python tools/exp/msda_probe/probe.py [exp ...]
Inputs as the pixel decoder issues them: 30 frames, 720p shapes, init-rule offsets + a small learned part."""
import ctypes
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmsda_probe.so"))
lib.msda_probe.restype = ctypes.c_int
dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, M, D, L, P = int(os.environ.get("PROBE_N", "30")), 8, 32, 3, 4
shapes_py = [(23, 40), (46, 80), (92, 160)]
shapes = torch.tensor(shapes_py, dtype=torch.long, device=dev)
lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
S = Lq = int(shapes.prod(1).sum())
value = torch.randn(N, S, M, D, device=dev)
value_hm = value.permute(0, 2, 1, 3).contiguous()
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w,
                                            indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in shapes_py])
ref = ref[None, :, None, :].expand(1, Lq, L, 2).contiguous()
# init rule: head m looks along angle 2 pi m / M on the unit square, point p sits p + 1 steps out; + learned part
ang = torch.arange(M, device=dev) * (2 * math.pi / M)
d = torch.stack([ang.cos(), ang.sin()], -1)
d = d / d.abs().max(-1, keepdim=True)[0]
bias = (d[:, None, None, :] * torch.arange(1, P + 1, device=dev)[None, None, :, None]).expand(M, L, P, 2)
spread = float(os.environ.get("PROBE_SPREAD", "0.16"))          # std of the learned part in pixels (0.01 * |src| ~ 0.16)
off = (bias[None] + spread * torch.randn(N * Lq, M, L, P, 2, device=dev)).reshape(N * Lq, -1).contiguous()
lg = (0.1 * torch.randn(N * Lq, M * L * P, device=dev)).contiguous()
out = torch.empty(N, Lq, M * D, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(exp):
    v = value_hm if exp & 4 else value
    rc = lib.msda_probe(exp, p(v), p(shapes), p(lsi), p(ref), 1, p(off), ctypes.c_int64(off.stride(0)), p(lg),
                        ctypes.c_int64(lg.stride(0)), N, S, M, Lq, p(out), st)
    assert rc == 0, rc


def timeit(exp, iters=20):
    for _ in range(3):
        run(exp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run(exp)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


names = {0: "baseline", 64: "coarsest level via TA -> LDS (buffer_load lds) + ds_read_b128", 65: "level 2 only, TA -> LDS (n/a: coarsest skipped)", 1: "gather level 2 only", 2: "no gather (set-up + stores)", 4: "head-major value", 5: "head-major, level 2 only", 8: "left corners of odd queries skipped", 16: "left corners of all queries skipped", 32: "one corner of four", 100: "4 samples (16 loads) in flight per wave", 101: "1 sample (4 loads) in flight per wave"}
exps = [int(a) for a in sys.argv[1:]] or [0, 4, 1, 5, 2]
base = None
for e in exps:
    if e in (0, 4, 64):
        run(e)
        cur = out.clone()
        if base is None:
            base = cur
        else:
            print(f"   max|diff| vs baseline: {(cur - base).abs().max().item():.1e}")
    us = timeit(e)
    print(f"exp {e} ({names[e]:28s}): {us:8.1f} us/launch = {us / N:6.2f} us/frame-layer")
