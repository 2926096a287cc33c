"""First launch of the chain (conv1 64 -> 64 -> operand image): raw halves of the elements that differ."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_bneck_x3_gpu as T      # noqa: E402
from dvis_plus_amd import functions as Fn, native      # noqa: E402

lib = native.lib()
DEV = T.DEV
with torch.no_grad():
    N, H, W = 4, 30, 96
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
    blocks = T._blocks(3, g, integer=True)
    x = (torch.rand(N, 64, H, W, generator=g) < 0.3).float().to(DEV) * torch.randint(1, 4, (N, 64, H, W), generator=g).float().to(DEV)
    b = blocks[0]
    a1 = F.relu(F.conv2d(x.double(), b["w1"].double(), b["b1"].double()))
    xe = Fn.X3_CONV_XEXP
    sp = native.stream_ptr(torch.device(DEV))
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    w2d = b["w1"].reshape(64, 64).contiguous()
    e = Fn._x3_exp(w2d)
    buf = torch.empty(lib.dvis_conv1x1_x3_packed_bytes(64, 64), dtype=torch.uint8, device=DEV)
    native.check(lib.dvis_conv1x1_x3_pack(P(w2d), 64, 64, e, P(buf), sp), "pack")
    nbytes = lib.dvis_bneck_x3_image_bytes(N, H, W)
    XG = (W + 31) // 32
    for rep in range(3):
        # the fp32-map form of the same layer on the same data
        y = Fn.conv1x1_x3(x, b["w1"], b["b1"], None, True)
        print("rep", rep, "fp32 map form differs in", int((y.double() != a1).sum()))
        img = torch.full((nbytes,), 0x7e, dtype=torch.uint8, device=DEV)
        native.check(lib.dvis_conv1x1_x3_image(P(x), P(buf), P(b["b1"]), P(img), N, 64, H, W, xe, e, xe, 1, sp), "img")
        torch.cuda.synchronize()
        t = img.view(torch.float16).view(N, H, XG, 4, 2, 2, 32, 8).float().cpu()      # n y xg S hl g x e
        a1c = a1.cpu()
        bad = 0
        for S in range(4):
            for gg in range(2):
                for ee in range(8):
                    c = 32 * (S >> 1) + 16 * (S & 1) + 8 * (ee >> 2) + 4 * gg + (ee & 3)
                    hi, lo = t[:, :, :, S, 0, gg, :, ee], t[:, :, :, S, 1, gg, :, ee]
                    want = a1c[:, c].reshape(N, H, XG, 32) * 4
                    m = (hi + lo).double() != want
                    if m.any():
                        idx = m.nonzero()
                        bad += len(idx)
                        i = tuple(idx[0].tolist())
                        print(f"   S {S} g {gg} e {ee} (c {c}): {len(idx)} differ; lanes {sorted(set(idx[:, 3].tolist()))}; first {i}: hi {float(hi[i])} lo {float(lo[i])} "
                              f"want {float(want[i])}; neighbours e-1/e+1 hi {float(t[i[0], i[1], i[2], S, 0, gg, i[3], max(ee - 1, 0)])} "
                              f"{float(t[i[0], i[1], i[2], S, 0, gg, i[3], min(ee + 1, 7)])}; want of lane-1 {float(want[i[0], i[1], i[2], i[3] - 1])}")
        print("   image form differs in", bad)
