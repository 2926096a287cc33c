"""Launch by launch: which launch of the bottleneck chain first differs from the fp64 reference?  (development aid)"""
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


def decode(img, N, H, W, xexp):
    XG = (W + 31) // 32
    t = img.view(torch.float16).view(N, H, XG, 4, 2, 2, 32, 8).float()      # n y xg S hl g x e
    v = (t[:, :, :, :, 0] + t[:, :, :, :, 1]) / (2.0 ** xexp)               # n y xg S g x e
    out = torch.zeros(N, 64, H, XG * 32, device=img.device)
    for S in range(4):
        for g in range(2):
            for e in range(8):
                c = 32 * (S >> 1) + 16 * (S & 1) + 8 * (e >> 2) + 4 * g + (e & 3)
                out[:, c] = v[:, :, :, S, g, :, e].reshape(N, H, XG * 32)
    return out[:, :, :, :W]


def cmp(tag, got, ref):
    bad = got.double() != ref.double()
    n = int(bad.sum())
    s = f"  {tag}: {n} of {bad.numel()} differ"
    if n:
        idx = bad.nonzero()
        s += f"; x % 32 in {sorted(set((idx[:, 3] % 32).tolist()))}, c in {sorted(set(idx[:, 1].tolist()))[:24]}..., first {idx[0].tolist()} {float(got[tuple(idx[0].tolist())])} vs {float(ref[tuple(idx[0].tolist())])}"
    print(s)
    return n


with torch.no_grad():
    N, H, W = [int(v) for v in os.environ.get("BNECK_DEBUG_SHAPE", "4,30,96").split(",")]
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + W)
    blocks = T._blocks(3, g, integer=True)
    x = (torch.rand(N, 64, H, W, generator=g) < 0.3).float().to(DEV) * torch.randint(1, 4, (N, 64, H, W), generator=g).float().to(DEV)
    # reference intermediates
    refs = []
    xx = x.double()
    for b in blocks:
        c = {k: (None if v is None else v.double()) for k, v in b.items()}
        a1 = F.relu(F.conv2d(xx, c["w1"], c["b1"]))
        a2 = F.relu(F.conv2d(a1, c["w2"], c["b2"], padding=1))
        sc = xx if c["ws"] is None else F.conv2d(xx, c["ws"], c["bs"])
        xx = F.relu(F.conv2d(a2, c["w3"], c["b3"]) + sc)
        refs.append((a1, xx))
    xe = Fn.X3_CONV_XEXP
    sp = native.stream_ptr(torch.device(DEV))

    def P(t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    def pack1(w):
        w2d = w.reshape(64, 64).contiguous()
        e = Fn._x3_exp(w2d)
        buf = torch.empty(lib.dvis_conv1x1_x3_packed_bytes(64, 64), dtype=torch.uint8, device=DEV)
        native.check(lib.dvis_conv1x1_x3_pack(P(w2d), 64, 64, e, P(buf), sp), "pack")
        return buf, e

    def packb(b, nxt):
        w2, w3 = b["w2"].contiguous(), b["w3"].reshape(256, 64).contiguous()
        ws = None if b["ws"] is None else b["ws"].reshape(256, 64).contiguous()
        w1 = None if nxt is None else nxt["w1"].reshape(64, 256).contiguous()
        e2, e1 = Fn._x3_exp(w2), (0 if w1 is None else Fn._x3_exp(w1))
        e3 = Fn._x3_exp(w3 if ws is None else torch.cat([w3, ws], 1))
        buf = torch.empty(lib.dvis_bneck_x3_packed_bytes(0 if w1 is None else 1, 0 if ws is None else 1), dtype=torch.uint8, device=DEV)
        native.check(lib.dvis_bneck_x3_pack(P(w2), P(w3), P(ws), P(w1), e2, e3, e1, P(buf), sp), "packb")
        b3 = b["b3"] if b["bs"] is None else b["b3"] + b["bs"]
        return buf, e2, e3, e1, b3.contiguous()

    nbytes = lib.dvis_bneck_x3_image_bytes(N, H, W)
    MODE = os.environ.get("BNECK_DEBUG_MODE", "sync")
    buf1, e1 = pack1(blocks[0]["w1"])
    packs = [packb(b, blocks[i + 1] if i + 1 < len(blocks) else None) for i, b in enumerate(blocks)]
    torch.cuda.synchronize()
    for rep in range(int(os.environ.get("BNECK_DEBUG_REPS", "4"))):
        print("rep", rep, MODE)
        fill = 0x7e if MODE == "sync" else (0 if rep % 2 == 0 else 0x3c)
        imgs = [torch.full((nbytes,), fill, dtype=torch.uint8, device=DEV) for _ in range(len(blocks))]
        ys = [torch.full((N, 256, H, W), float("nan") if MODE == "sync" else 1.0, device=DEV) for _ in blocks]
        if MODE == "alias":      # two alternating buffers, as functions.bneck_stage_x3 uses them
            imgs = [imgs[i % 2] for i in range(len(blocks))]
            ys = [ys[i % 2] for i in range(len(blocks))]
            keep = []
        torch.cuda.synchronize()
        native.check(lib.dvis_conv1x1_x3_image(P(x), P(buf1), P(blocks[0]["b1"]), P(imgs[0]), N, 64, H, W, xe, e1, xe, 1, sp), "img")
        for i, b in enumerate(blocks):
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            buf, e2, e3, e1n, b3 = packs[i]
            if MODE == "sync":
                torch.cuda.synchronize()
            native.check(lib.dvis_bneck_x3(P(imgs[i]), None if i == 0 else P(ys[i - 1]), P(x) if i == 0 else None, P(buf), P(b["b2"]), P(b3),
                                           None if nxt is None else P(nxt["b1"]), P(ys[i]), None if nxt is None else P(imgs[i + 1]),
                                           N, H, W, xe, e2, e3, e1n, sp), "bneck")
            if MODE == "alias":
                keep.append((ys[i].clone(), None if nxt is None else imgs[i + 1].clone()))
        torch.cuda.synchronize()
        if MODE == "alias":
            ys = [k[0] for k in keep]
            imgs = [imgs[0]] + [k[1] for k in keep[:-1]]
        cmp("first conv1 -> image", decode(imgs[0], N, H, W, xe), refs[0][0])
        for i in range(len(blocks)):
            cmp(f"block {i} y", ys[i], refs[i][1])
            if i + 1 < len(blocks):
                cmp(f"block {i} -> next conv1 image", decode(imgs[i + 1], N, H, W, xe), refs[i + 1][0])
