"""Does the split-f16 key-split attention want K / V head-major?  Calls dvis_attention_forward_k with explicit strides:
row-major (Lk, B, 3 x 256) slices as the decoder holds them today against (B, H, Lk, 32) planes.  (development)"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn, native  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
lib = native.lib()


def run(q, k, v, out, mask, allowed, ks, vs, kern, ws):
    Lq, B, C = q.shape
    H, d = 8, 32
    Lk = mask.shape[-1]
    qs = (ctypes.c_int64 * 3)(q.stride(1), d, q.stride(0))
    os_ = (ctypes.c_int64 * 3)(out.stride(1), d, out.stride(0))
    rc = lib.dvis_attention_forward_k(ctypes.c_void_p(q.data_ptr()), qs, ctypes.c_void_p(k.data_ptr()), (ctypes.c_int64 * 3)(*ks),
                                      ctypes.c_void_p(v.data_ptr()), (ctypes.c_int64 * 3)(*vs), ctypes.c_void_p(out.data_ptr()), os_,
                                      ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(allowed.data_ptr()), B, H, Lq, Lk, d, 1.0 / d ** 0.5,
                                      ctypes.c_void_p(ws.data_ptr()), native.stream_ptr(q.device), kern)
    native.check(rc, "attention")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


with torch.no_grad():
    Fn.X3_GUARD.word(torch.device(dev))
    for Lk, B in ((920, 30), (3680, 30), (14720, 30)):
        Lq, H, d, C = 100, 8, 32, 256
        q = torch.randn(Lq, B, C, device=dev)
        kall = torch.randn(B, Lk, 3 * C, device=dev)                 # as the projection leaves it today: rows of 3 layers x 256
        vall = torch.randn(B, Lk, 3 * C, device=dev)
        k_rm, v_rm = kall[..., C:2 * C], vall[..., C:2 * C]          # layer 1's slice: row stride 768
        k_hm = k_rm.reshape(B, Lk, H, d).permute(0, 2, 1, 3).contiguous()      # (B, H, Lk, 32)
        v_hm = v_rm.reshape(B, Lk, H, d).permute(0, 2, 1, 3).contiguous()
        mask = (torch.rand(B, Lq, Lk, device=dev) < 0.6).view(torch.uint8) if False else (torch.rand(B, Lq, Lk, device=dev) < 0.6).to(torch.uint8)
        allowed = (mask == 0).sum(-1).to(torch.int32)
        ws = torch.empty(lib.dvis_attention_ws_bytes_k(B * H, Lq, Lk, d, 0) // 4 + 16, device=dev)
        outs = {}
        for name, kk, vv, ks, vs in (("row-major", k_rm, v_rm, (Lk * 3 * C, d, 3 * C), (Lk * 3 * C, d, 3 * C)),
                                     ("head-major", k_hm, v_hm, (H * Lk * d, Lk * d, d), (H * Lk * d, Lk * d, d))):
            for kern in (0, 3):
                out = torch.empty(Lq, B, C, device=dev)
                t = timed(lambda: run(q, kk, vv, out, mask, allowed, ks, vs, kern, ws))
                outs[(name, kern)] = out.clone()
                print(f"Lk {Lk:6d}  {name:10s}  kernel {'split-f16' if kern == 3 else 'fp32     '}  {t:7.1f} us")
        assert torch.equal(outs[("row-major", 0)], outs[("head-major", 0)]) and torch.equal(outs[("row-major", 3)], outs[("head-major", 3)])
    Fn.X3_GUARD.check_now(torch.device(dev))
