"""Probe (not product): the pyramid form of the decoder's attention masks with its two small kernels (pool_threshold.hip)
around the existing contraction, against dvis_attn_mask: bits, allowed counts, time per clip.
    python tools/exp/attn_mask_pyramid/run.py

Measured on MI355X at the end of round 4 (profiles/r04_attn_mask_pyramid_kernels.txt): 1.72 ms per clip against 3.70 ms.
Wiring it into the product (not done: no GPU budget was left to re-run the parity tests that count attention-mask flips):
  * the two kernels into csrc/fused_elementwise.hip behind C-ABI entry points (+ include/dvis_hip.h, native.SIGNATURES),
    better: the threshold + allowed count as an epilogue mode of mask_gemm.hip's MODE 0 (no fp32 logits written);
  * transformer_decoder._run_layers: pool mask_features once per call (the three maps serve all nine layers), then
    `Fn.attn_mask(emb, mask_features, size)` -> contraction on the level's pooled map + threshold;
  * the fall-back for maps whose H or W is not a multiple of 8 stays dvis_attn_mask;
  * re-run: tests/test_mask_gemm_gpu.py, test_golden_gpu.py (g3), test_model_gpu.py, test_pipeline_720p_gpu.py (the literal
    1e-3 test: its error is a sum of mask-flip events), test_properties_gpu.py."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from dvis_plus_amd import functions as Fn   # noqa: E402

so = os.path.join(HERE, "libpool_threshold.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "pool_threshold.hip"), "-o", so])
lib = ctypes.CDLL(so)
P, LL, I = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
lib.probe_center_pool3.argtypes = [P, P, P, P, LL, I, I, P]
lib.probe_threshold_count.argtypes = [P, P, P, LL, I, P]

dev = torch.device("cuda", 0)
B, Q, C, H, W = 30, 100, 256, 184, 320
g = torch.Generator(device=dev).manual_seed(0)
emb = torch.randn(B, Q, C, device=dev, generator=g)
mf = torch.randn(B, C, H, W, device=dev, generator=g)
stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pool_torch(f, s):
    o = s // 2 - 1
    a, b = f[:, :, o::s, o::s], f[:, :, o::s, o + 1::s]
    c, d = f[:, :, o + 1::s, o::s], f[:, :, o + 1::s, o + 1::s]
    return ((a + b) + (c + d)) * 0.25


pooled = {s: torch.empty(B, C, H // s, W // s, device=dev) for s in (2, 4, 8)}


def pool():
    rc = lib.probe_center_pool3(mf.data_ptr(), pooled[2].data_ptr(), pooled[4].data_ptr(), pooled[8].data_ptr(), B * C, H, W, stream())
    assert rc == 0, rc


def threshold(logits):
    rows, n = logits.shape[0] * logits.shape[1], logits.shape[2] * logits.shape[3]
    mask = torch.empty(logits.shape[0], logits.shape[1], n, dtype=torch.uint8, device=dev)
    allowed = torch.empty(logits.shape[0], logits.shape[1], dtype=torch.int32, device=dev)
    rc = lib.probe_threshold_count(logits.data_ptr(), mask.data_ptr(), allowed.data_ptr(), rows, n, stream())
    assert rc == 0, rc
    return mask, allowed


with torch.no_grad():
    pool()
    for s in (2, 4, 8):
        print(f"pooled map s = {s}: torch.equal to the torch expression: {torch.equal(pooled[s], pool_torch(mf, s))}")
    us_pool = t(pool)
    print(f"center_pool3: {us_pool:.1f} us per clip ({(mf.numel() * 4 * (1 + 1 / 4 + 1 / 16 + 1 / 64)) / us_pool / 1e6:.2f} TB/s)")
    tot_old = tot_new = 0.0
    for s, (h, w) in ((8, (23, 40)), (4, (46, 80)), (2, (92, 160))):
        old_mask, old_allowed = Fn.attn_mask(emb, mf, (h, w))
        logits = Fn.mask_logits(emb, pooled[s])
        mask, allowed = threshold(logits)
        ok_thr = torch.equal(mask, (logits.view(B, Q, -1) < 0).to(torch.uint8)) and \
            torch.equal(allowed, (logits.view(B, Q, -1) >= 0).sum(-1).to(torch.int32))
        diff = int((mask != old_mask).sum())
        dall = int((allowed != old_allowed).sum())
        us_old = t(lambda: Fn.attn_mask(emb, mf, (h, w)))
        us_new = t(lambda: threshold(Fn.mask_logits(emb, pooled[s])))
        tot_old += us_old
        tot_new += us_new
        print(f"level {h}x{w}: threshold kernel == torch: {ok_thr}; {diff} of {mask.numel()} bits and {dall} of {allowed.numel()} "
              f"allowed counts differ from dvis_attn_mask; dvis_attn_mask {us_old:.1f} us, contraction + threshold {us_new:.1f} us")
    print(f"per clip: dvis_attn_mask x 9 = {3 * tot_old / 1e3:.2f} ms; pyramid = pool {us_pool / 1e3:.2f} + 9 x (contraction + threshold) "
          f"{3 * tot_new / 1e3:.2f} = {(us_pool + 3 * tot_new) / 1e3:.2f} ms")
