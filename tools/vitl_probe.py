"""ViT-L block pieces at the 720p token count (3681 tokens, 16 heads x 64, batch 30), timed (dev tool)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd import functions as Fn  # noqa: E402
from dvis_plus_amd.vit_adapter import Block  # noqa: E402

dev = "cuda:0"
B, N, C, H = int(sys.argv[1]) if len(sys.argv) > 1 else 30, 3681, 1024, 16


def t(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    blk = Block(C, H, 4.0, True, True, True, 1e-5).to(dev).eval()
    x = torch.randn(B, N, C, device=dev)
    print(f"block total          {t(lambda: blk(x)):8.2f} ms  (x24 per clip)")
    qkv = Fn.linear(x, blk.attn.qkv.weight, blk.attn.qkv.bias)
    v = qkv.transpose(0, 1)
    out = torch.empty(B, N, C, device=dev)
    ta = t(lambda: Fn.attention(v[..., :C], v[..., C:2 * C], v[..., 2 * C:], H, out=out.transpose(0, 1)))
    fl = 4.0 * B * H * N * N * 64
    print(f"  attention kernel   {ta:8.2f} ms  ({fl / ta / 1e9:6.1f} TFLOP/s)")
    q4 = qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ts = t(lambda: F.scaled_dot_product_attention(q4[0], q4[1], q4[2]))
    print(f"  torch SDPA (ref)   {ts:8.2f} ms  ({fl / ts / 1e9:6.1f} TFLOP/s)")
    print(f"  LayerNorm          {t(lambda: Fn.add_layer_norm(x, None, blk.norm1)):8.2f} ms")
    print(f"  qkv GEMM           {t(lambda: Fn.linear(x, blk.attn.qkv.weight, blk.attn.qkv.bias)):8.2f} ms")
    print(f"  proj GEMM          {t(lambda: Fn.linear(x, blk.attn.proj.weight, blk.attn.proj.bias)):8.2f} ms")
    print(f"  fc1 GEMM           {t(lambda: Fn.linear(x, blk.mlp.fc1.weight, blk.mlp.fc1.bias)):8.2f} ms")
    h = Fn.linear(x, blk.mlp.fc1.weight, blk.mlp.fc1.bias)
    print(f"  GELU               {t(lambda: F.gelu(h)):8.2f} ms")
    print(f"  fc2 GEMM           {t(lambda: Fn.linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias)):8.2f} ms")
    print(f"  residual add       {t(lambda: x + x):8.2f} ms")
