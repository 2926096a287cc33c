"""One ViT-L-shaped long self-attention call on the split-f16 kernel (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dvis_plus_amd import functions as Fn
dev = torch.device("cuda:0")
L, B, H, d = 3681, 10, 16, 64
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(L, B, H * d, generator=g).to(dev) for _ in range(3))
with torch.no_grad():
    for _ in range(3):
        Fn.attention(q, k, v, H, None, None)
torch.cuda.synchronize()
