"""A few launches of the ViT-L qkv projection + split-f16 self-attention for a rocprofv3 --pmc pass (tools/exp: development)."""
import torch
from dvis_plus_amd import functions as Fn

dev = "cuda:0"
torch.manual_seed(0)
with torch.no_grad():
    x = torch.randn(30, 3681, 1024, device=dev)
    wq = torch.randn(3072, 1024, device=dev) / 32
    bq = torch.randn(3072, device=dev) * 0.1
    img = Fn.x3_rows_image(x)
    for _ in range(3):
        Fn.x3_qkv_attention(img, wq, bq, 16, out_image=True)
    torch.cuda.synchronize()
