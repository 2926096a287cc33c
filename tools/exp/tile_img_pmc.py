"""A few launches of the ViT-L fc1 / fc2 row-image GEMMs for a rocprofv3 --pmc pass (tools/exp: development)."""
import torch
from dvis_plus_amd import functions as Fn

dev = "cuda:0"
torch.manual_seed(0)
with torch.no_grad():
    M = 30 * 3681
    x = torch.randn(M, 1024, device=dev)
    w1 = torch.randn(4096, 1024, device=dev) / 32
    b1 = torch.randn(4096, device=dev) * 0.1
    w2 = torch.randn(1024, 4096, device=dev) / 64
    img = Fn.x3_rows_image(x)
    for _ in range(3):
        hid = Fn.x3_tile_linear(img, w1, b1, act="gelu")
        y = Fn.x3_tile_linear(hid, w2, None, residual=x)
        y0 = Fn.x3_tile_linear(x, w1, b1, act="gelu")
    torch.cuda.synchronize()
