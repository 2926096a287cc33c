"""Root-cause probe for the >= 4 GiB activation problem: does a library / own kernel of the deformable
encoder's FFN compute wrong values once a tensor crosses 2^32 bytes?  Single stream, bounded memory, no concurrency: safe.
For n = 56 frames of 720p tokens the FFN hidden tensor (n * 19320, 1024) fp32 is 4.43 GB.  Every op is run on the whole
batch and on two halves; rows are compared frame by frame."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dvis_plus_amd import functions as Fn  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
S, C, H = 19320, 256, 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 56
lin1, lin2 = torch.nn.Linear(C, H).to(dev), torch.nn.Linear(H, C).to(dev)
norm = torch.nn.LayerNorm(C).to(dev)
x = torch.randn(n, S, C, device=dev)


def per_frame(a, b, what):
    d = (a - b).abs().flatten(1).max(1)[0]
    bad = torch.nonzero(d > 1e-3).flatten().tolist()
    print(f"{what:58s} max|whole - halves| {float(d.max()):.3e}; frames off by > 1e-3: {bad[:8]}{' ...' if len(bad) > 8 else ''}"
          f" ({len(bad)} of {n})")


with torch.no_grad():
    half = n // 2
    for name, fn in (("relu(linear1) via hipBLASLt epilogue (_addmm_activation)", lambda t: Fn.linear_relu(t, lin1)),
                     ("linear1 (F.linear) + torch.relu", lambda t: torch.relu(F.linear(t, lin1.weight, lin1.bias)))):
        whole = fn(x)
        print(f"hidden tensor: {whole.numel() * 4 / 2 ** 30:.2f} GiB")
        parts = torch.cat([fn(x[:half]), fn(x[half:])], 0)
        per_frame(whole, parts, name)
        y_whole = F.linear(whole, lin2.weight, lin2.bias)
        y_parts = torch.cat([F.linear(parts[:half], lin2.weight, lin2.bias), F.linear(parts[half:], lin2.weight, lin2.bias)], 0)
        per_frame(y_whole, y_parts, "linear2 on that hidden tensor")
        del whole, parts
        torch.cuda.empty_cache()
    a = Fn.add_layer_norm(y_whole, x, norm)
    b = torch.cat([Fn.add_layer_norm(y_whole[:half].contiguous(), x[:half].contiguous(), norm),
                   Fn.add_layer_norm(y_whole[half:].contiguous(), x[half:].contiguous(), norm)], 0)
    per_frame(a, b, "dvis_add_layernorm (own kernel)")
    # an elementwise torch op across the 4 GiB line
    big = torch.randn(n, S, H, device=dev)
    r = torch.relu(big)
    per_frame(r, torch.cat([torch.relu(big[:half]), torch.relu(big[half:])], 0), "torch.relu on a 4.4 GB tensor")
print("done")
