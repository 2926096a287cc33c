"""Are the split-f16 kernels bit-reproducible when several PROCESSES share the GPU (waves of different processes on the same
CUs, compute-wave save / restore)?  N processes loop over the kernels and compare every result with their first one.
    python tools/exp/x3_contention.py [procs] [iters]"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(iters):
    import torch.nn as nn
    from dvis_plus_amd import functions as Fn
    dev = "cuda:0"
    torch.manual_seed(0)
    with torch.no_grad():
        l1, l2, norm = nn.Linear(256, 1024).to(dev), nn.Linear(1024, 256).to(dev), nn.LayerNorm(256).to(dev)
        lin, lin288, kv = nn.Linear(256, 256).to(dev), nn.Linear(256, 288).to(dev), nn.Linear(256, 768).to(dev)
        x = torch.randn(4, 4830, 256, device=dev)
        res, pos = torch.randn_like(x), torch.randn(1, 4830, 256, device=dev)
        xc = torch.randn(4, 512, 23, 40, device=dev)
        w1 = torch.randn(128, 512, 1, 1, device=dev) * 0.05
        w3 = torch.randn(512, 512, 3, 3, device=dev) * 0.02
        xr = torch.randn(4, 128, 23, 40, device=dev)
        w2 = torch.randn(512, 128, 1, 1, device=dev) * 0.05
        r2 = torch.randn(4, 512, 23, 40, device=dev)
        cases = {
            "ffn": lambda: Fn.x3_ffn_ln(x, l1, l2, norm, pos=pos)[1],
            "linear_ln": lambda: Fn.x3_linear_ln(x, lin.weight, lin.bias, res, norm),
            "linear288+pos": lambda: Fn.x3_linear(x, lin288.weight, lin288.bias, xadd=pos),
            "linear768": lambda: Fn.x3_linear(x, kv.weight, kv.bias),
            "conv1x1": lambda: Fn.conv1x1_x3(xc, w1, None, None, True, 1),
            "conv1x1+res": lambda: Fn.conv1x1_x3(xr, w2, None, r2, True, 1),
            "conv3x3": lambda: Fn.conv3x3_x3(xc, w3, None, None, True, 1),
            "conv3x3s2": lambda: Fn.conv3x3_x3(xc, w3, None, None, True, 2),
        }
        first = {k: f().clone() for k, f in cases.items()}
        bad = {k: 0 for k in cases}
        for _ in range(iters):
            for k, f in cases.items():
                if not torch.equal(f(), first[k]):
                    bad[k] += 1
        torch.cuda.synchronize()
    print("pid", os.getpid(), "mismatches per kernel:", bad, flush=True)
    return sum(bad.values())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        sys.exit(1 if worker(int(sys.argv[2])) else 0)
    procs, iters = (int(sys.argv[1]) if len(sys.argv) > 1 else 3), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", str(iters)]) for _ in range(procs)]
    rc = [p.wait() for p in ps]
    print("CONTENTION", "OK" if not any(rc) else "MISMATCH", rc)
