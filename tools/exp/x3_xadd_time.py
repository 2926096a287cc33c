"""(x + pos) W^T: torch add pass + streaming projection kernel vs the position added inside the resident-fragment kernel."""
import sys, torch, torch.nn as nn
sys.path.insert(0, ".")
from dvis_plus_amd import functions as Fn
DEV = "cuda:0"
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
with torch.no_grad():
    for N, S in ((768, 14720), (768, 3680), (288, 19320)):
        lin = nn.Linear(256, N).to(DEV)
        x = torch.randn(30, S, 256, device=DEV); pos = torch.randn(1, S, 256, device=DEV)
        a = timeit(lambda: Fn.x3_linear(x + pos, lin.weight, lin.bias)); b = timeit(lambda: Fn.x3_linear(x, lin.weight, lin.bias))
        c = timeit(lambda: Fn.x3_linear(x, lin.weight, lin.bias, xadd=pos))
        print(f"N={N} S={S}: add pass + projection {a:.3f} ms; projection alone {b:.3f}; position added in the kernel {c:.3f}")
