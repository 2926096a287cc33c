"""Library GEMM time of the decoder's memory K / V projections (tokens of one level of 30 frames, 256 -> n_layers * 256)."""
import torch, torch.nn.functional as F
dev = "cuda:0"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
with torch.no_grad():
    for hw in (14720, 3680, 920):
        tok = torch.randn(30, hw, 256, device=dev)
        for n in (768, 1024, 256):
            w = torch.randn(n, 256, device=dev); b = torch.randn(n, device=dev)
            us = t(lambda: F.linear(tok, w, b))
            us2 = t(lambda: F.linear(tok.view(-1, 256), w, b))
            print(f"hw={hw:6d} N={n:5d}: 3-D input {us:8.1f} us  2-D {us2:8.1f} us  ({2.0 * 30 * hw * 256 * n / us2 / 1e6:6.1f} TF/s)", flush=True)
