"""1x1 convolutions of the R50 backbone / pixel decoder: MIOpen conv2d vs the same contraction as a batched GEMM
(W (Co,Ci) @ X (N,Ci,HW), NCHW kept).  Dev tool."""
import torch
import torch.nn.functional as F

dev = "cuda:0"
N = 30


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


shapes = [  # (Ci, Co, H, W, tag)
    (64, 64, 184, 320, "res2.0.conv1"), (256, 64, 184, 320, "res2.x.conv1"), (64, 256, 184, 320, "res2.x.conv3"),
    (256, 128, 184, 320, "res3.0.conv1"), (512, 128, 92, 160, "res3.x.conv1"), (128, 512, 92, 160, "res3.x.conv3"),
    (512, 256, 92, 160, "res4.0.conv1"), (1024, 256, 46, 80, "res4.x.conv1"), (256, 1024, 46, 80, "res4.x.conv3"),
    (1024, 512, 46, 80, "res5.0.conv1"), (2048, 512, 23, 40, "res5.x.conv1"), (512, 2048, 23, 40, "res5.x.conv3"),
    (256, 256, 184, 320, "pd.mask_features / lateral"), (512, 256, 92, 160, "pd.input_proj res3"),
    (1024, 256, 46, 80, "pd.input_proj res4"), (2048, 256, 23, 40, "pd.input_proj res5"),
]
with torch.no_grad():
    for Ci, Co, H, W, tag in shapes:
        x = torch.randn(N, Ci, H, W, device=dev)
        w = torch.randn(Co, Ci, 1, 1, device=dev) * 0.05
        w2 = w.view(Co, Ci)
        a = F.conv2d(x, w)
        b = torch.matmul(w2, x.view(N, Ci, H * W)).view(N, Co, H, W)
        err = (a - b).abs().max().item()
        out = torch.empty(N, Co, H * W, device=dev)
        tc = t(lambda: F.conv2d(x, w))
        tm = t(lambda: torch.matmul(w2, x.view(N, Ci, H * W)))
        tb = t(lambda: torch.bmm(w2.expand(N, Co, Ci), x.view(N, Ci, H * W), out=out))
        gf = 2.0 * N * Ci * Co * H * W / 1e9
        gb = 4.0 * N * (Ci + Co) * H * W / 1e9
        print(f"{tag:28s} Ci={Ci:4d} Co={Co:4d} {H}x{W}: conv {tc:6.3f} ms  matmul {tm:6.3f} ms  bmm {tb:6.3f} ms  "
              f"({gf / min(tc, tm, tb):6.1f} TF/s best, {gb / min(tc, tm, tb):5.2f} TB/s)  maxdiff {err:.1e}", flush=True)
