"""Minimal reproduction of round 6's root cause: does a hipMemsetAsync issued on a capturing stream become a node
that REPLAYS?  buf starts at 7; the captured work is `memset(buf, 0)` followed by `buf += 1`.  A replaying memset node gives 1 after
every replay; a memset that took effect at capture time and is absent from the graph gives 1, 2, 3, ...
    python tools/exp/memset_graph_repro.py        -> gpurun_out/r06/memset_graph_repro.txt (tools/exp/r06_final.sh)"""
import ctypes

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = torch.device("cuda", 0)
for nbytes in (2000, 4096, 1 << 20):
    n = nbytes // 4
    buf = torch.full((n,), 7, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        buf.add_(0)                              # warm-up of the torch kernel outside capture
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, nbytes, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        buf.add_(1)
    torch.cuda.synchronize()
    after_capture = int(buf[0])
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        seen.append((int(buf[0]), int(buf[-1])))
    verdict = "memset node replays" if seen == [(1, 1)] * 3 else "memset does NOT replay (values accumulate)" if seen[2][0] > seen[0][0] else "?"
    print(f"hipMemsetAsync of {nbytes} bytes under capture: rc {rc}; buf[0] after capture {after_capture}; after replays {seen} -> {verdict}")
print("torch", torch.__version__, "hip", torch.version.hip)
