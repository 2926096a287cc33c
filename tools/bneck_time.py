"""Time the res2 stage of the R50 at the benchmark shape (T frames of 184 x 320 after the stem): the chain of csrc/bneck_x3.hip
against the layer-by-layer path (DVIS_X3_BNECK=0), per launch with HIP events.

    python tools/bneck_time.py [T] [H W]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd import functions as Fn      # noqa: E402
from dvis_plus_amd.backbone import build_resnet50      # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (184, 320)
    dev = "cuda:0"
    torch.manual_seed(0)
    m = build_resnet50().to(dev).eval()
    x = torch.randn(T, 64, H, W, device=dev).relu()
    with torch.no_grad():
        blocks = m._chain_blocks(m.res2)
        assert Fn.bneck_stage_x3_ok(x, blocks)
        ref = m.res2(x)
        got = Fn.bneck_stage_x3(x, blocks)
        print(f"T = {T}, {H} x {W}: max |chain - layer by layer| = {float((got - ref).abs().max()):.3e} (max |y| {float(ref.abs().max()):.2f})")
        t_chain = timed(lambda: Fn.bneck_stage_x3(x, blocks))
        t_layers = timed(lambda: m.res2(x))
        px = T * H * W
        gb_chain = px * 4 * (64 + 64 + 3 * (64 * 1.2 + 256 + 256 + 64) - 64) / 1e9
        print(f"res2 stage: chain {t_chain:.3f} ms ({gb_chain / t_chain:.0f} GB/s over ~{gb_chain:.1f} GB), layer by layer {t_layers:.3f} ms")
        # per launch
        lib = Fn.native.lib()
        for i in range(3):
            sub = blocks[:i + 1] if i > 0 else None
        # the chain's launches one by one: first conv1 -> image, then each block
        import ctypes
        img = torch.empty(lib.dvis_bneck_x3_image_bytes(T, H, W), dtype=torch.uint8, device=dev)
        w = blocks[0]["w1"]
        t0 = timed(lambda: Fn.bneck_stage_x3(x, blocks[:2]))
        t1 = timed(lambda: Fn.bneck_stage_x3(x, blocks[:3]))
        print(f"two blocks {t0:.3f} ms, three blocks {t1:.3f} ms -> an identity block with a chained conv1 ~ {t1 - t0:.3f} ms")
        c1 = timed(lambda: Fn.conv1x1_x3(x, blocks[0]['w1'], blocks[0]['b1'], None, True))
        print(f"(first conv1 64 -> 64 as an fp32 map: {c1:.3f} ms)")


if __name__ == "__main__":
    main()
