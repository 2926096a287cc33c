"""Per-call time of every convolution of the segmenter's front-end (R50 + pixel-decoder convs) on the headline clip:
F.conv2d / functions.conv1x1 / functions.bias_act_ / max_pool2d are wrapped with HIP events (synchronous: a development
aid, not a benchmark)."""
import collections
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd import functions as Fn                                    # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50             # noqa: E402

rows = collections.OrderedDict()


def timed(name, fn, key):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        kk = (name,) + key(*a, **k)
        r = rows.setdefault(kk, [0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
        return out
    return wrapper


def main():
    dev = torch.device("cuda", 0)
    model = build_dvis_plus_r50("offline", task="vps").to(dev).eval()
    x = torch.rand(30, 3, 720, 1280, device=dev) * 255
    conv2d, conv1x1, bias_act, pool = F.conv2d, Fn.conv1x1, Fn.bias_act_, F.max_pool2d
    with torch.no_grad():
        images, _ = model.preprocess(x)
        model.encode(images)                                               # warm-up (algorithm search)
        F.conv2d = timed("conv2d", conv2d, lambda x, w, b=None, s=1, p=0, *a, **k: (tuple(x.shape), tuple(w.shape), str(s), str(p)))
        Fn.conv1x1 = timed("conv1x1", conv1x1, lambda x, w, *a, **k: (tuple(x.shape), tuple(w.shape)))
        Fn.bias_act_ = timed("bias_act", bias_act, lambda x, *a, **k: (tuple(x.shape),))
        F.max_pool2d = timed("max_pool", pool, lambda x, *a, **k: (tuple(x.shape),))
        for _ in range(3):
            model.encode(images)
    tot = collections.Counter()
    for k, (n, ms) in rows.items():
        print(f"{ms / 3:8.3f} ms/clip  {n // 3:3d} calls  {ms / n * 1e3:9.1f} us  {k}")
        tot[k[0]] += ms / 3
    print(dict(tot))


if __name__ == "__main__":
    main()
