"""Which stage of phase A (backbone -> pixel decoder -> decoder) is not bit-reproducible run to run?
Runs the segmenter twice on the same 720p frames with a hook on every leaf module + the functional ops of
dvis_plus_amd.functions, and reports the first outputs that differ (max |d|, count).
    python tools/determinism_probe.py [--frames 30] [--own]       (--own: DVIS_DETERMINISTIC=1 GEMMs)"""
import os
import sys

import torch

if "--own" in sys.argv:
    os.environ["DVIS_DETERMINISTIC"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dvis_plus_amd import functions as Fn  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402


def main():
    T = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 30
    dev = torch.device("cuda:0")
    m = build_dvis_plus_r50("offline", task="vps").to(dev)
    clip = bench.synthetic_clip(T, dev, seed=1234)
    log = []

    def rec(name, out):
        if torch.is_tensor(out) and out.dtype == torch.float32 and out.numel():
            t = out.detach().contiguous()
            # exact, order-independent fingerprint of the BITS (a clone of every activation would need ~100 GB)
            log.append((name, int(t.view(torch.int32).to(torch.int64).sum()), float(t.double().abs().sum()), t.numel()))
        elif isinstance(out, (tuple, list)):
            for i, o in enumerate(out):
                rec(f"{name}[{i}]", o)
        elif isinstance(out, dict):
            for k, o in out.items():
                rec(f"{name}[{k}]", o)

    for name, mod in m.named_modules():
        if not list(mod.children()):
            mod.register_forward_hook(lambda mod_, inp, out, name=name: rec("module " + name, out))
    import torch.nn.functional as F
    wrapped = {}
    for ns, names in ((Fn, ["linear", "conv1x1", "conv1x1_bias_act", "msda_fused_forward", "attention", "attn_mask",
                            "add_layer_norm", "maps_to_tokens", "upsample_add", "bias_act_", "group_norm_affine"]),
                      (F, ["linear", "conv2d"]), (torch, ["bmm", "_addmm_activation"])):
        for n in names:
            orig = getattr(ns, n)
            wrapped[(ns, n)] = orig

            def w(*a, _orig=orig, _n=f"{ns.__name__.split('.')[-1]}.{n}", **k):
                out = _orig(*a, **k)
                rec("op " + _n, out)
                return out
            setattr(ns, n, w)
    runs = []
    with torch.no_grad():
        for _ in range(2):
            log.clear()
            images, _ = m.preprocess(clip)
            m.segment(images)
            torch.cuda.synchronize()
            runs.append(list(log))
    a, b = runs
    assert len(a) == len(b)
    ndiff, first, per_name = 0, None, {}
    for i, ((n0, h0, s0, numel), (n1, h1, s1, _)) in enumerate(zip(a, b)):
        if h0 != h1:
            ndiff += 1
            key = n0.split("[")[0]
            per_name.setdefault(key, []).append(abs(s0 - s1) / max(s0, 1e-30))
            if first is None:
                first = (i, n0, abs(s0 - s1) / max(s0, 1e-30), numel)
    print(f"{len(a)} recorded outputs, {ndiff} differ (bitwise) between two runs of the segmenter on the same {T} frames"
          f" (own GEMMs: {Fn.OWN_GEMM_DEFAULT})")
    if first:
        print("first differing output: #%d %s, relative change of sum|x| %.3e, %d elements" % first)
        for k, v in list(per_name.items())[:40]:
            print(f"  {k}: {len(v)} outputs differ, largest relative change of sum|x| {max(v):.3e}")


if __name__ == "__main__":
    main()
