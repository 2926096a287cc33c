"""Is phase A (backbone -> pixel decoder -> masked-attention decoder) FRAME-INVARIANT — does a frame get the same bits
whatever other frames share the call — and bit-reproducible run to run?  north_star's frame sharding relies on both (ranks
batch other frame counts; the replicated tracker must see identical queries: dvis_Plus/video_mask2former_transformer_decoder.py:327-335
folds frames into the batch).

Runs the segmenter on the first n frames of one seeded clip for every n in --frames; records every op / leaf-module output
of the n = first run, and for each later n compares frame 0's slice of every output against it (the op whose output is the
first to differ is the culprit).  Also runs every n --repeat times and compares fingerprints run to run.

    python tools/phase_a_invariance.py [--height 360 --width 640] [--frames 1,2,3,5] [--repeat 2] [--list-ops]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvis_plus_amd import functions as Fn  # noqa: E402
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--frames", default="1,2,3,5")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--list-ops", action="store_true")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    ns = [int(v) for v in args.frames.split(",")]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = build_dvis_plus_r50("offline", task="vps").to(dev).eval()
    g = torch.Generator().manual_seed(7)
    clip = torch.randint(0, 256, (max(ns), 3, args.height, args.width), generator=g, dtype=torch.uint8).to(dev)

    log = []          # (name, tensor-or-None, fingerprint, shape)
    state = {"keep": False, "n": 1, "base": None, "i": 0, "report": [], "seq_break": None}

    def fingerprint(t):
        return int(t.contiguous().view(torch.int32).to(torch.int64).sum())

    def frame0(t, shape1, n):
        """frame 0's part of an n-frame output, given the 1-frame (base n0) run's shape of the same output."""
        if tuple(t.shape) == tuple(shape1):
            return t
        if t.dim() != len(shape1):
            return None
        cand = [d for d in range(t.dim()) if t.shape[d] != shape1[d]]
        if len(cand) != 1:
            return None
        d = cand[0]
        if t.shape[d] * state["n0"] != shape1[d] * n:
            return None
        return t.narrow(d, 0, shape1[d] // state["n0"]) if state["n0"] == 1 else None

    def rec(name, out):
        if torch.is_tensor(out) and out.dtype == torch.float32 and out.numel():
            t = out.detach()
            i = state["i"]
            state["i"] += 1
            if state["keep"]:
                log.append((name, t.clone(), fingerprint(t), tuple(t.shape)))
            else:
                log.append((name, None, fingerprint(t), tuple(t.shape)))
                base = state["base"]
                if base is not None and state["seq_break"] is None:
                    if i >= len(base) or base[i][0] != name:
                        state["seq_break"] = (i, name, base[i][0] if i < len(base) else "<end>")
                        return
                    part = frame0(t, base[i][3], state["n"])
                    if part is None:
                        state["report"].append((i, name, "unsliceable", tuple(t.shape), base[i][3]))
                    elif part.shape == base[i][1].shape:
                        ref = base[i][1]
                        if not torch.equal(part, ref):
                            d = (part - ref).abs()
                            state["report"].append((i, name, "differs", float(d.max()), int((part != ref).sum()), part.numel()))
        elif isinstance(out, (tuple, list)):
            for k, o in enumerate(out):
                rec(f"{name}[{k}]", o)
        elif isinstance(out, dict):
            for k, o in out.items():
                rec(f"{name}[{k}]", o)

    for name, mod in m.named_modules():
        if not list(mod.children()):
            mod.register_forward_hook(lambda mod_, inp, out, name=name: rec("module " + name, out))
    import torch.nn.functional as F
    fn_names = [n for n in dir(Fn) if callable(getattr(Fn, n)) and not n.startswith("_") and getattr(getattr(Fn, n), "__module__", "")
                == Fn.__name__ and isinstance(getattr(Fn, n), type(main))]
    skip = {"x3_ok", "x3_ffn_ok", "conv1x1_x3_ok", "conv3x3_x3_ok", "gemm_ln_ok", "conv1x1s2_supported", "normalize_pad_ok",
            "x3_pack"}
    for ns_, names in ((Fn, [n for n in fn_names if n not in skip]), (F, ["linear", "conv2d", "interpolate", "group_norm", "layer_norm"]),
                       (torch, ["bmm", "_addmm_activation", "matmul"])):
        for n in names:
            orig = getattr(ns_, n)

            def w(*a, _orig=orig, _n=f"{ns_.__name__.split('.')[-1]}.{n}", **k):
                out = _orig(*a, **k)
                rec("op " + _n, out)
                return out
            setattr(ns_, n, w)

    def run(n):
        log.clear()
        state["i"] = 0
        with torch.no_grad():
            images, _ = m.preprocess(clip[:n])
            out = m.segment(images)
            rec("segment", out)
        torch.cuda.synchronize()

    tag = args.tag
    n0 = ns[0]
    state["n0"] = n0
    state["keep"], state["n"] = True, n0
    run(n0)
    base = list(log)
    if args.list_ops:
        for i, (name, _, _, shape) in enumerate(base):
            print(i, name, shape)
    state["keep"] = False
    ok = True
    for n in ns:
        fps = []
        for r in range(args.repeat):
            state["n"], state["base"], state["report"], state["seq_break"] = n, (base if r == 0 else None), [], None
            run(n)
            fps.append([(e[0], e[2]) for e in log])
            if r == 0:
                rep, brk = state["report"], state["seq_break"]
                diffs = [x for x in rep if x[2] == "differs"]
                uns = [x for x in rep if x[2] == "unsliceable"]
                print(f"{tag}n={n}: {len(log)} outputs; vs the {n0}-frame run, frame 0: {len(diffs)} outputs differ, {len(uns)} unsliceable"
                      + (f"; op sequence changes at #{brk[0]}: {brk[1]} (was {brk[2]})" if brk else ""), flush=True)
                for x in diffs[:6]:
                    print(f"   #{x[0]} {x[1]}: max|d| {x[3]:.3e}, {x[4]} of {x[5]} elements")
                for x in uns[:4]:
                    print(f"   #{x[0]} {x[1]}: unsliceable {x[3]} vs {x[4]}")
                ok &= not diffs and not brk
        for r in range(1, args.repeat):
            bad = [(a[0], i) for i, (a, b) in enumerate(zip(fps[0], fps[r])) if a != b]
            print(f"{tag}n={n}: run 0 vs run {r}: {len(bad)} of {len(fps[0])} outputs differ bitwise" +
                  (f"; first: #{bad[0][1]} {bad[0][0]}" if bad else ""), flush=True)
            ok &= not bad
    print(f"{tag}PHASE_A_INVARIANT {'OK' if ok else 'FAILED'}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
