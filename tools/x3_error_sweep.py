"""Who spends BASELINE.json's literal 1e-3 on the mask logits?  The benchmarked configuration (#3: offline R50, T = 30,
720p, 100 queries, mask heads at a natural logit scale — tests/test_pipeline_720p_gpu.py::test_T30_natural_logit_scale…)
over several seeds (clips), product vs the CPU oracle, with the split-f16 arithmetic (csrc/gemm_x3.hip,
csrc/conv1x1_x3.hip) switched off stage by stage (Fn.X3_OFF: pd_proj, mask_path, encoder, decoder_kv).  The oracle is fed the
product's backbone outputs (pipeline_parity.gpu_backbone), so one oracle run per seed serves every configuration as long as
the backbone's own arithmetic stays the same — it does: `backbone` is never in the swept sets.

    python tools/x3_error_sweep.py [--seeds 1234,1,2,3,4] [--frames 30] [--out gpurun_out/r05/x3_error_sweep.txt]

Per (seed, configuration): max |product - oracle| over ALL 100 queries' final mask logits, the decoder's differing attention
mask bits, the per-frame query error — and ms per clip of the configuration (steady state, 3 clips through stream()).
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = [
    ("x3 everywhere (default)", ""),
    ("exact: mask_path", "mask_path"),
    ("exact: pd_proj", "pd_proj"),
    ("exact: encoder", "encoder"),
    ("exact: decoder_kv", "decoder_kv"),
    ("exact: mask_path + decoder_kv", "mask_path,decoder_kv"),
    ("exact: mask_path + encoder", "mask_path,encoder"),
    ("exact: all four (x3 only in the backbone)", "pd_proj,mask_path,encoder,decoder_kv"),
    ("x3 everywhere, attention masks WITHOUT the feature pyramid (dvis_attn_mask)", "nopyramid"),
    ("exact: all four, attention masks WITHOUT the feature pyramid", "pd_proj,mask_path,encoder,decoder_kv,nopyramid"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1234,1,2,3,4")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--gain", type=float, default=2.0)
    ap.add_argument("--out", default="gpurun_out/r05/x3_error_sweep.txt")
    ap.add_argument("--configs", default="")          # indices, e.g. 0,1,7
    ap.add_argument("--no-timing", action="store_true")
    args = ap.parse_args()
    import bench
    import pipeline_parity as PPar
    from dvis_plus_amd import functions as Fn
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    dev = torch.device("cuda:0")
    m = build_dvis_plus_r50("offline", task="vps", object_mask_threshold=0.0)
    PPar.perturb_msda(m.sem_seg_head.pixel_decoder)
    PPar.sharpen_masks(m, args.gain)
    sd = PPar.cpu_state(m)
    m = m.to(dev)
    m.overlap_threshold = 0.0
    configs = CONFIGS if not args.configs else [CONFIGS[int(i)] for i in args.configs.split(",")]
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")

    say(f"# x3 error sweep: offline R50, T={args.frames}, 720p, mask heads x{args.gain:g}; final mask logits of all 100 queries, product vs oracle "
        f"(oracle fed the product's backbone outputs); DVIS_X3={os.environ.get('DVIS_X3', '1')}")
    if not args.no_timing:
        clip = bench.synthetic_clip(args.frames, dev, seed=99)
        video = {"image": clip, "height": 720, "width": 1280}
        m.object_mask_threshold = bench.calibrate_threshold(m, [video], 20)
        for name, off in configs:
            Fn.X3_OFF = frozenset(v for v in off.split(",") if v and v != "nopyramid")
            Fn.ATTN_MASK_PYRAMID = "nopyramid" not in off
            for o in m.stream([video] * 2):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for o in m.stream([video] * 6):
                pass
            torch.cuda.synchronize()
            say(f"timing | {name}: {(time.perf_counter() - t0) / 6 * 1e3:.2f} ms per clip (6 clips through stream())")
    worst = {name: 0.0 for name, _ in configs}
    for seed in [int(v) for v in args.seeds.split(",")]:
        clip = bench.synthetic_clip(args.frames, dev, seed=seed)
        video = {"image": clip, "height": 720, "width": 1280}
        Fn.X3_OFF, Fn.ATTN_MASK_PYRAMID = frozenset(), True
        m.object_mask_threshold = bench.calibrate_threshold(m, [video], 20)
        t0 = time.perf_counter()
        ref, stages = PPar.run_oracle(m, sd, [f for f in clip.cpu()], offline=True, task="vps", attn_masks=True,
                                      object_mask_threshold=m.object_mask_threshold, overlap_threshold=0.0, out_hw=(720, 1280))
        say(f"seed {seed}: oracle {time.perf_counter() - t0:.0f} s; max |oracle logit| {float(stages['masks'].abs().max()):.2f}")
        for name, off in configs:
            Fn.X3_OFF = frozenset(v for v in off.split(",") if v and v != "nopyramid")
            Fn.ATTN_MASK_PYRAMID = "nopyramid" not in off
            m.debug_stages = {}
            m.sem_seg_head.predictor.debug_masks = []
            m([video])
            pmasks, m.sem_seg_head.predictor.debug_masks = m.sem_seg_head.predictor.debug_masks, None
            with torch.no_grad():
                logits = m.debug_stages["mask_fn"](None).float().cpu()
            err = float((logits - stages["masks"].float()).abs().max())
            flips, bits = PPar.attention_mask_flips(pmasks, stages["attn_masks"], args.frames)
            fe = (m.debug_stages["frame_embds_no_norm"].float().cpu() - stages["frame_embds_no_norm"]).abs().amax((0, 1, 3))
            mf = float("nan")
            if "mask_features" in m.debug_stages and stages.get("mask_features") is not None:
                b = stages["mask_features"]
                mf = float((m.debug_stages["mask_features"].float().cpu() - (b[0] if b.dim() == 5 else b)).abs().max())
            worst[name] = max(worst[name], err)
            say(f"seed {seed} | {name}: mask-logit error {err:.3e}; attention-mask bits differing {int(flips.sum())} "
                f"(frames touched {int((flips.sum((0, 2)) > 0).sum())}/{args.frames}); decoder query error max {float(fe.max()):.2e}, "
                f"median frame {float(fe.median()):.2e}; mask_features error {mf:.2e}")
    Fn.X3_OFF, Fn.ATTN_MASK_PYRAMID = frozenset(), True
    say("# worst over seeds")
    for name, _ in configs:
        say(f"worst | {name}: {worst[name]:.3e}")


if __name__ == "__main__":
    main()
