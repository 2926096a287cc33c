"""GPU parity at BASELINE.json's own configurations (720p frames, full R50 configuration: hidden 256, 100 queries,
6 encoder / 6 tracker / 6 refiner layers, 9 decoder layers), product vs the CPU oracle's windowed pipeline:

  config #2  DVIS++ online,  T=5,  VPS and VIS   (dvis_Plus/meta_architecture.py:591-706, 774-816; masks from the
                                                  tracker: tracker.py:368-380 incl. mask_feature_proj)
  config #3  DVIS++ offline, T=30, VPS, through stream() — THE BENCHMARKED WORKLOAD: bench.synthetic_clip, the bench's
             threshold calibration, 20 panoptic candidates; 30-frame tracker recurrence + refiner over T=30 + fused
             panoptic post-processing at 720p (meta_architecture.py:1446-1500, :869-952).

The oracle needs ~0.1 frames/s of CPU time (the T=30 case takes minutes); these are the slowest tests of the suite.
"""
import os
import sys

import pytest
import torch

import pipeline_parity as PPar

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _model(mode, task, gain=PPar.SHARPEN, **kw):
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    m = build_dvis_plus_r50(mode, task=task, object_mask_threshold=0.0, **kw)
    PPar.perturb_msda(m.sem_seg_head.pixel_decoder)
    PPar.sharpen_masks(m, gain)
    return m, PPar.cpu_state(m)


@pytest.mark.parametrize("task", ["vps", "vis"])
def test_online_T5_720p_vs_oracle(task):
    import bench
    m, sd = _model("online", task, max_num=10)
    m = m.to(DEV)
    clip = bench.synthetic_clip(5, torch.device(DEV), seed=1234)
    video = {"image": clip, "height": 720, "width": 1280}
    if task == "vps":
        m.object_mask_threshold = bench.calibrate_threshold(m, [video], 20)
        # random masks overlap heavily: with the reference's 0.8 overlap rule few segments survive; with the rule off
        # every candidate that wins a pixel becomes a segment, i.e. the whole arg-max map is compared
        m.overlap_threshold = 0.0
    m.debug_stages = {}
    out = m([video])
    ref, stages = PPar.run_oracle(m, sd, [f for f in clip.cpu()], offline=False, task=task, max_num=10,
                                  object_mask_threshold=m.object_mask_threshold, overlap_threshold=m.overlap_threshold,
                                  out_hw=(720, 1280))
    what = f"config #2 online {task} T=5 720p"
    ids = stages["vps_query_ids"].tolist() if task == "vps" else sorted(set(ref[2].tolist()))
    err, scale = PPar.measured_logit_error(m.debug_stages, stages["masks"], ids, what)
    tol = PPar.logit_tolerance(scale)
    assert err <= tol
    if task == "vps":
        assert out["pred_masks"].shape == (5, 720, 1280) and out["num_candidates"] == 20 and len(ref[1]) > 0
        PPar.compare_vps(out, ref, stages, what, tol_logit=tol)
    else:
        assert out["pred_masks"].shape == (10, 5, 720, 1280)
        PPar.compare_vis(out, ref, stages, what, tol=tol)


def test_bench_workload_vps_stream_vs_oracle():
    """The benchmark's workload (bench.py clips, calibrated candidate count, stream(), T = 30) against the oracle.  (Round 5 ran it
    at T = 12 for the suite's time limit; round 6 got the time back: the oracle's windows run on several host threads at the thread
    count the CPU ops actually scale to — profiles/r06_oracle_threads.txt.  DVIS_TEST_STREAM_T overrides.)"""
    import bench
    T = int(os.environ.get("DVIS_TEST_STREAM_T", "30"))
    m, sd = _model("offline", "vps")
    m = m.to(DEV)
    dev = torch.device(DEV)
    clips = [bench.synthetic_clip(T, dev, seed=1234 + i) for i in range(2)]
    videos = [{"image": c, "height": 720, "width": 1280} for c in clips]
    m.object_mask_threshold = bench.calibrate_threshold(m, videos[:1], 20)
    outs = []
    for out in m.stream(videos):                       # consumed on the current stream, no device-wide synchronize
        outs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()})
    assert outs[0]["num_candidates"] == 20 and outs[0]["pred_masks"].shape == (T, 720, 1280)
    # clip 0 (the calibrated one) against the oracle; clip 1 keeps stream()'s overlap honest: it must equal forward()
    from oracle import dvis_torch as O
    ref, stages = PPar.run_oracle(m, sd, [f for f in clips[0].cpu()], offline=True, task="vps",
                                  object_mask_threshold=m.object_mask_threshold, out_hw=(720, 1280))
    what = f"config #3 offline vps T={T} 720p through stream() (bench workload)"
    tol = PPar.logit_tolerance(float(stages["masks"][stages["vps_query_ids"]].abs().max()))
    PPar.compare_vps(outs[0], ref, stages, what, tol_logit=tol)
    # Round 3: the pipeline is bit-reproducible (phase A's library kernels measured reproducible run to run —
    # tools/determinism_probe.py — and phase B runs on the own deterministic GEMM): stream() next to another clip's
    # segmenter == forward() alone, EXACTLY.  (Rounds 1-2: 1.8 - 2.0 k pixels differed, excused as library noise.)
    again = m([videos[1]])
    assert again["segments_infos"] == outs[1]["segments_infos"] and again["pred_ids"] == outs[1]["pred_ids"]
    PPar.intcmp.exact(outs[1]["pred_masks"], again["pred_masks"],
                      "config #3 clip 1: stream() vs forward() of the same clip, panoptic map")
    # Random masks overlap heavily, so the reference's 0.8 overlap rule keeps few segments.  Second comparison on the
    # same clip with the overlap rule off: every candidate that wins a pixel becomes a segment, i.e. the whole
    # T x 720 x 1280 arg-max map is compared.  (The oracle's post-processing is re-run on its stored class
    # logits and masks; the product re-runs the clip.)
    m.overlap_threshold = 0.0
    m.debug_stages = {}
    out0 = m([videos[0]])
    err, scale = PPar.measured_logit_error(m.debug_stages, stages["masks"], stages["vps_query_ids"].tolist(), what)
    assert err <= PPar.logit_tolerance(scale)
    diag = {}
    with torch.no_grad():
        ref0 = O.inference_video_vps(stages["cls"], stages["masks"], (720, 1280), (720, 1280), (736, 1280), 124, 58,
                                     m.object_mask_threshold, 0.0, stages["aux"], diag=diag)
    assert len(ref0[1]) >= 1       # (random class heads put every candidate in one stuff class: the segments merge)
    PPar.compare_vps(out0, ref0, diag, f"config #3 offline vps T={T} 720p, overlap rule off (full arg-max map)",
                     tol_logit=tol)


def test_t64_clips_streamed_equal_clip_by_clip():
    """BASELINE config #4's clip length on one GPU: T = 64 at 720p through stream(), one 64-frame segmenter call per clip
    (4.7 GiB FFN activation) with phase B on the second stream — the schedule that stalled in rounds 1-2 and was capped
    (DESIGN.md section 9); phase B no longer issues library GEMMs, the caps are gone (tests/test_stream_gpu.py soaks it).
    Streamed == clip by clip, bit for bit."""
    import bench
    from dvis_plus_amd.meta_architecture import segmenter_frames_per_call
    assert segmenter_frames_per_call(64, 736, 1280) == 64
    m, _ = _model("offline", "vps")
    m = m.to(DEV)
    dev = torch.device(DEV)
    videos = [{"image": bench.synthetic_clip(64, dev, seed=77 + i), "height": 720, "width": 1280} for i in range(2)]
    m.object_mask_threshold = bench.calibrate_threshold(m, videos[:1], 20)
    got = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in m.stream(videos)]
    for o, v in zip(got, videos):
        want = m([v])
        assert o["pred_masks"].shape == (64, 720, 1280)
        assert o["segments_infos"] == want["segments_infos"] and o["pred_ids"] == want["pred_ids"]
        PPar.intcmp.exact(o["pred_masks"], want["pred_masks"],
                          "config #4 clip length (T=64) streamed vs clip by clip, panoptic map")


def test_T30_natural_logit_scale_literal_1e3_and_error_budget():
    """BASELINE.json: "mask logits within 1e-3 of reference" — asserted LITERALLY on the benchmarked configuration (#3,
    T = 30, 720p, full R50 sizes) at a natural logit scale: mask heads scaled so that max |logit| is 4 - 10 (trained
    DVIS++ masks live there; the x40 `sharpen_masks` of the other tests pushes |logit| beyond 100 to make every arg-max
    decisive, which multiplies the absolute error by the same factor).  Also records the per-stage error budget
    (encoder -> decoder -> 30-frame tracker recurrence -> refiner -> mask embeddings -> logits)."""
    import bench
    # max |logit| ~ 4.6.  The error is a sum of discrete events (attention-mask bits that differ where a down-sized logit sits
    # within rounding distance of 0).  Round 5 swept it: 5 clips x 8 arithmetic configurations (profiles/r05_x3_error_sweep.txt) —
    # 1.1e-4 ... 2.0e-4 with the split-f16 kernels everywhere, 1.0e-4 ... 2.2e-4 with the exact-fp32 kernels in all four stages
    # after the backbone: the split arithmetic does not spend the budget, the summation-order difference to the CPU oracle does,
    # equally for both.  Asserted at 5e-4 = 2.5x the worst of the sweep, half of BASELINE's literal 1e-3.
    gain = 2.0
    m, sd = _model("offline", "vps", gain=gain)
    m = m.to(DEV)
    clip = bench.synthetic_clip(30, torch.device(DEV), seed=1234)
    video = {"image": clip, "height": 720, "width": 1280}
    m.object_mask_threshold = bench.calibrate_threshold(m, [video], 20)
    m.overlap_threshold = 0.0
    m.debug_stages = {}
    m.sem_seg_head.predictor.debug_masks = []
    out = m([video])
    pmasks, m.sem_seg_head.predictor.debug_masks = m.sem_seg_head.predictor.debug_masks, None
    ref, stages = PPar.run_oracle(m, sd, [f for f in clip.cpu()], offline=True, task="vps", attn_masks=True,
                                  object_mask_threshold=m.object_mask_threshold, overlap_threshold=0.0, out_hw=(720, 1280))
    what = f"config #3 offline vps T=30 720p, mask heads x{gain:g} (natural logit scale)"
    with torch.no_grad():
        all_logits = m.debug_stages["mask_fn"](None)                                    # (Q, T, h, w), all 100 queries
    rows = PPar.error_budget(m.debug_stages, stages, all_logits, what)
    err, scale = rows["mask_logits"]
    assert 2.0 <= scale <= 16.0, f"the test's premise: natural logit scale (got max |logit| {scale:.1f})"
    assert err <= 5e-4, f"mask logits: max |product - oracle| {err:.3e} exceeds 5e-4 (BASELINE's literal bar: 1e-3) at max |logit| {scale:.1f}"
    # Where the budget's largest entry (the decoder's per-frame queries) comes from: not rounding, but BOOLEAN attention-mask
    # bits that differ where a down-sized mask logit sits within rounding distance of 0 — that query then attends to a
    # different key set in that layer.  Frames without a single differing bit must agree to rounding level.
    flips, _ = PPar.attention_mask_flips(pmasks, stages["attn_masks"], 30)
    per_frame_flips = flips.sum((0, 2))                                                 # (T,)
    fe_err = (m.debug_stages["frame_embds_no_norm"].float().cpu() - stages["frame_embds_no_norm"]).abs().amax((0, 1, 3))
    clean = per_frame_flips == 0
    PPar.intcmp._report(f"{what}: {int(clean.sum())} of 30 frames have identical attention masks in all 9 layers; decoder "
                        f"query error on those frames {float(fe_err[clean].max()) if clean.any() else float('nan'):.2e}, "
                        f"on frames with differing bits {float(fe_err[~clean].max()) if (~clean).any() else 0.0:.2e}")
    if clean.any():
        assert float(fe_err[clean].max()) <= 1e-4
    assert rows["frame_embds"][0] <= 5e-2 and rows["instance_embds"][0] <= 5e-2 and rows["refiner_embds"][0] <= 1e-2
    PPar.compare_vps(out, ref, stages, what, tol_logit=PPar.TOL_LOGIT)


def test_config4_T64_vs_oracle_prefix_suffix_and_refiner():
    """BASELINE config #4's clip (T = 64, 720p, full R50 sizes) on one GPU against the ORACLE, not against itself.  The
    oracle's windowed CPU pipeline over 64 frames would take ~10 minutes, so the comparison uses the structure of the
    path: (1) PREFIX — the segmenter is per-frame and the tracker causal, so frames 0..5 of the 64-frame run must equal
    the oracle's run on those 6 frames (2 reference windows); (2) SUFFIX — the oracle's tracker, given the product's
    state after frame 60 (last_outputs / last_frame_embeds / last_reference, tracker.py:175-185), run on the last
    reference window (frames 61, 62, 63) with `resume`, must reproduce the product's instance embeddings / logits /
    assignment indices there: 60 frames of recurrence have not drifted; (3) REFINER — the oracle's refiner over all 64
    frames, fed the product's tracker outputs, must reproduce the product's refined embeddings, mask embeddings and class
    logits (the refiner attends over the whole clip: no prefix property)."""
    import bench
    import numpy as np
    from oracle import dvis_torch as O
    m, sd = _model("offline", "vps", gain=3.0)
    m = m.to(DEV)
    T = 64
    clip = bench.synthetic_clip(T, torch.device(DEV), seed=4242)
    video = {"image": clip, "height": 720, "width": 1280}
    m.object_mask_threshold = bench.calibrate_threshold(m, [video], 20)
    m.debug_stages = {}
    m.sem_seg_head.predictor.debug_masks = []
    out = m([video])
    pmasks, m.sem_seg_head.predictor.debug_masks = m.sem_seg_head.predictor.debug_masks, None
    assert out["pred_masks"].shape == (T, 720, 1280)
    P = {k: (v.detach().float().cpu() if torch.is_tensor(v) else v) for k, v in m.debug_stages.items() if k != "mask_fn"}
    idx_prod = np.array(m.tracker.last_indices)
    rep = PPar.intcmp._report
    # ---- (1) prefix.  The decoder's attention masks are booleans: where a down-sized mask logit sits within the two
    # pipelines' rounding distance of 0 the bit differs and that query attends to a different key set — a
    # discrete event, O(1e-2) on the query.  Frames whose masks agree in all 9 layers must agree to rounding level; with a
    # differing bit somewhere the bound is the loose one, and the tracker (it consumes the queries) inherits it.
    ref, st = PPar.run_oracle(m, sd, [f for f in clip[:6].cpu()], offline=True, task="vps", attn_masks=True,
                              object_mask_threshold=m.object_mask_threshold, out_hw=(720, 1280))
    flips, _ = PPar.attention_mask_flips([pm[:6] for pm in pmasks], st["attn_masks"], 6)
    clean = flips.sum((0, 2)) == 0                                                       # (6,) frames without a flip
    for key in ("frame_embds", "frame_embds_no_norm"):
        per_frame = (P[key][:, :, :6] - st[key]).abs().amax((0, 1, 3))
        rep(f"config #4 T=64 prefix (frames 0-5) {key}: max |product - oracle| per frame "
            f"{[float(f'{e:.1e}') for e in per_frame.tolist()]} (frames without a differing mask bit: {clean.tolist()})")
        assert float(per_frame.max()) <= 5e-2
        if clean.any():
            assert float(per_frame[clean].max()) <= 1e-4
    tight = 1e-3 if bool(clean.all()) else 5e-2
    e = float((P["instance_embds"][:, :, :6] - st["instance_embds"]).abs().max())
    rep(f"config #4 T=64 prefix (frames 0-5) instance_embds: max |product - oracle| {e:.2e} at max |value| "
        f"{float(st['instance_embds'].abs().max()):.1f} (bound {tight:g})")
    assert e <= tight
    e = float((P["online_logits"][:, :6] - st["online_logits"]).abs().max())
    rep(f"config #4 T=64 prefix tracker class logits: max |product - oracle| {e:.2e} (bound {tight:g})")
    assert e <= tight
    e = float((P["mask_features"][:6] - st["mask_features"][0]).abs().max())
    rep(f"config #4 T=64 prefix mask_features: max |product - oracle| {e:.2e} at max |value| {float(st['mask_features'].abs().max()):.1f}")
    assert e <= 1e-4
    # ---- (2) suffix: frames 61..63 = the reference's last full window before the ragged end (64 = 21 * 3 + 1)
    t0 = 61
    trk = O.Tracker(O._sub(sd, "tracker."), 8, 6)
    inst, fe = P["instance_embds"], P["frame_embds"]                       # (1, C, T, Q)
    prev = t0 - 1
    trk.last_outputs = inst[0, :, prev].t()[None, :, None, :]              # (1 layer kept, Q, 1, C): only [-1] is read
    trk.last_frame_embeds = fe[0, :, prev].t()[torch.as_tensor(idx_prod[prev])][:, None, :]
    with torch.no_grad():
        t_out = trk.forward(fe[:, :, t0:], None, resume=True, frame_embeds_no_norm=P["frame_embds_no_norm"][:, :, t0:],
                            with_masks=False)
    assert np.array_equal(t_out["indices"], idx_prod[t0:]), "assignment indices of the last window differ"
    e = float((t_out["pred_embds"] - inst[:, :, t0:]).abs().max())
    rep(f"config #4 T=64 suffix (frames 61-63, oracle tracker resumed from the product's state at frame 60): instance "
        f"embeddings max |product - oracle| {e:.2e} at max |value| {float(inst.abs().max()):.1f}")
    assert e <= 1e-3
    e = float((t_out["pred_logits"] - P["online_logits"][:, t0:]).abs().max())
    rep(f"config #4 T=64 suffix tracker class logits: max |product - oracle| {e:.2e}")
    assert e <= 1e-3
    # ---- (3) refiner over all 64 frames on the product's tracker outputs
    with torch.no_grad():
        r = O.refiner_forward(O._sub(sd, "refiner."), inst, P["frame_embds_no_norm"],
                              torch.zeros(1, T, 256, 1, 1), 8, 6)
    for key, got in (("pred_embds", P["refiner_embds"]), ("pred_logits", P["refiner_logits"]), ("mask_embed", P["refiner_mask_embed"])):
        e = float((r[key] - got).abs().max())
        rep(f"config #4 T=64 refiner over 64 frames, {key}: max |product - oracle| {e:.2e} at max |value| {float(r[key].abs().max()):.1f}")
        assert e <= 1e-3
