"""Shared by the `-m gpu` pipeline tests: run the CPU oracle's restatement of the reference pipeline
(oracle/dvis_torch.py: windowed, frame-by-frame tracker) on the same frames and weights as the product, and compare
task outputs.  Test infrastructure only.

Parity is asserted from the backbone OUTPUTS onward (the R50 is un-vendored third-party code, "parity unpinned"; a
50-layer random-init conv net also amplifies MIOpen-vs-CPU rounding): the oracle's windows get the features the GPU
backbone produced for the same frames.

Integer outputs: segment lists, query ids, top-k (query, class) pairs must be EQUAL.  Per-pixel maps are compared with
intcmp.near_boundary: the product's mask logits agree with the oracle's to ~1e-5 (different fp32 summation orders), so a
pixel may differ only where the ORACLE's own value is within `tol` of the decision boundary; the differing count and the
worst distance are reported.  tol = BASELINE.json's 1e-3 on logits (instance masks: |resized logit|), resp. 1e-3 x the
sigmoid's largest slope 0.25 on probabilities (panoptic arg-max margin in units of the largest class score, and the 0.5
confidence test).
"""
import torch

import intcmp

DEV = "cuda:0"
TOL_LOGIT = 1e-3
TOL_PROB = 2.5e-4
SHARPEN = 40.0            # sharpen_masks() gain used by the 720p tests
# With the mask heads scaled by SHARPEN the mask logits are SHARPEN x larger (|logit| up to ~50 instead of ~1), and so is
# the absolute error that a given RELATIVE accuracy of the embeddings leaves on them.  BASELINE.json's "1e-3 on mask logits"
# is stated for logits of order 1; the sharpened tests keep the same relative bar (logit_tolerance) — and they MEASURE the
# product-vs-oracle logit error, assert it, and allow a pixel to differ only where that much logit error can explain it.
def logit_tolerance(max_abs_logit):
    """BASELINE's 1e-3 at |logit| <= 4, the same RELATIVE accuracy (2.5e-4) beyond."""
    return max(TOL_LOGIT, 2.5e-4 * max_abs_logit)


def measured_logit_error(model_debug, oracle_masks, ids, what):
    """max |product mask logit - oracle mask logit| over the candidate queries `ids` (stride-4 logits, all frames)."""
    with torch.no_grad():
        got = model_debug["mask_fn"](torch.as_tensor(ids, device=DEV)).cpu()
    want = oracle_masks[torch.as_tensor(ids)]
    err, scale = float((got - want).abs().max()), float(want.abs().max())
    intcmp._report(f"{what}: mask logits of {len(ids)} candidates: max |product - oracle| {err:.3e} at max |logit| {scale:.1f}")
    return err, scale


def cpu_state(m):
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    sd["pixel_mean"], sd["pixel_std"] = m.pixel_mean.detach().cpu().clone(), m.pixel_std.detach().cpu().clone()
    return sd


def perturb_msda(pd):
    """The reference's init makes every sampling offset a constant and every attention weight uniform; give the
    deformable attention something to do."""
    with torch.no_grad():
        for l in pd.transformer.encoder.layers:
            l.self_attn.sampling_offsets.weight.normal_(0, 0.02)
            l.self_attn.attention_weights.weight.normal_(0, 0.1)


def sharpen_masks(m, gain):
    """Random-init mask heads give logits of order 1e-2: every sigmoid is 0.5 +- 0.005, every candidate ties with every
    other and an arg-max comparison would accept anything.  Scale the last layer of the three mask-embedding MLPs (both
    sides read the same state_dict afterwards) so that masks are decisive — as trained ones are."""
    heads = [m.sem_seg_head.predictor.mask_embed]
    heads += [mod.mask_embed for mod in (m.tracker, m.refiner) if mod is not None]
    with torch.no_grad():
        for h in heads:
            h.layers[-1].weight.mul_(gain)


def gpu_backbone(m):
    def backbone_from_gpu(images_cpu):
        with torch.no_grad():
            return {k: v.cpu() for k, v in m.backbone(images_cpu.to(DEV)).items()}
    return backbone_from_gpu


def run_oracle(m, sd, frames_cpu, *, offline, task, attn_masks=False, **cfg):
    """-> (task outputs of oracle.dvis_plus_forward, stages dict with the floats behind the integer decisions)."""
    from oracle import dvis_torch as O
    stages = {"want_attn_masks": True} if attn_masks else {}
    # Host threads: torch's default on the GPU box (128 of 256 logical CPUs) makes the oracle's 3-frame windows 2.3x SLOWER than 32
    # threads and no faster than 8 (tools/exp/oracle_threads.py, profiles/r06_oracle_threads.txt) — so 8 threads per op, and the
    # windows' independent segmenter passes on up to 8 host threads at once (oracle.dvis_plus_forward: seg_workers).
    import os
    prev = torch.get_num_threads()
    ncpu = os.cpu_count() or 8
    workers = max(1, min(8, ncpu // 16, (len(frames_cpu) + 2) // 3))
    torch.set_num_threads(max(4, min(8 if workers > 1 else 32, ncpu)))
    try:
        with torch.no_grad():
            ref = O.dvis_plus_forward(sd, gpu_backbone(m), frames_cpu, offline=offline, task=task, stages=stages,
                                      seg_workers=workers, **cfg)
    finally:
        torch.set_num_threads(prev)
    return ref, stages


def compare_vps(out, ref, stages, what, max_count=None, tol_logit=TOL_LOGIT):
    """Panoptic map: pixel (t, y, x) takes candidate argmax_k s_k p_k and is kept if the winner's p >= 0.5.  A logit error
    of `tol_logit` moves p_k by at most tol_logit * p_k (1 - p_k) (the sigmoid's slope; the resizes are convex
    combinations), so a differing pixel is legitimate only if the oracle's top-2 gap s_1 p_1 - s_2 p_2 is below
    s_1 dp_1 + s_2 dp_2, or its winner's |p - 0.5| below dp.  The `distance` handed to intcmp is the gap MINUS that
    allowance (in units of the largest score): it must be <= a small slack for every pixel that differs.  Saturated
    masks (p = 0 or 1 exactly) therefore get no allowance at all: there the decision is the score order, which is exact."""
    pan, segs, ids = ref
    assert out["segments_infos"] == segs, f"{what}: segment lists differ\n{out['segments_infos']}\n{segs}"
    assert out["pred_ids"] == ids, f"{what}: query ids differ"
    got = out["pred_masks"].cpu()
    if not segs:
        assert int(got.abs().sum()) == 0
        intcmp._report(f"{what}: no segment survives on either side (empty maps)")
        return 0
    probs, scores, best = stages["vps_probs"], stages["vps_scores"], stages["vps_ids"]
    smax = float(scores.max())
    K = probs.shape[0]
    if K > 1:
        top_v, top_i = (scores.view(-1, 1, 1, 1) * probs).topk(2, dim=0)
        p12 = probs.gather(0, top_i)
        s12 = scores[top_i]
        allowance = (s12 * (tol_logit * p12 * (1 - p12) + 1e-7)).sum(0)
        gap = (top_v[0] - top_v[1] - allowance) / smax
    else:
        gap = torch.full_like(probs[0], float("inf"))
    pb = probs.gather(0, best[None])[0]
    conf = (pb - 0.5).abs() - (tol_logit * pb * (1 - pb) + 1e-7)
    return intcmp.near_boundary(got, pan, torch.minimum(gap, conf), 1e-5,
                                f"{what}: panoptic map vs oracle ({len(segs)} segments, {K} candidates, logit allowance "
                                f"{tol_logit:.1e})", max_count=max_count)


def compare_vis(out, ref, stages, what, max_count=None, tol=TOL_LOGIT):
    scores, labels, qidx, masks = ref
    # topk(sorted=False) returns the same SET in a device-dependent order: align on (query, label)
    key_ref = qidx * 1000 + labels
    key_out = out["pred_ids"].cpu() * 1000 + out["pred_labels"].cpu()
    o_ref, o_out = key_ref.argsort(), key_out.argsort()
    assert torch.equal(key_ref[o_ref], key_out[o_out]), f"{what}: top-k (query, class) pairs differ"
    torch.testing.assert_close(out["pred_scores"].cpu()[o_out], scores[o_ref], rtol=1e-3, atol=1e-4)
    return intcmp.near_boundary(out["pred_masks"].cpu()[o_out], masks[o_ref], stages["vis_values"][o_ref].abs(),
                                tol, f"{what}: instance masks vs oracle", max_count=max_count)


def compare_vss(out, ref, stages, what, max_count=None):
    margin = intcmp.argmax_margin(stages["vss_sums"])
    return intcmp.near_boundary(out["pred_masks"].cpu(), ref, margin, TOL_PROB, f"{what}: semantic map vs oracle",
                                max_count=max_count)


# ------------------------------------------------------------------------------------------------------------------
# Error budget: where does the product-vs-oracle difference of the final mask logits come from?  Both sides expose the
# same stage tensors (model.debug_stages / oracle `stages`, the reference's layouts); every row is max |product - oracle|,
# the oracle's max |value| and their ratio.  Written to $DVIS_PARITY_REPORT like every other comparison.
# ------------------------------------------------------------------------------------------------------------------
STAGES = (("mask_features", "pixel decoder: mask_features (T,Cm,h,w)"),
          ("frame_embds_no_norm", "decoder: per-frame queries, un-normed (1,2C,T,Q)"),
          ("frame_embds", "decoder: per-frame queries, normed"),
          ("instance_embds", "tracker: instance embeddings (1,2C,T,Q)"),
          ("online_logits", "tracker: class logits (1,T,Q,K+1)"),
          ("refiner_embds", "refiner: embeddings after decoder_norm (1,2C,T,Q)"),
          ("refiner_logits", "refiner: class logits"),
          ("refiner_mask_embed", "refiner: mask embeddings (1,T,Q,Cm)"))


def error_budget(product, oracle, mask_logits_product, what, frames=(0, 10, 20, 29)):
    """-> {stage: (max abs err, max |oracle|)}; reports one line per stage + the tracker's error at a few frames."""
    rows = {}
    _r = intcmp._report
    _r(f"error budget, {what}: stage | max |product - oracle| | max |oracle| | ratio")
    for key, title in STAGES:
        if key not in product or key not in oracle or oracle[key] is None:
            continue
        a, b = product[key].detach().float().cpu(), oracle[key].detach().float().cpu()
        if key == "mask_features" and b.dim() == 5:
            b = b[0]
        assert a.shape == b.shape, f"{key}: {tuple(a.shape)} vs {tuple(b.shape)}"
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        rows[key] = (err, scale)
        _r(f"  {title}: {err:.3e} | {scale:.3e} | {err / max(scale, 1e-30):.2e}")
        if key == "instance_embds":
            T = a.shape[2]
            per_t = [(t, float((a[:, :, t] - b[:, :, t]).abs().max())) for t in frames if t < T]
            _r("    tracker recurrence, per frame: " + ", ".join(f"t={t}: {e:.2e}" for t, e in per_t))
    if mask_logits_product is not None:
        a, b = mask_logits_product.detach().float().cpu(), oracle["masks"].float()
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        rows["mask_logits"] = (err, scale)
        _r(f"  final mask logits (Q,T,h,w): {err:.3e} | {scale:.3e} | {err / max(scale, 1e-30):.2e}")
    return rows


def attention_mask_flips(product_masks, oracle_masks, T):
    """Decoder attention masks are BOOLEANS derived from logits (sigmoid < 0.5): where a down-sized logit sits within the
    two pipelines' rounding difference of 0, the bit differs, and that query then attends to a different key set — a
    discrete event that moves its embedding by far more than rounding does.  product_masks: list over layers of
    (T, Q, hw) bool (model.sem_seg_head.predictor.debug_masks); oracle_masks: list over windows of lists over layers of
    (t_w, Q, hw).  -> (flips (layers, T, Q) int64, total bits per layer)."""
    L = len(product_masks)
    flips, bits = [], []
    for l in range(L):
        o = torch.cat([w[l] for w in oracle_masks], 0)
        p = product_masks[l].cpu()
        assert p.shape == o.shape, (p.shape, o.shape)
        flips.append((p != o).sum(-1))
        bits.append(o[0].numel() * T)
    flips = torch.stack(flips)
    intcmp._report("decoder attention masks, differing bits per layer (product vs oracle): "
                   + ", ".join(f"L{l}: {int(flips[l].sum())}/{bits[l]}" for l in range(L)))
    return flips, bits
