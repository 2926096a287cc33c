"""CPU: host logic of the product modules (wiring, state_dict compatibility, batching tricks, post-processing,
clip sharding) against the reference goldens, with the HIP ops replaced by oracle stand-ins (`oracle_ops`)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import Golden, ROOT

TOL = dict(rtol=1e-4, atol=1e-5)


def test_pixel_decoder_loads_reference_state_dict_and_matches(oracle_ops):
    from dvis_plus_amd.pixel_decoder import MSDeformAttnPixelDecoder
    from dvis_plus_amd.registry import ShapeSpec
    g = Golden("g2_pixel_decoder")
    chans = g.meta["cfg"]["chans"]
    strides = dict(res2=4, res3=8, res4=16, res5=32)
    pd = MSDeformAttnPixelDecoder({k: ShapeSpec(channels=chans[k], stride=strides[k]) for k in chans},
                                  transformer_dropout=0.0, transformer_nheads=2, transformer_dim_feedforward=64,
                                  transformer_enc_layers=2, conv_dim=32, mask_dim=16, norm="GN",
                                  transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    pd.load_state_dict(g.sd, strict=True)                       # checkpoint surface: identical keys
    feats = {k[5:]: v for k, v in g.ins.items() if k.startswith("feat_")}
    with torch.no_grad():
        mf, out0, ms = pd.forward_features(feats)
        attn = pd.transformer.encoder.layers[0].self_attn
        shapes = torch.tensor([(2, 3), (4, 6), (8, 12)])
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        a = attn(g.ins["attn_query"], g.ins["attn_ref"], g.ins["attn_src"], shapes, lsi, None)
    torch.testing.assert_close(mf, g.outs["mask_features"], **TOL)
    torch.testing.assert_close(out0, g.outs["out0"], **TOL)
    for x, k in zip(ms, ("ms0", "ms1", "ms2")):
        torch.testing.assert_close(x, g.outs[k], **TOL)
    torch.testing.assert_close(a, g.outs["attn_out"], **TOL)


def test_msdeformattn_op_has_no_silent_fallback():
    """The CUDA-op surface (MSDeformAttnFunction / ms_deform_attn_forward) raises on CPU tensors — the reference's module hides
    that behind a bare `except` (ms_deform_attn.py:116-121).  The MODULE serves CPU tensors through the reference's torch
    formulation by device (test_config1_cpu.py); a GPU tensor never takes it."""
    from dvis_plus_amd import functions as Fn
    shapes = torch.tensor([(2, 3), (4, 6), (8, 12)])
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value, loc, w = torch.randn(1, S, 2, 16), torch.rand(1, S, 2, 3, 4, 2), torch.rand(1, S, 2, 3, 4)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        Fn.MSDeformAttnFunction.apply(value, shapes, lsi, loc, w, 128)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        Fn.ms_deform_attn_forward(value, shapes, lsi, loc, w)


def test_decoders_match_goldens(oracle_ops):
    from dvis_plus_amd.transformer_decoder import (MultiScaleMaskedTransformerDecoder,
                                                   VideoMultiScaleMaskedTransformerDecoder_dvisPlus)
    g = Golden("g3_decoder_dvisplus")
    dec = VideoMultiScaleMaskedTransformerDecoder_dvisPlus(
        32, True, num_classes=7, hidden_dim=32, num_queries=6, nheads=2, dim_feedforward=64, dec_layers=3,
        pre_norm=False, mask_dim=16, enforce_input_project=False, num_frames=2, num_reid_head_layers=3,
        reid_hidden_dim=32).eval()
    dec.load_state_dict(g.sd, strict=True)
    i, o = g.ins, g.outs
    with torch.no_grad():
        out = dec([i["x0"], i["x1"], i["x2"]], i["mask_features"])
    for k in ("pred_logits", "pred_masks", "pred_embds", "pred_embds_without_norm", "pred_reid_embed"):
        torch.testing.assert_close(out[k], o[k], **TOL)
    g = Golden("g3_decoder_image")
    dec = MultiScaleMaskedTransformerDecoder(32, True, num_classes=7, hidden_dim=32, num_queries=6, nheads=2,
                                             dim_feedforward=64, dec_layers=3, pre_norm=False, mask_dim=16,
                                             enforce_input_project=False).eval()
    dec.load_state_dict(g.sd, strict=True)
    with torch.no_grad():
        out = dec([g.ins["x0"], g.ins["x1"], g.ins["x2"]], g.ins["mask_features"])
    torch.testing.assert_close(out["pred_logits"], g.outs["pred_logits"], **TOL)
    torch.testing.assert_close(out["pred_masks"], g.outs["pred_masks"], **TOL)


def test_static_query_version_shim(oracle_ops):
    """state-dict shim of the reference (video_mask2former_transformer_decoder.py:213-234)."""
    from dvis_plus_amd.transformer_decoder import MultiScaleMaskedTransformerDecoder
    g = Golden("g3_decoder_image")
    sd = {k.replace("query_feat", "static_query"): v for k, v in g.sd.items()}
    dec = MultiScaleMaskedTransformerDecoder(32, True, num_classes=7, hidden_dim=32, num_queries=6, nheads=2,
                                             dim_feedforward=64, dec_layers=3, pre_norm=False, mask_dim=16,
                                             enforce_input_project=False)
    missing, unexpected = dec.load_state_dict(sd, strict=False)
    assert not missing and not unexpected


def test_tracker_with_resume_matches_golden_indices_bit_exact(oracle_ops):
    from dvis_plus_amd.tracker import ReferringTracker_noiser
    g = Golden("g4_tracker")
    cfg, i, o = g.meta["cfg"], g.ins, g.outs
    trk = ReferringTracker_noiser(hidden_channel=cfg["C"], feedforward_channel=cfg["ffn"], num_head=cfg["heads"],
                                  decoder_layer_num=cfg["layers"], noise_mode="wa", mask_dim=cfg["mask_dim"],
                                  class_num=cfg["K"]).eval()
    trk.load_state_dict(g.sd, strict=True)
    T1 = cfg["T1"]
    fe, fn, mf = i["frame_embeds"], i["frame_embeds_no_norm"], i["mask_features"]
    with torch.no_grad():
        a, ia = trk(fe[:, :, :T1], mf[:, :T1], resume=False, return_indices=True, frame_embeds_no_norm=fn[:, :, :T1])
        b, ib = trk(fe[:, :, T1:], mf[:, T1:], resume=True, return_indices=True, frame_embeds_no_norm=fn[:, :, T1:])
    for tag, r, idx in (("a", a, ia), ("b", b, ib)):
        assert np.array_equal(np.stack(idx), o[f"{tag}_indices"].numpy())
        for k in ("pred_logits", "pred_masks", "pred_embds", "pred_references"):
            torch.testing.assert_close(r[k], o[f"{tag}_{k}"], **TOL)


def test_refiner_matches_golden(oracle_ops):
    from dvis_plus_amd.refiner import TemporalRefiner
    g = Golden("g4_refiner")
    cfg, i, o = g.meta["cfg"], g.ins, g.outs
    ref = TemporalRefiner(hidden_channel=cfg["C"], feedforward_channel=cfg["ffn"], num_head=cfg["heads"],
                          decoder_layer_num=cfg["layers"], mask_dim=cfg["mask_dim"], class_num=cfg["K"],
                          windows=2).eval()
    ref.load_state_dict(g.sd, strict=True)
    with torch.no_grad():
        r = ref(i["instance_embeds"], i["frame_embeds"], i["mask_features"])
        sub = ref(i["instance_embeds"], i["frame_embeds"], i["mask_features"], query_index=torch.tensor([4, 1]))
    for k in ("pred_logits", "pred_masks", "pred_embds"):
        torch.testing.assert_close(r[k], o[k], **TOL)
    torch.testing.assert_close(sub["pred_masks"], o["pred_masks"][:, [4, 1]], **TOL)


def test_match_chain_equals_framewise_reference_matching():
    """The one-call clip recurrence == Noiser.match_embds applied frame by frame on re-ordered embeddings."""
    from dvis_plus_amd.tracker import cosine_costs, match_chain
    from oracle.dvis_torch import match_embds
    g = torch.Generator().manual_seed(0)
    T, Q, C = 7, 20, 32
    base = torch.randn(Q, C, generator=g)
    cur = torch.stack([base[torch.randperm(Q, generator=g)] + 0.4 * torch.randn(Q, C, generator=g) for _ in range(T)])
    idx = match_chain(cosine_costs(cur, cur[0]))
    last = None
    for i in range(T):
        ref = cur[i] if last is None else last
        want = match_embds(ref[:, None], cur[i][:, None])
        assert np.array_equal(idx[i], want), i
        last = cur[i][want]


@pytest.mark.parametrize("kind", ["tracked", "unrelated", "ties", "identical", "zeros"])
def test_threaded_match_chain_equals_sequential_solver(kind):
    """dvis_match_chain solves the frames' canonical problems on host threads and composes the permutations where the
    optimum is certified unique (no alternating cycle of tight reduced costs), else re-solves in the chain's row order:
    the indices must be those of the frame-by-frame loop — scipy's scan order on tied matrices included (duplicate queries,
    zero vectors), which is what `ties` / `zeros` / `identical` exercise."""
    from dvis_plus_amd.tracker import cosine_costs, match_chain
    from oracle.dvis_torch import match_embds
    g = torch.Generator().manual_seed(3)
    T, Q, C = 9, 40, 64
    if kind == "tracked":
        base = torch.randn(Q, C, generator=g)
        cur = torch.stack([base[torch.randperm(Q, generator=g)] + 0.3 * torch.randn(Q, C, generator=g) for _ in range(T)])
    else:
        cur = torch.randn(T, Q, C, generator=g)
    if kind == "ties":
        cur[:, 4] = cur[:, 5]
        cur[:, 20] = cur[:, 21]
        cur[2:5, 10:14] = 0
    if kind == "identical":
        cur[:] = cur[0].clone()
    if kind == "zeros":
        cur[:] = 0
    idx = match_chain(cosine_costs(cur, cur[0]))
    last = None
    for i in range(T):
        ref = cur[i] if last is None else last
        want = match_embds(ref[:, None], cur[i][:, None])
        assert np.array_equal(idx[i], want), (kind, i)
        last = cur[i][want]


def test_lsap_matches_golden_and_scipy():
    import ctypes
    from scipy.optimize import linear_sum_assignment
    from dvis_plus_amd import native
    g = Golden("g5_match")

    def solve(C):
        C = np.ascontiguousarray(C, dtype=np.float64)
        out = np.empty(C.shape[0], dtype=np.int64)
        rc = native.lib().dvis_lsap_solve(C.ctypes.data_as(ctypes.c_void_p), C.shape[0], C.shape[1],
                                          out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        return out
    for n in range(g.meta["ncases"]):
        ref, cur = g.ins[f"ref{n}"][:, 0], g.ins[f"cur{n}"][:, 0]
        ref = ref / (ref.norm(dim=1)[:, None] + 1e-6)
        cur = cur / (cur.norm(dim=1)[:, None] + 1e-6)
        C = 1 - torch.mm(cur, ref.t())
        C = torch.where(torch.isnan(C), torch.zeros_like(C), C)
        assert np.array_equal(solve(C.t().numpy()), g.outs[f"idx{n}"].numpy()), n
    rng = np.random.default_rng(0)
    for t in range(200):                       # half of these are heavily tied integer matrices
        nr = int(rng.integers(1, 30))
        nc = nr + int(rng.integers(0, 5))
        C = rng.integers(0, 4, size=(nr, nc)).astype(np.float64) if t % 2 else rng.random((nr, nc))
        assert np.array_equal(solve(C), linear_sum_assignment(C)[1])
    bad = np.array([[0.0, np.nan], [1.0, 2.0]])
    out = np.empty(2, dtype=np.int64)
    assert native.lib().dvis_lsap_solve(bad.ctypes.data_as(ctypes.c_void_p), 2, 2, out.ctypes.data_as(ctypes.c_void_p)) < 0


def test_postprocess_matches_golden_bit_exact():
    from dvis_plus_amd import postprocess as P
    g = Golden("g6_postprocess")
    i, o, cfg = g.ins, g.outs, g.meta["cfg"]
    logits, aux = P.mean_logits(i["pred_logits"], i["aux_logits"])
    masks = i["pred_masks"][0]
    fn = lambda idx: masks if idx is None else masks[idx]
    img, out_hw, first = cfg["img_size"], cfg["out_hw"], cfg["first_resize"]
    v = P.inference_video_vis(logits, fn, img, out_hw, first, cfg["K"], cfg["max_num"], aux)
    assert torch.equal(v["pred_scores"], o["vis_scores"]) and torch.equal(v["pred_labels"], o["vis_labels"])
    assert torch.equal(v["pred_ids"], o["vis_ids"]) and torch.equal(v["pred_masks"], o["vis_masks"])
    p = P.inference_video_vps(logits.clone(), fn, img, out_hw, first, cfg["K"], cfg["n_things"],
                              cfg["object_mask_threshold"], cfg["overlap_threshold"], aux, num_frames=cfg["T"])
    assert torch.equal(p["pred_masks"], o["vps_masks"])
    assert [s["id"] for s in p["segments_infos"]] == o["vps_seg_id"].tolist()
    assert [s["category_id"] for s in p["segments_infos"]] == o["vps_seg_cat"].tolist()
    assert p["pred_ids"] == o["vps_ids"].tolist()
    s = P.inference_video_vss(logits, fn, img, out_hw, first, aux, frame_chunk=2)
    assert torch.equal(s["pred_masks"], o["vss_masks"])
    assert torch.equal(P.get_instance_labels(i["pred_logits"]), o["instance_labels"])


TINY = dict(num_classes=10, num_queries=8, n_things=5, hidden_dim=64, nheads=2, dim_feedforward=64, dec_layers=4,
            enc_layers=1, tracker_layers=2, refiner_layers=1, object_mask_threshold=0.05)


def _tiny_model(mode="offline", task="vps"):
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    m = build_dvis_plus_r50(mode, task=task, **TINY)
    with torch.no_grad():
        for l in m.sem_seg_head.pixel_decoder.transformer.encoder.layers:
            l.self_attn.sampling_offsets.weight.normal_(0, 0.05)
            l.self_attn.attention_weights.weight.normal_(0, 0.3)
    return m


def _tiny_clip(T=5, seed=0, hw=(70, 100)):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, 256, (3, *hw), dtype=torch.uint8, generator=g) for _ in range(T)]


@pytest.mark.parametrize("mode,task", [("offline", "vps"), ("offline", "vis"), ("offline", "vss"), ("online", "vps")])
def test_whole_pipeline_equals_oracle_pipeline(oracle_ops, mode, task):
    """Whole-clip batched product pipeline == the reference's windowed, frame-by-frame pipeline (oracle)."""
    from oracle import dvis_torch as O
    m = _tiny_model(mode, task)
    frames = _tiny_clip()
    out = m([{"image": frames, "height": 70, "width": 100}])
    sd = dict(m.state_dict())
    sd["pixel_mean"], sd["pixel_std"] = m.pixel_mean, m.pixel_std
    with torch.no_grad():
        ref = O.dvis_plus_forward(sd, m.backbone, frames, offline=(mode == "offline"), nheads=2, enc_layers=1,
                                  dec_layers=3, tracker_layers=2, refiner_layers=1, num_classes=10, n_things=5,
                                  task=task, object_mask_threshold=0.05, max_num=20)
    if task == "vps":
        pan, segs, ids = ref
        assert torch.equal(out["pred_masks"], pan) and out["segments_infos"] == segs and out["pred_ids"] == ids
        assert len(segs) > 0
    elif task == "vis":
        scores, labels, qidx, masks = ref
        torch.testing.assert_close(out["pred_scores"], scores, rtol=1e-5, atol=1e-6)
        assert torch.equal(out["pred_labels"], labels) and torch.equal(out["pred_masks"], masks)
    else:
        assert torch.equal(out["pred_masks"], ref)


def test_keep_flag_resumes_tracker_across_calls(oracle_ops):
    """demo_long_video semantics (meta_architecture.py:629-632,793): two half clips with keep == one clip (online)."""
    m = _tiny_model("online", "vis")
    frames = _tiny_clip(6)
    whole = m([{"image": frames, "height": 70, "width": 100}])
    m([{"image": frames[:3], "height": 70, "width": 100}])
    second = m([{"image": frames[3:], "height": 70, "width": 100, "keep": True}])
    # tracker state carried over: the second half's instance order follows the first half's
    last_whole = m.tracker.last_frame_embeds.clone()
    m([{"image": frames, "height": 70, "width": 100}])
    torch.testing.assert_close(m.tracker.last_frame_embeds, last_whole, rtol=1e-5, atol=1e-6)
    assert second["pred_masks"].shape[1] == 3 and whole["pred_masks"].shape[1] == 6


STREAM_CLIPS = [(5, 3), (4, 4), (5, 6)]          # (T, seed): ragged rounds — with 2 ranks the last round holds one clip


def _stream_worker(rank, world, port, out_dir, owner_rounds):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DVIS_OWNER_ROUNDS=str(owner_rounds))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import conftest as c
    from dvis_plus_amd import functions as Fn
    Fn.attention, Fn.attn_mask, Fn.mask_logits = c._o_attention, c._o_attn_mask, c._o_mask_logits
    Fn.msda_fused_forward, Fn.MSDeformAttnFunction = c._o_msda_fused, c._OMSDAFunction
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _tiny_model("offline", "vps")
    calls = []
    core = m._track_core
    m._track_core = lambda *a: (calls.append(1), core(*a))[1]
    clips = [{"image": _tiny_clip(T, seed=s), "height": 70, "width": 100} for T, s in STREAM_CLIPS]
    outs = [{"masks": o["pred_masks"], "segs": o["segments_infos"], "frame_ids": o["frame_ids"]} for o in m.stream(clips)]
    torch.save({"outs": outs, "tracked": len(calls)}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,owner_rounds", [(2, 1), (2, 0), (3, 1)])
def test_clip_stream_world_size_2_gloo(oracle_ops, tmp_path, world, owner_rounds):
    """stream() under frame sharding: phase A has no collective, phase B issues them in the same order on every rank.
    owner_rounds=1: clips go in rounds of `world`, clip j of a round is tracked + refined by rank j only (3 clips on 2
    ranks -> rank 0 tracks two, rank 1 one; on 3 ranks one each) and the results are all-gathered; a rank's frames of a
    round are one segmenter batch; the ragged split rotates clip by clip.  owner_rounds=0: every rank tracks every clip,
    the split does not rotate."""
    import torch.multiprocessing as mp
    port = 33500 + (os.getpid() % 2000) + 10 * world + owner_rounds
    mp.spawn(_stream_worker, args=(world, port, str(tmp_path), owner_rounds), nprocs=world, join=True)
    m = _tiny_model("offline", "vps")
    parts = [torch.load(tmp_path / f"s{r}.pt") for r in range(world)]
    n = len(STREAM_CLIPS)
    want_tracked = [len(range(r, n, world)) for r in range(world)] if owner_rounds else [n] * world
    assert [p["tracked"] for p in parts] == want_tracked
    for ci, (T, seed) in enumerate(STREAM_CLIPS):
        single = m([{"image": _tiny_clip(T, seed=seed), "height": 70, "width": 100}])
        # owner rounds: clip ci shards with the block -> rank assignment rotated by ci (the short block changes rank every
        # clip, a round's merged batches are equal); one clip per round (replicated tracker): fixed split, one batch
        # shape per rank
        per = (T + world - 1) // world
        blocks = [list(range(min(T, b * per), min(T, (b + 1) * per))) for b in range(world)]
        order = [(r - (ci if owner_rounds else 0)) % world for r in range(world)]               # block held by rank r
        assert [p["outs"][ci]["frame_ids"] for p in parts] == [blocks[b] for b in order]
        by_block = sorted(range(world), key=lambda r: order[r])
        assert torch.equal(torch.cat([parts[r]["outs"][ci]["masks"] for r in by_block], 0), single["pred_masks"])
        assert all(p["outs"][ci]["segs"] == single["segments_infos"] for p in parts)


def _replica_worker(rank, world, port, out_dir, perturb):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      DVIS_OWNER_ROUNDS="0", DVIS_CHECK_REPLICAS="1")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import conftest as c
    from dvis_plus_amd import functions as Fn
    Fn.attention, Fn.attn_mask, Fn.mask_logits = c._o_attention, c._o_attn_mask, c._o_mask_logits
    Fn.msda_fused_forward, Fn.MSDeformAttnFunction = c._o_msda_fused, c._OMSDAFunction
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _tiny_model("offline", "vps")
    m.debug_stages = {}
    core = m._track_core
    if perturb and rank == 1:          # one ulp-sized disagreement on one rank: what a non-deterministic kernel would do
        def core_off(*a):
            emb, cls, aux = core(*a)
            cls = cls.clone()
            cls.view(-1)[3] = torch.nextafter(cls.view(-1)[3], cls.new_tensor(float("inf")))
            return emb, cls, aux
        m._track_core = core_off
    err = None
    try:
        m([{"image": _tiny_clip(5), "height": 70, "width": 100}])
    except RuntimeError as e:
        err = str(e)
    torch.save({"err": err, "cls": m.debug_stages.get("cls"), "aux": m.debug_stages.get("aux")},
               os.path.join(out_dir, f"c{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("perturb", [False, True])
def test_replicated_tracker_agrees_across_ranks_gloo(oracle_ops, tmp_path, perturb):
    """north_star's split runs tracker + refiner REPLICATED and relies on every rank computing the same bits (no
    broadcast): the class logits of both ranks are torch.equal, and the DVIS_CHECK_REPLICAS=1 guard (an all-gathered
    checksum before post-processing) turns a one-ulp disagreement into an error on every rank instead of a hang."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000) + int(perturb)
    mp.spawn(_replica_worker, args=(2, port, str(tmp_path), perturb), nprocs=2, join=True)
    parts = [torch.load(tmp_path / f"c{r}.pt") for r in range(2)]
    if perturb:
        assert all(p["err"] is not None and "differ between rank 0" in p["err"] for p in parts), [p["err"] for p in parts]
    else:
        assert all(p["err"] is None for p in parts), [p["err"] for p in parts]
        assert torch.equal(parts[0]["cls"], parts[1]["cls"]) and torch.equal(parts[0]["aux"], parts[1]["aux"])


def _shard_worker(rank, world, port, out_dir, rounds=1, T=5):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import conftest as c
    from dvis_plus_amd import functions as Fn
    Fn.attention, Fn.attn_mask, Fn.mask_logits = c._o_attention, c._o_attn_mask, c._o_mask_logits
    Fn.msda_fused_forward, Fn.MSDeformAttnFunction = c._o_msda_fused, c._OMSDAFunction
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _tiny_model("offline", "vps")
    m.pipeline_rounds = rounds
    out = m([{"image": _tiny_clip(T), "height": 70, "width": 100}])
    torch.save({"masks": out["pred_masks"], "segs": out["segments_infos"], "ids": out["pred_ids"],
                "range": out["frame_range"], "frame_ids": out["frame_ids"]}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_clip_sharding_world_size_2_gloo(oracle_ops, tmp_path):
    """Frames sharded over 2 ranks + ONE all-gather of the per-frame queries == the single-process result
    (uneven split: 5 frames -> 3 + 2)."""
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = _tiny_model("offline", "vps")([{"image": _tiny_clip(5), "height": 70, "width": 100}])
    parts = [torch.load(tmp_path / f"r{r}.pt") for r in range(2)]
    assert [p["range"] for p in parts] == [(0, 3), (3, 5)]
    assert torch.equal(torch.cat([p["masks"] for p in parts], 0), single["pred_masks"])
    for p in parts:
        assert p["segs"] == single["segments_infos"] and p["ids"] == single["pred_ids"]
    assert len(single["segments_infos"]) > 0


def test_round_plan_covers_every_frame_once_in_order():
    from dvis_plus_amd.clip_shard import ClipShard
    for world in (1, 2, 3, 8):
        for T in (1, 2, 5, 30, 64):
            for rounds in (1, 2, 3, 4, 7):
                seen = []
                plans = []
                for rank in range(world):
                    sh = ClipShard.__new__(ClipShard)
                    sh.group, sh.world, sh.rank = None, world, rank
                    plans.append(sh.round_plan(T, rounds))
                k = plans[0][1]
                for c in range(len(plans[0][0])):
                    start, end = plans[0][0][c][:2]
                    for rank in range(world):
                        s2, e2, lo, hi = plans[rank][0][c]
                        assert (s2, e2) == (start, end) and hi - lo <= k and lo == min(end, start + rank * k)
                        seen += list(range(lo, hi))
                assert seen == list(range(T)), (world, T, rounds)


def test_pipelined_rounds_equal_single_pass(oracle_ops):
    """Tracker fed span by span (resume between spans) == tracker over the whole clip; same integer outputs."""
    frames = _tiny_clip(7)
    ref = _tiny_model("offline", "vps")([{"image": frames, "height": 70, "width": 100}])
    for rounds in (2, 3, 7):
        m = _tiny_model("offline", "vps")
        m.pipeline_rounds = rounds
        out = m([{"image": frames, "height": 70, "width": 100}])
        assert torch.equal(out["pred_masks"], ref["pred_masks"]) and out["segments_infos"] == ref["segments_infos"]
        assert out["frame_ids"] == list(range(7))
    assert len(ref["segments_infos"]) > 0


def test_pipelined_sharding_world_size_2_gloo(oracle_ops, tmp_path):
    """2 ranks x 2 rounds (interleaved spans, one gather per span) == single process; 7 frames -> spans of 4 + 3."""
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path), 2, 7), nprocs=2, join=True)
    single = _tiny_model("offline", "vps")([{"image": _tiny_clip(7), "height": 70, "width": 100}])
    parts = [torch.load(tmp_path / f"r{r}.pt") for r in range(2)]
    assert [p["frame_ids"] for p in parts] == [[0, 1, 4, 5], [2, 3, 6]]
    for p in parts:
        assert torch.equal(p["masks"], single["pred_masks"][p["frame_ids"]])
        assert p["segs"] == single["segments_infos"] and p["ids"] == single["pred_ids"]


def test_image_maskformer_config1_plumbing(oracle_ops):
    """BASELINE config #1: image Mask2Former through the op's torch formulation on the CPU (plumbing check)."""
    from dvis_plus_amd.meta_architecture import build_mask2former_r50
    from oracle import dvis_torch as O
    m = build_mask2former_r50(num_classes=9, num_queries=7, hidden_dim=64, nheads=2, dim_feedforward=64, dec_layers=3,
                              enc_layers=1, semantic_on=True, panoptic_on=True, instance_on=True, thing_ids=(0, 1, 2),
                              object_mask_threshold=0.05, test_topk_per_image=5)
    img = torch.randint(0, 256, (3, 60, 90), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    out = m([{"image": img, "height": 48, "width": 72}])[0]
    sd = dict(m.state_dict())
    sd["pixel_mean"], sd["pixel_std"] = m.pixel_mean, m.pixel_std
    with torch.no_grad():
        sem, logits, masks = O.maskformer_image_forward(sd, m.backbone, img, nheads=2, enc_layers=1, dec_layers=2,
                                                        num_classes=9, out_hw=(48, 72))
    torch.testing.assert_close(out["sem_seg"], sem, rtol=1e-5, atol=1e-6)
    pan, segs = out["panoptic_seg"]
    assert pan.shape == (48, 72) and pan.dtype == torch.int32 and int(pan.max()) == len(segs)
    inst = out["instances"]
    assert inst["pred_masks"].shape[1:] == (48, 72) and inst["scores"].shape == inst["pred_classes"].shape


def test_minvis_meta_architecture_matches_reference_post_processing(oracle_ops):
    """MinVIS (registry name, `_minvis` decoder): the one-call alignment chain + top-10 selection + masks for the selected
    slots only == the reference's frame-by-frame post_processing + inference_video applied to the same decoder outputs."""
    from dvis_plus_amd.meta_architecture import build_dvis_plus_r50
    from oracle import dvis_torch as O
    m = build_dvis_plus_r50("minvis", num_classes=6, num_queries=14, hidden_dim=64, nheads=2, dim_feedforward=64,
                            dec_layers=3, enc_layers=1)
    frames = _tiny_clip(4)
    out = m([{"image": frames, "height": 60, "width": 90}])
    with torch.no_grad():
        images, img_size = m.preprocess(frames)
        ref_dec = m.sem_seg_head(m.backbone(images))                       # the `_minvis` decoder's own outputs
        logits, masks, perms = O.minvis_post_processing(ref_dec["pred_logits"], ref_dec["pred_masks"], ref_dec["pred_embds"])
        s, l, mk, q = O.minvis_inference_video(logits[0], masks[0], img_size, (60, 90), images.shape[-2:], 6, 10)
    assert np.array_equal(out["aligned_indices"].numpy(), perms)           # Hungarian chain: bit-exact
    key_ref, key_out = (q * 100 + l).numpy(), np.array(out["pred_ids"]) * 100 + np.array(out["pred_labels"])
    o_ref, o_out = np.argsort(key_ref), np.argsort(key_out)
    assert np.array_equal(key_ref[o_ref], key_out[o_out])
    np.testing.assert_allclose(np.array(out["pred_scores"])[o_out], s.numpy()[o_ref], rtol=1e-5)
    got = torch.stack(out["pred_masks"])[torch.as_tensor(o_out)]
    assert (got == mk[torch.as_tensor(o_ref)]).float().mean().item() > 0.999


def test_clip_stream_equals_clip_by_clip_forward(oracle_ops):
    """stream(): phase A of clip i+1 is issued before phase B of clip i — same outputs as forward, clip by clip."""
    m = _tiny_model("offline", "vps")
    clips = [{"image": _tiny_clip(4, seed=s), "height": 70, "width": 100} for s in (0, 1, 2)]
    want = [m([c]) for c in clips]
    got = list(m.stream(clips))
    assert len(got) == 3
    for a, b in zip(got, want):
        assert torch.equal(a["pred_masks"], b["pred_masks"]) and a["segments_infos"] == b["segments_infos"]
        assert a["pred_ids"] == b["pred_ids"] and a["frame_ids"] == b["frame_ids"]
    assert list(m.stream([])) == []


def test_rotated_blocks_cover_every_frame_once():
    """ClipShard.local_range(T, shift): for every shift the ranks' blocks tile [0, T) exactly once, and over a round of
    `world` clips every rank holds the same number of frames (T=30 over 8: 30 each instead of 32 on seven ranks)."""
    from dvis_plus_amd.clip_shard import ClipShard
    for world, T in [(8, 30), (4, 30), (3, 5), (2, 1), (8, 3)]:
        total = [0] * world
        for shift in range(world):
            seen = []
            for r in range(world):
                sh = ClipShard()
                sh.world, sh.rank = world, r
                lo, hi = sh.local_range(T, shift)
                seen += list(range(lo, hi))
                total[r] += hi - lo
            assert sorted(seen) == list(range(T))
        assert len(set(total)) == 1 and total[0] == T


def test_segmenter_calls_are_uncapped_and_equal_shares():
    """The 4 GiB-per-activation cap of rounds 1-2 (55 frames at 720p) is gone with its cause (two concurrent library
    stream-K GEMMs); a user-requested chunk is still cut into equal shares."""
    from dvis_plus_amd.meta_architecture import segmenter_frames_per_call as f
    assert f(30, 736, 1280) == 30 and f(64, 736, 1280) == 64 and f(200, 480, 640) == 200   # one call
    assert f(30, 736, 1280, requested=4) == 4 and f(64, 736, 1280, requested=60) == 32
    assert f(0, 736, 1280) == 1 and f(1, 4000, 6000) == 1


def _keep_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import conftest as c
    from dvis_plus_amd import functions as Fn
    Fn.attention, Fn.attn_mask, Fn.mask_logits = c._o_attention, c._o_attn_mask, c._o_mask_logits
    Fn.msda_fused_forward, Fn.MSDeformAttnFunction = c._o_msda_fused, c._OMSDAFunction
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _tiny_model("offline", "vps")
    m.window_inference = False       # the reference's non-window branch, where `keep` resumes (meta_architecture.py:1329-1334)
    frames = _tiny_clip(8, seed=9)
    clips = [{"image": frames[:4], "height": 70, "width": 100},
             {"image": frames[4:], "height": 70, "width": 100, "keep": True}]     # second half resumes the tracker
    outs = [{"masks": o["pred_masks"], "segs": o["segments_infos"], "frame_ids": o["frame_ids"]} for o in m.stream(clips)]
    torch.save(outs, os.path.join(out_dir, f"k{rank}.pt"))
    dist.destroy_process_group()


def test_clip_stream_with_resumed_tracker_state_takes_the_replicated_path(oracle_ops, tmp_path):
    """A round that contains a `keep` clip (tracker state carried over from the previous call) cannot hand its clips to
    different tracker ranks: every rank runs the tracker of both clips, in order — same results as one process."""
    import torch.multiprocessing as mp
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_keep_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m = _tiny_model("offline", "vps")
    m.window_inference = False       # the reference's non-window branch, where `keep` resumes (meta_architecture.py:1329-1334)
    frames = _tiny_clip(8, seed=9)
    first = m([{"image": frames[:4], "height": 70, "width": 100}])
    second = m([{"image": frames[4:], "height": 70, "width": 100, "keep": True}])
    parts = [torch.load(tmp_path / f"k{r}.pt") for r in range(2)]
    for ci, single in enumerate((first, second)):
        order = sorted(range(2), key=lambda r: parts[r][ci]["frame_ids"][0])
        assert torch.equal(torch.cat([parts[r][ci]["masks"] for r in order], 0), single["pred_masks"])
        assert all(p[ci]["segs"] == single["segments_infos"] for p in parts)


def _keep_after_round_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import conftest as c
    from dvis_plus_amd import functions as Fn
    Fn.attention, Fn.attn_mask, Fn.mask_logits = c._o_attention, c._o_attn_mask, c._o_mask_logits
    Fn.msda_fused_forward, Fn.MSDeformAttnFunction = c._o_msda_fused, c._OMSDAFunction
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _tiny_model("offline", "vps")
    m.window_inference = False       # the reference's non-window branch, where `keep` resumes (meta_architecture.py:1329-1334)
    a, b = _tiny_clip(4, seed=20), _tiny_clip(8, seed=21)
    clips = [{"image": a, "height": 70, "width": 100},                              # round 1 (owner rounds): A on rank 0,
             {"image": b[:4], "height": 70, "width": 100},                          #                         B on rank 1
             {"image": b[4:], "height": 70, "width": 100, "keep": True}]            # round 2: resumes B's tracker state
    outs = [{"masks": o["pred_masks"], "segs": o["segments_infos"], "frame_ids": o["frame_ids"]} for o in m.stream(clips)]
    torch.save(outs, os.path.join(out_dir, f"ka{rank}.pt"))
    dist.destroy_process_group()


def test_keep_clip_after_an_owner_round_resumes_the_last_clips_state(oracle_ops, tmp_path):
    """[A, B, C(keep)] on 2 ranks: the owner round leaves A's tracker state on rank 0 and B's on rank 1; C resumes B, so
    every rank must hold B's state (handed over inside the round's result all-gather) before the replicated path runs."""
    import torch.multiprocessing as mp
    port = 36500 + (os.getpid() % 2000)
    mp.spawn(_keep_after_round_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m = _tiny_model("offline", "vps")
    m.window_inference = False       # the reference's non-window branch, where `keep` resumes (meta_architecture.py:1329-1334)
    a, b = _tiny_clip(4, seed=20), _tiny_clip(8, seed=21)
    want = [m([{"image": a, "height": 70, "width": 100}]), m([{"image": b[:4], "height": 70, "width": 100}]),
            m([{"image": b[4:], "height": 70, "width": 100, "keep": True}])]
    parts = [torch.load(tmp_path / f"ka{r}.pt") for r in range(2)]
    for ci, single in enumerate(want):
        order = sorted(range(2), key=lambda r: parts[r][ci]["frame_ids"][0])
        assert torch.equal(torch.cat([parts[r][ci]["masks"] for r in order], 0), single["pred_masks"]), ci
        assert all(p[ci]["segs"] == single["segments_infos"] for p in parts), ci


def test_reloaded_weights_reach_the_fused_kv_projection(oracle_ops):
    """The tracker / refiner concatenate the cross-attention K/V weights once (one GEMM for all layers); captured
    hipGraphs keep pointers to that tensor, so a reload must refresh it IN PLACE — same storage, new values."""
    from dvis_plus_amd.graphs import GraphRunner
    m, other = _tiny_model("offline", "vps"), _tiny_model("offline", "vps")
    with torch.no_grad():
        for p in other.parameters():
            p.add_(0.05 * torch.randn_like(p))
    clip = [{"image": _tiny_clip(4, seed=2), "height": 70, "width": 100}]
    m(clip)
    ptrs = [mod._kv_weights()[0].data_ptr() for mod in (m.tracker, m.refiner)]
    m.load_state_dict(other.state_dict())
    got = m(clip)
    want = other(clip)
    assert [mod._kv_weights()[0].data_ptr() for mod in (m.tracker, m.refiner)] == ptrs
    assert torch.equal(got["pred_masks"], want["pred_masks"]) and got["segments_infos"] == want["segments_infos"]
    W, _ = m.tracker._kv_weights()
    C = m.tracker.decoder_norm.weight.shape[0]
    # tracker layout: all layers' K rows, then all layers' V rows (layer l's head h = head 8 l + h of ONE attention call)
    L = m.tracker.num_layers
    ipw = other.tracker.transformer_cross_attention_layers[0].multihead_attn.in_proj_weight
    assert torch.equal(W[:C], ipw[C:2 * C]) and torch.equal(W[L * C:(L + 1) * C], ipw[2 * C:])
    Wo, bo = m.tracker._o_weights()
    assert torch.equal(Wo[1], other.tracker.transformer_cross_attention_layers[1].multihead_attn.out_proj.weight)
    assert torch.equal(bo[1], other.tracker.transformer_cross_attention_layers[1].multihead_attn.out_proj.bias)
    # the graph cache is bounded (least recently used entry dropped)
    calls = []
    g = GraphRunner(lambda x: calls.append(1) or x, max_entries=2)
    assert g.max_entries == 2 and len(g._cache) == 0


RAGGED8 = [30] * 9 + [5]        # the last clip leaves three of the eight ranks without a frame


def _northstar_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import conftest as c
    from dvis_plus_amd import functions as Fn
    Fn.attention, Fn.attn_mask, Fn.mask_logits = c._o_attention, c._o_attn_mask, c._o_mask_logits
    Fn.msda_fused_forward, Fn.MSDeformAttnFunction = c._o_msda_fused, c._OMSDAFunction
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _tiny_model("offline", "vps")
    m.owner_rounds = False                        # north_star's split: what bench.py reports as the headline at N > 1
    m.tracker_batch = int(os.environ.get("TEST_TRACKER_BATCH", "1"))
    seen = []
    for name in ("all_gather_into_tensor", "all_reduce", "broadcast", "all_gather", "all_to_all", "reduce_scatter_tensor"):
        orig = getattr(dist, name)
        setattr(dist, name, (lambda o, n: lambda *a, **k: (seen.append(n), o(*a, **k))[1])(orig, name))
    clips = [{"image": _tiny_clip(5, seed=40 + i), "height": 70, "width": 100} for i in range(3)]
    per_clip = []
    for o in m.stream(clips):
        per_clip.append((list(seen), o["frame_ids"], o["segments_infos"], o["pred_masks"]))
        seen.clear()
    torch.save(per_clip, os.path.join(out_dir, f"n{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("tracker_batch", [1, 2])
def test_north_star_split_is_one_all_gather_per_clip(oracle_ops, tmp_path, tracker_batch, monkeypatch):
    """BASELINE.json north_star: "a single RCCL all-gather ... of per-frame object queries before the temporal refiner".
    With the tracker replicated (owner rounds off) a clip costs exactly ONE all-gather; the only other collective is the
    VPS post-processing's sum of the per-segment areas (a few hundred bytes) — no broadcast (phase B is deterministic, every
    rank computes identical tracker / refiner outputs) — and the fixed (non-rotating) ragged split 3 + 2."""
    import torch.multiprocessing as mp
    port = 38500 + (os.getpid() % 2000) + 7 * tracker_batch
    monkeypatch.setenv("TEST_TRACKER_BATCH", str(tracker_batch))
    mp.spawn(_northstar_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = [torch.load(tmp_path / f"n{r}.pt") for r in range(2)]
    m = _tiny_model("offline", "vps")
    singles = [m([{"image": _tiny_clip(5, seed=40 + i), "height": 70, "width": 100}]) for i in range(3)]
    for r, part in enumerate(parts):
        names_all = [n for names, *_ in part for n in names]
        # 3 clips: exactly one all-gather per clip, at most one (VPS area) all-reduce per clip, no broadcast — whether the
        # clips' trackers run one by one or two clips of a round advance together (tracker_batch = 2: rounds of 2 + 1)
        assert names_all.count("all_gather_into_tensor") == 3 and "broadcast" not in names_all, names_all
        assert set(names_all) <= {"all_gather_into_tensor", "all_reduce"} and names_all.count("all_reduce") <= 3
    for ci, single in enumerate(singles):            # stitched maps == the single-process result
        held = sorted((p[ci][1][0], r) for r, p in enumerate(parts) if p[ci][1])
        assert sorted(f for p in parts for f in p[ci][1]) == list(range(5))
        assert torch.equal(torch.cat([parts[r][ci][3] for _, r in held], 0), single["pred_masks"]), ci
        assert all(p[ci][2] == single["segments_infos"] for p in parts), ci


def _ragged8_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import conftest as c
    from dvis_plus_amd import functions as Fn
    Fn.attention, Fn.attn_mask, Fn.mask_logits = c._o_attention, c._o_attn_mask, c._o_mask_logits
    Fn.msda_fused_forward, Fn.MSDeformAttnFunction = c._o_msda_fused, c._OMSDAFunction
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _tiny_model("offline", "vps")
    n_coll = [0]
    for name in ("all_gather_into_tensor", "all_reduce", "broadcast"):
        orig = getattr(dist, name)
        setattr(dist, name, (lambda o: lambda *a, **k: (n_coll.__setitem__(0, n_coll[0] + 1), o(*a, **k))[1])(orig))
    clips = [{"image": _tiny_clip(T, seed=100 + i, hw=(40, 64)), "height": 40, "width": 64} for i, T in enumerate(RAGGED8)]
    outs = [{"masks": o["pred_masks"], "segs": o["segments_infos"], "frame_ids": o["frame_ids"]} for o in m.stream(clips)]
    torch.save({"outs": outs, "collectives": n_coll[0]}, os.path.join(out_dir, f"e{rank}.pt"))
    dist.destroy_process_group()


def test_stream_8_ranks_ragged_T30_ten_clips_gloo(oracle_ops, tmp_path):
    """The bench's multi-GPU shape on CPU: 8 ranks, T=30 (ragged split 4,4,4,4,4,4,4,2 rotating clip by clip), K=10
    clips = one full round + a quarter-full one whose last clip has only 5 frames (three ranks hold none of it).  Every
    rank issues the SAME number of collectives (a rank without a tracker job or without frames must not skip one — the
    post-processing reduction included), and the stitched maps equal the single-process result."""
    import torch.multiprocessing as mp
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_ragged8_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    parts = [torch.load(tmp_path / f"e{r}.pt") for r in range(8)]
    assert len({p["collectives"] for p in parts}) == 1, [p["collectives"] for p in parts]
    m = _tiny_model("offline", "vps")
    for ci, T in enumerate(RAGGED8):
        single = m([{"image": _tiny_clip(T, seed=100 + ci, hw=(40, 64)), "height": 40, "width": 64}])
        held = sorted((p["outs"][ci]["frame_ids"][0], r) for r, p in enumerate(parts) if p["outs"][ci]["frame_ids"])
        assert sorted(f for p in parts for f in p["outs"][ci]["frame_ids"]) == list(range(T))
        assert torch.equal(torch.cat([parts[r]["outs"][ci]["masks"] for _, r in held], 0), single["pred_masks"]), ci
        assert all(p["outs"][ci]["segs"] == single["segments_infos"] for p in parts), ci


# ---- a12 composition against the reference's OWN forward (golden g10_window_loop: DVIS_Plus_offline / _online forward
# eval branches + run_window_inference + post_processing + inference_video_*, meta_architecture.py:1301-1317, 1376-1396,
# 1446-1500, 629-642, 687-706, 774-816, 758-772)
def _g10_check_vps(out, o, tag):
    assert [s["id"] for s in out["segments_infos"]] == o[f"{tag}_seg_id"].tolist(), tag
    assert [s["category_id"] for s in out["segments_infos"]] == o[f"{tag}_seg_cat"].tolist(), tag
    assert [s["isthing"] for s in out["segments_infos"]] == o[f"{tag}_seg_isthing"].tolist(), tag
    assert list(out["pred_ids"]) == o[f"{tag}_ids"].tolist(), tag
    want = o[f"{tag}_masks"].to(torch.int64)
    n = int((out["pred_masks"].cpu().to(torch.int64) != want).sum())
    assert n == 0, f"{tag}: {n} of {want.numel()} panoptic pixels differ from the reference's forward"


@pytest.mark.parametrize("task", ["vps", "vis", "vss"])
def test_offline_meta_architecture_equals_reference_forward_g10(oracle_ops, task):
    import g10_model as G
    m, g, cfg, frames = G.build("offline", task)
    o = g.outs
    with torch.no_grad():
        out = m([G.video(frames, cfg)])
    if task == "vps":
        _g10_check_vps(out, o, "off_vps")
        # `keep` on the offline model: read, but the window loop never resumes from it (meta_architecture.py:1479-1486)
        with torch.no_grad():
            m([G.video(frames, cfg, 0, 4)])
            _g10_check_vps(m([G.video(frames, cfg, 4, None, keep=True)]), o, "off_keep_vps")
    elif task == "vis":
        key_ref = o["off_vis_ids"] * 1000 + o["off_vis_labels"]
        key_out = out["pred_ids"] * 1000 + out["pred_labels"]
        a, b = key_ref.argsort(), key_out.argsort()
        assert torch.equal(key_ref[a], key_out[b])
        torch.testing.assert_close(out["pred_scores"][b], o["off_vis_scores"][a], rtol=1e-4, atol=1e-6)
        assert torch.equal(out["pred_masks"][b], o["off_vis_masks"][a])
    else:
        assert torch.equal(out["pred_masks"], o["off_vss_masks"].to(out["pred_masks"].dtype))


def test_online_meta_architecture_equals_reference_forward_g10(oracle_ops):
    import g10_model as G
    m, g, cfg, frames = G.build("online", "vps")
    o = g.outs
    with torch.no_grad():
        _g10_check_vps(m([G.video(frames, cfg)]), o, "on_vps")
        _g10_check_vps(m([G.video(frames, cfg, 0, 4)]), o, "on_keep_a_vps")
        _g10_check_vps(m([G.video(frames, cfg, 4, None, keep=True)]), o, "on_keep_b_vps")     # resumes (:793)


@pytest.mark.parametrize("mode,task", [("offline", "vis"), ("offline", "vps"), ("offline", "vss"), ("online", "vis")])
def test_reference_output_format_feeds_the_reference_evaluators(oracle_ops, mode, task):
    """`reference_outputs=True` (what `Cls(cfg)` sets): the dict of meta_architecture.py:603-626 as the reference's evaluators
    read it — ytvis_eval.py:268-295 zips python floats / ints with per-instance (T, H, W) masks it turns into numpy and
    serialises to json; vps_eval.py:106-135 / vss_eval.py:92-93 call .numpy() on the maps.  Same values as the device dict."""
    import json
    m = _tiny_model(mode, task)
    frames = _tiny_clip(4, seed=3)
    video = {"image": frames, "height": 70, "width": 100}
    dev_out = m([video])
    m.reference_outputs = True
    out = m([video])
    streamed = list(m.stream([video])) if mode == "offline" else [out]
    for o in (out, streamed[0]):
        assert o["image_size"] == (70, 100) and o["task"] == task and "ready_event" not in o
        if task == "vis":
            assert isinstance(o["pred_scores"], list) and isinstance(o["pred_scores"][0], float)
            assert isinstance(o["pred_labels"][0], int) and isinstance(o["pred_ids"][0], int)
            json.dumps({"score": o["pred_scores"][0], "category_id": o["pred_labels"][0]})
            assert isinstance(o["pred_masks"], list) and len(o["pred_masks"]) == len(o["pred_scores"])
            m0 = o["pred_masks"][0]
            assert m0.device.type == "cpu" and m0.dtype == torch.bool and m0.shape == (4, 70, 100)
            np.array(m0[0][:, :, None], order="F", dtype="uint8")                  # what the evaluator feeds to RLE
            assert o["pred_scores"] == dev_out["pred_scores"].tolist()
            assert torch.equal(torch.stack(o["pred_masks"]), dev_out["pred_masks"].cpu())
        elif task == "vps":
            assert o["pred_masks"].device.type == "cpu" and o["pred_masks"].numpy().shape == (4, 70, 100)
            assert all(isinstance(i, int) for i in o["pred_ids"]) and o["segments_infos"] == dev_out["segments_infos"]
            assert torch.equal(o["pred_masks"], dev_out["pred_masks"].cpu())
        else:
            assert o["pred_masks"].numpy().astype(np.uint8).shape == (4, 70, 100)


def test_tracker_batch_advances_two_clips_together_same_results(oracle_ops):
    """stream() with tracker_batch = 2: the two clips of a round share ONE tracker pass (batch 2 through the recurrence) —
    same outputs per clip as one by one; a clip of another length, and a single left-over clip, take the unbatched path."""
    m = _tiny_model("offline", "vps")
    clips = [{"image": _tiny_clip(T, seed=60 + i), "height": 70, "width": 100} for i, T in enumerate((5, 5, 4, 5, 5))]
    want = [m([c]) for c in clips]
    calls = []
    fwd = m.tracker.forward
    m.tracker.forward = lambda fe, *a, **k: (calls.append(fe.shape[0]), fwd(fe, *a, **k))[1]
    m.tracker_batch = 2
    got = list(m.stream(clips))
    assert calls == [2, 1, 1, 1]          # (5, 5) together; (4, 5) differ in length -> one by one; the last clip alone
    for g, w in zip(got, want):
        assert g["segments_infos"] == w["segments_infos"] and g["pred_ids"] == w["pred_ids"]
        assert torch.equal(g["pred_masks"], w["pred_masks"])


@pytest.mark.parametrize("owner_rounds", [False, True])
def test_emulated_rank_does_one_ranks_work(oracle_ops, owner_rounds):
    """clip_shard.EmulatedShard (tools/rank_emulation.py): one process, the work of rank 0 of 4 — its block of every clip's
    frames through the segmenter, gathered buffers of the real size, the tracker over all T frames (replicated) or over
    the clip it owns (owner rounds), outputs for its own frames only."""
    from dvis_plus_amd.clip_shard import EmulatedShard
    m = _tiny_model("offline", "vps")
    m.owner_rounds = owner_rounds
    m._clip_shard = EmulatedShard(4)
    seg_frames, trk_frames = [], []
    seg, trk = m.segment, m.tracker.forward
    m.segment = lambda images: (seg_frames.append(len(images)), seg(images))[1]
    m.tracker.forward = lambda fe, *a, **k: (trk_frames.append(tuple(fe.shape)), trk(fe, *a, **k))[1]
    clips = [{"image": _tiny_clip(6, seed=s), "height": 70, "width": 100} for s in range(4)]
    outs = list(m.stream(clips))
    assert len(outs) == 4
    if owner_rounds:     # one merged call for the round; the rotation hands rank 0 blocks 0, 3, 2, 1 of (2, 2, 2, 0) frames
        assert seg_frames == [6] and len(trk_frames) == 1 and trk_frames[0][2] == 6
        assert sorted(len(o["frame_ids"]) for o in outs) == [0, 2, 2, 2]
    else:                # four clips, 2 of 6 frames each; the tracker sees all 6 frames of every clip
        assert seg_frames == [2, 2, 2, 2] and [t[2] for t in trk_frames] == [6, 6, 6, 6]
        assert all(o["frame_ids"] == [0, 1] and o["pred_masks"].shape[0] == 2 for o in outs)
    m._clip_shard = None
    assert m.clip_shard.world == 1
