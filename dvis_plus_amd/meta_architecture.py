"""DVIS++ meta-architectures (inference) — SURVEY.md §8 rows a12, a13, a15.

Mirrors the registry names and the ``forward(batched_inputs)`` contract of
  MaskFormerHead      mask2former/modeling/meta_arch/mask_former_head.py:19-152
  DVIS_Plus_online    dvis_Plus/meta_architecture.py:404-1065  (forward :591-706, run_window_inference :774-816)
  DVIS_Plus_offline   dvis_Plus/meta_architecture.py:1068-1580 (forward :1264-1396, run_window_inference :1446-1500)
``batched_inputs = [{"image": [uint8/float (3,H,W)] * T, "height": h, "width": w, optional "keep": bool}]``;
returns the task dict documented at meta_architecture.py:603-626 (tensors stay on the device).
Checkpoint layout: backbone.*, sem_seg_head.{pixel_decoder,predictor}.*, tracker.*, refiner.*

Differences by design (same results):
  * no windowing / CPU off-loading: the whole clip's mask_features stay in HBM (1.8 GB at T=30, 720p);
    the reference's 3-frame windows, per-window .cpu() and per-frame scipy syncs disappear;
  * masks are only contracted for the queries post-processing keeps;
  * optional frame sharding over the GPUs of a node (clip_shard.ClipShard), one all-gather per clip;
  * DVIS_Plus_offline.stream(videos): throughput mode, the next clip's segmenter overlaps the previous clip's
    tracker / refiner / post-processing on a second stream (same results as forward, clip by clip).
Also here: MinVIS (meta_architecture.py:23-407) and the image MaskFormer (mask2former/maskformer_model.py).
"""
import os

import torch
from torch import nn

from . import d2
from . import functions as Fn
from . import native
from . import postprocess as PP
from .clip_shard import ClipShard, EmulatedShard
from .d2 import configurable
from .registry import META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY


def _get(node, key, default):
    """cfg key that only some of the reference's config revisions define."""
    if hasattr(node, "get"):
        return node.get(key, default)
    return getattr(node, key, default)


@SEM_SEG_HEADS_REGISTRY.register()
class MaskFormerHead(nn.Module):
    """mask2former/modeling/meta_arch/mask_former_head.py:19-152: ``MaskFormerHead(cfg, input_shape)`` or explicit
    keyword arguments."""

    @configurable
    def __init__(self, input_shape=None, *, num_classes, pixel_decoder, loss_weight=1.0, ignore_value=-1,
                 transformer_predictor, transformer_in_feature="multi_scale_pixel_decoder",
                 return_transformer_feature=False):
        super().__init__()
        assert transformer_in_feature == "multi_scale_pixel_decoder", \
            "DVIS++ / Mask2Former configs feed the decoder from the multi-scale pixel decoder"
        if input_shape is not None:
            self.in_features = [k for k, _ in sorted(input_shape.items(), key=lambda x: x[1].stride)]
        self.pixel_decoder, self.predictor = pixel_decoder, transformer_predictor
        self.num_classes, self.ignore_value, self.loss_weight = num_classes, ignore_value, loss_weight
        self.transformer_in_feature = transformer_in_feature
        self.return_transformer_feature = return_transformer_feature
        self.common_stride = 4

    @classmethod
    def from_config(cls, cfg, input_shape):
        """mask_former_head.py:88-116 (same keys; TRANSFORMER_IN_FEATURE selects the decoder's input width)."""
        hd, mf = cfg.MODEL.SEM_SEG_HEAD, cfg.MODEL.MASK_FORMER
        feat = mf.TRANSFORMER_IN_FEATURE
        in_channels = hd.MASK_DIM if feat == "pixel_embedding" else hd.CONVS_DIM \
            if feat in ("transformer_encoder", "multi_scale_pixel_decoder") else input_shape[feat].channels
        return {
            "input_shape": {k: v for k, v in input_shape.items() if k in hd.IN_FEATURES},
            "ignore_value": _get(hd, "IGNORE_VALUE", 255),
            "return_transformer_feature": _get(hd, "RETURN_TRANSFORMER_FEATURE", False),
            "num_classes": hd.NUM_CLASSES,
            "pixel_decoder": d2.build_pixel_decoder(cfg, input_shape),
            "loss_weight": _get(hd, "LOSS_WEIGHT", 1.0),
            "transformer_in_feature": feat,
            "transformer_predictor": d2.build_transformer_decoder(cfg, in_channels, mask_classification=True),
        }

    @Fn.fp32_island
    def forward(self, features, mask=None):
        mask_features, _, multi_scale_features = self.pixel_decoder.forward_features(features)
        return self.predictor(multi_scale_features, mask_features, mask)


def segmenter_frames_per_call(n, H, W, requested=0):
    """Frames per segmenter call for n frames (requested: the user's chunk, 0 = all of them in one call), cut into equal
    shares.  (Rounds 1-2 capped a call at 4 GiB per activation — 55 frames at 720p — because two segmenter passes in
    flight on two streams stopped making progress beyond that.  The cause was two hipBLASLt stream-K GEMMs running
    concurrently; stream() no longer puts a library GEMM on its second stream, so the cap is gone:
    a 64-frame clip is one call.)"""
    chunk = min(requested or max(1, n), max(1, n))
    if 0 < chunk < n:
        calls = (n + chunk - 1) // chunk
        chunk = (n + calls - 1) // calls          # equal shares
    return max(1, chunk)


class _VideoBase(nn.Module):
    """Constructor surface of the reference's meta-architectures (dvis_Plus/meta_architecture.py:29-90, 409-500,
    1073-1160): ``Cls(cfg)`` through ``from_config`` like detectron2's ``build_model`` does, or explicit keyword
    arguments — exactly the reference's names; its training-only ones (criterion, num_frames, max_iter_num, use_cl) are
    accepted and ignored, anything else is a TypeError;
    ``metadata`` supplies the thing classes (``thing_dataset_id_to_contiguous_id``), ``n_things`` is the short form
    for datasets whose thing classes are 0..n-1 (VIPSeg)."""

    @configurable
    def __init__(self, *, backbone, sem_seg_head, num_queries, object_mask_threshold=0.8, overlap_threshold=0.8,
                 n_things=0, size_divisibility=32, pixel_mean=(123.675, 116.280, 103.530),
                 pixel_std=(58.395, 57.120, 57.375), tracker=None, refiner=None, task="vis", max_num=20,
                 window_size=3, segmenter_chunk=0, metadata=None, criterion=None,
                 sem_seg_postprocess_before_inference=True, num_frames=1, window_inference=True, max_iter_num=0,
                 use_cl=False, reference_outputs=False):
        super().__init__()
        self.backbone, self.sem_seg_head, self.tracker, self.refiner = backbone, sem_seg_head, tracker, refiner
        self.num_queries = num_queries
        self.object_mask_threshold, self.overlap_threshold = object_mask_threshold, overlap_threshold
        self.metadata = metadata
        ids = d2.thing_ids_from_metadata(metadata, video=True)      # meta_architecture.py:919: cls < len(thing table)
        self.thing_ids = frozenset(range(int(n_things))) if ids is None else ids
        self.num_frames, self.window_inference = num_frames, window_inference
        self.size_divisibility = size_divisibility
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        assert task in ("vis", "vss", "vps")
        self.task, self.max_num = task, max_num
        # True: forward() / stream() return the reference's output format (python lists, CPU tensors: what its evaluators
        # consume, postprocess.to_reference_format); False: everything stays on the device.  `Cls(cfg)` sets True.
        self.reference_outputs = bool(reference_outputs)
        self.window_size = window_size        # reference knob (TEST.WINDOW_SIZE); results do not depend on it
        self.segmenter_chunk = segmenter_chunk  # frames per segmenter call, 0 = whole (local) clip at once
        self.keep = False
        self._clip_shard = None
        # offline mode: spans of the clip handed to the tracker while the segmenter works on the next span
        self.pipeline_rounds = int(os.environ.get("DVIS_PIPELINE_ROUNDS", "1"))
        self._tracker_stream = None
        # stream(): clips go in rounds of `world`, each clip's tracker + refiner on its own rank (DVIS_OWNER_ROUNDS=0:
        # one clip per round, tracker replicated on every rank)
        self.owner_rounds = os.environ.get("DVIS_OWNER_ROUNDS", "1") != "0"
        # stream() with the tracker replicated (single GPU, or owner rounds off): clips per round whose tracker recurrences
        # advance TOGETHER in one pass (same results per clip; the recurrence's ~65 launch-bound kernels per frame are paid
        # once for the whole round).  Per-clip latency grows by (tracker_batch - 1) segmenter passes.
        self.tracker_batch = max(1, int(os.environ.get("DVIS_TRACKER_BATCH", "1")))
        # stream(): phase B on its own host thread (DVIS_STREAM_THREAD=1).  Off by default: measured per rank with
        # tools/rank_emulation.py it buys 0 - 3 % — the serial host is not what keeps the two phases from overlapping
        self.stream_thread = os.environ.get("DVIS_STREAM_THREAD", "0") == "1"
        self.stream_timing = False            # stream(): make the per-clip "ready_event" a timing event (bench latency)
        # bench.py only: let a clip's input dict carry its own calibrated "object_mask_threshold" (random-init class
        # scores are near-uniform).  Off by default: the reference's input dicts have no such key.
        self.allow_input_threshold = False
        self.debug_stages = None              # tests: a dict here receives the floats behind the last clip's decisions
        self._seg_graph = None                # online mode: hipGraph of the segmenter per window shape (_segmenter_graph_ok)
        if hasattr(self.sem_seg_head.predictor, "compute_pred_masks"):
            self.sem_seg_head.predictor.compute_pred_masks = False

    @property
    def device(self):
        return self.pixel_mean.device

    @property
    def n_things(self):
        return len(self.thing_ids)

    @classmethod
    def _common_from_config(cls, cfg):
        """Keys every video meta-architecture of the reference reads (meta_architecture.py:91-181, 502-588, 1161-1262),
        minus the training criterion."""
        mf = cfg.MODEL.MASK_FORMER
        backbone = d2.build_backbone(cfg)
        sem_seg_head = d2.build_sem_seg_head(cfg, backbone.output_shape())
        test = mf.TEST
        return {
            "backbone": backbone, "sem_seg_head": sem_seg_head, "criterion": None,
            "num_queries": mf.NUM_OBJECT_QUERIES,
            "object_mask_threshold": test.OBJECT_MASK_THRESHOLD, "overlap_threshold": test.OVERLAP_THRESHOLD,
            "metadata": d2.dataset_metadata(cfg),
            "size_divisibility": mf.SIZE_DIVISIBILITY, "sem_seg_postprocess_before_inference": True,
            "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD,
            "num_frames": cfg.INPUT.SAMPLING_FRAME_NUM, "window_inference": _get(test, "WINDOW_INFERENCE", False),
            "task": _get(test, "TASK", "vis"), "max_num": _get(test, "MAX_NUM", 20),
            "window_size": _get(test, "WINDOW_SIZE", 3),
            "reference_outputs": True,            # built the detectron2 way: drop-in for train_net_video.py's evaluators
        }

    @staticmethod
    def _tracker_from_config(cfg):
        from .tracker import ReferringTracker_noiser
        mf = cfg.MODEL.MASK_FORMER
        hidden = mf.HIDDEN_DIM * (2 if _get(mf, "REID_BRANCH", True) else 1)               # meta_architecture.py:550-553
        trk = cfg.MODEL.TRACKER
        return ReferringTracker_noiser(hidden_channel=hidden, feedforward_channel=mf.DIM_FEEDFORWARD,
                                       num_head=mf.NHEADS, decoder_layer_num=trk.DECODER_LAYERS,
                                       noise_mode=_get(trk, "NOISE_MODE", "none"),
                                       noise_ratio=_get(trk, "NOISE_RATIO", 0.5), mask_dim=mf.HIDDEN_DIM,
                                       class_num=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES), hidden

    @staticmethod
    def _new_tracker_stream():
        """The stream phase B runs on.  Its kernels are few-workgroup links of a strictly sequential chain; next to phase A's
        device-filling kernels each of them takes 4-6x its stand-alone time (profiles/r03_bench_kernel_stats.csv) — invisible
        while phase A is 140 ms long, the critical path when it is 20 ms (a rank of an 8-GPU job).  Measured and NOT the
        answer (profiles/r03_rank_emulation_matrix.txt): a high-priority stream (DVIS_SIDE_PRIORITY=-1:
        140 -> 162 ms per clip on one GPU) and disjoint compute-unit masks for the two streams (hipExtStreamCreateWithCUMask,
        16 / 32 / 64 CUs for phase B: 219 - 232 ms)."""
        pr = int(os.environ.get("DVIS_SIDE_PRIORITY", "0"))
        return torch.cuda.Stream(priority=pr)

    @property
    def clip_shard(self):
        """Frame sharding follows the default process group of the moment (none -> single GPU)."""
        if isinstance(self._clip_shard, EmulatedShard):        # tools/rank_emulation.py
            return self._clip_shard
        world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        if self._clip_shard is None or self._clip_shard.world != world:
            self._clip_shard = ClipShard()
        return self._clip_shard

    # ---- meta_architecture.py:1306-1311: normalise, then pad bottom/right to a multiple of size_divisibility
    def preprocess(self, frames):
        x = frames if torch.is_tensor(frames) else torch.stack([f.to(self.device) for f in frames])
        H, W = x.shape[-2:]
        d = self.size_divisibility
        Hp, Wp = ((H + d - 1) // d * d, (W + d - 1) // d * d) if d > 1 else (H, W)
        x = x.to(self.device)
        if x.is_cuda and x.dim() == 4 and not torch.is_grad_enabled():
            # one pass on the device (csrc/fused_elementwise.hip); other integer / half dtypes convert to fp32 exactly first
            exact = x.dtype in (torch.int8, torch.int16, torch.float16, torch.bfloat16, torch.bool)    # (float64 keeps torch's path)
            x = (x.to(torch.float32) if exact else x).contiguous()
        if Fn.normalize_pad_ok(x, self.pixel_mean):
            return Fn.normalize_pad(x, self.pixel_mean, self.pixel_std, Hp, Wp), (H, W)
        x = (x.to(torch.float32) - self.pixel_mean) / self.pixel_std
        if (Hp, Wp) != (H, W):
            x = torch.nn.functional.pad(x, (0, Wp - W, 0, Hp - H))
        return x, (H, W)

    def _reserve_scope(self, stage):
        """Inside stream(): the persistent split-f16 grids of `stage` leave `_stream_reserve` CUs to the side stream when the
        reserve is scoped to that stage (DVIS_X3_RESERVE_SCOPE=backbone; development).  Default "phase_a": the whole of phase A
        leaves the CUs free.  Measured in round 6 (profiles/r06_reserve_scope.txt): phase B of the previous round starts on the
        device when this round's phase A does, so scoping the reserve to the 22 ms backbone looked free — it is not: 380 against
        388 frames/s (phase B's host-paced chain outlasts the backbone, and its few-workgroup kernels then queue behind
        full-device grids)."""
        import contextlib
        r = getattr(self, "_stream_reserve_now", 0)
        if not r or self._reserve_scope_name != stage:
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def scope():
            prev = native.lib().dvis_x3_set_reserve(r)
            try:
                yield
            finally:
                native.lib().dvis_x3_set_reserve(prev)
        return scope()

    _reserve_scope_name = os.environ.get("DVIS_X3_RESERVE_SCOPE", "phase_a")

    def encode(self, images):
        """Backbone + pixel decoder over this rank's frames: (multi_scale_features, mask_features (t,Cm,h,w))."""
        chunk = segmenter_frames_per_call(len(images), images.shape[-2], images.shape[-1], self.segmenter_chunk)
        ms, mf = [], []
        for s in range(0, len(images), chunk):
            with Fn.x3_stage("backbone"), self._reserve_scope("backbone"):
                feats = self.backbone(images[s:s + chunk])
            f, _, m = self.sem_seg_head.pixel_decoder.forward_features(feats)
            mf.append(f)
            ms.append(m)
        if len(mf) == 1:
            return ms[0], mf[0]
        return [torch.cat(level, 0) for level in zip(*ms)], torch.cat(mf, 0)

    def decode(self, multi_scale, mask_features):
        """Masked-attention decoder over a batch of frames: per-frame queries (t,Q,2C) normed / un-normed, logits."""
        if len(mask_features) == 0:                # a rank without frames in this span (T < world * k)
            pred = self.sem_seg_head.predictor
            C2, Q = self.tracker.decoder_norm.weight.shape[0], self.num_queries
            z = lambda *shape: mask_features.new_zeros(shape)
            return z(0, Q, C2), z(0, Q, C2), z(0, Q, pred.class_embed.out_features)
        out = self.sem_seg_head.predictor(multi_scale, mask_features, None)
        return (out["pred_embds"][0].permute(1, 2, 0), out["pred_embds_without_norm"][0].permute(1, 2, 0),
                out["pred_logits"][0])

    def segment(self, images):
        """Segmenter over this rank's frames.  Returns per-frame queries (t,Q,·) and mask_features (t,Cm,h,w)."""
        ms, mf = self.encode(images)
        return (*self.decode(ms, mf), mf)

    def _segmenter_graph_ok(self, images):
        """hipGraph replay of the segmenter for small windows (online mode: a 5-frame window is ~450 launches for ~20 ms of
        kernels — the device waits for the host).  DVIS_SEGMENTER_GRAPH=0 switches it off, =N sets the largest window (frames)
        that is captured (default 8: a graph keeps its private memory pool alive).
        (Round 5 withdrew this: replays "went wrong after a tracker call".  Cause, round 6: the attention masks' allowed_count
        was zeroed with hipMemsetAsync, and a memset node replays correctly only ONCE on this ROCm (garbage fill afterwards) — csrc/dvis_common.h
        dvis_zero_words; every replay after the first accumulated on stale counts.)"""
        limit = int(os.environ.get("DVIS_SEGMENTER_GRAPH", "8"))
        ok = bool(images.is_cuda and 0 < len(images) <= limit and not torch.is_grad_enabled() and self.debug_stages is None
                  and getattr(self.sem_seg_head.predictor, "debug_masks", None) is None)
        if ok and self._seg_graph is None:
            from .graphs import GraphRunner
            self._seg_graph = GraphRunner(lambda im: tuple(self.segment(im)), max_entries=4)
        return ok

    # ---- range guard of the split-f16 kernels (functions._X3RangeGuard): snapshot behind phase A, verify before its results
    # are used.  The reference's fp32 island (msdeformattn.py:314,320) has no range to leave; here leaving it is an error (or
    # a re-run on the exact kernels), never silently wrong masks.
    def _guard_snapshot(self):
        return Fn.x3_range_snapshot(self.device) if not getattr(self, "_x3_off", False) else None

    def _guard_verify(self, snap):
        Fn.x3_range_verify(snap, self)

    def _gather_guarded(self, st):
        """The clip's ONE all-gather of the per-frame queries.  Unsharded: the local range-guard check, then nothing to gather.
        Sharded: the rank's guard word travels IN the gather and the check runs on the gathered tags — every rank raises (or
        none): a rank that raised alone would leave the others waiting in the next collective."""
        shard = self.clip_shard
        sharded = shard.world > 1 or shard.force
        snap = st.get("guard")
        if not sharded:
            self._guard_verify(snap)
        out = shard.all_gather_frames([st["embds"], st["embds_nn"], st["logits"]], st["T"], shift=st["shift"],
                                      guard=snap[3] if sharded and snap is not None else None)
        if sharded:
            Fn.X3_GUARD.verify_gathered(shard.last_guard_tags, self.device, self)
        return out

    def _x3_rerun(self, err, run):
        """X3RangeError policy of a single-GPU call: re-run `run()` on the exact-fp32 kernels (and stay on them), or raise."""
        sharded = self.clip_shard.world > 1 or self.clip_shard.force
        if sharded or Fn.X3_ON_OVERFLOW != "rerun":
            raise err
        import warnings
        warnings.warn(f"{err}  Re-running the clip on the exact-fp32 kernels; this model stays on them (DVIS_X3_ON_OVERFLOW=raise "
                      "turns this into an error).", RuntimeWarning, stacklevel=3)
        self._x3_off = True
        return run()

    def _x3_scope(self):
        """Context for a model call: the exact kernels once a range error was seen (`_x3_off`), else nothing."""
        import contextlib
        return Fn.x3_disabled() if getattr(self, "_x3_off", False) else contextlib.nullcontext()

    def _task_output(self, cls, aux, mask_fn, img_size, out_hw, padded_size, T_local, video=None):
        """video: the input dict; an optional "object_mask_threshold" entry overrides the model's for this clip (used by
        bench.py: random-init class scores are near-uniform, the threshold is calibrated per synthetic clip)."""
        K = self.sem_seg_head.num_classes
        thr = self.object_mask_threshold
        if video is not None and self.allow_input_threshold:
            thr = video.get("object_mask_threshold", thr)
        if self.task == "vis":
            return PP.inference_video_vis(cls, mask_fn, img_size, out_hw, padded_size, K, self.max_num, aux)
        if self.task == "vps":
            return PP.inference_video_vps(cls, mask_fn, img_size, out_hw, padded_size, K, self.thing_ids,
                                          thr, self.overlap_threshold, aux,
                                          num_frames=T_local, reduce_fn=self.clip_shard.all_reduce_sum)
        return PP.inference_video_vss(cls, mask_fn, img_size, out_hw, padded_size, aux)


@META_ARCH_REGISTRY.register()
class MinVIS(_VideoBase):
    """MinVIS (dvis_Plus/meta_architecture.py:23-407, eval): per-frame segmenter, frame-by-frame Hungarian alignment of
    the decoder embeddings (match_from_embds :255-264 — cosine WITHOUT the tracker's 1e-6), logits averaged over the
    aligned frames, top-10 (query, class) pairs.  Takes the `_minvis` decoder.

    Same results, different schedule: every frame's cost matrix against its predecessor comes from one batched GEMM and
    the alignment recurrence runs in one host call (dvis_match_chain — permuting the previous frame only permutes the
    columns of the cost matrix); masks are contracted for the 10 selected query slots only."""
    TOPK = 10      # hard-coded in the reference (:371)

    @classmethod
    def from_config(cls, cfg):
        """meta_architecture.py:91-181."""
        ret = cls._common_from_config(cfg)
        ret["task"] = "vis"
        return ret

    @torch.no_grad()
    @Fn.fp32_island
    def forward(self, batched_inputs):
        assert len(batched_inputs) == 1 and not self.training
        from . import functions as Fn
        from .tracker import match_chain
        video = batched_inputs[0]
        images, img_size = self.preprocess(video["image"])
        pred = self.sem_seg_head.predictor
        with self._x3_scope():
            ms, mask_features = self.encode(images)
            dec, logits, _ = pred._final_heads(pred._run_layers(ms, mask_features), mask_features, False)   # (T,Q,C), (T,Q,K+1)
        try:
            self._guard_verify(self._guard_snapshot())
        except Fn.X3RangeError as e:
            return self._x3_rerun(e, lambda: self.forward(batched_inputs))
        T, Q, _ = dec.shape
        # ---- alignment chain: frame 0 keeps its order, frame t is matched to the ALIGNED frame t-1
        idx = torch.arange(Q, device=dec.device).unsqueeze(0).repeat(T, 1)
        if T > 1:
            nrm = dec / dec.norm(dim=-1, keepdim=True)
            cost = 1 - torch.bmm(nrm[1:], nrm[:-1].transpose(1, 2))                    # (T-1, Q, Q): cur x previous
            idx[1:] = torch.from_numpy(match_chain(cost)).to(dec.device)
        ar = torch.arange(T, device=dec.device)[:, None]
        cls = logits[ar, idx].mean(0)                                                   # (Q, K+1), aligned
        K = self.sem_seg_head.num_classes
        scores, labels, slot = PP.vis_select(cls, K, self.TOPK)
        emb = pred.mask_embed(dec[ar, idx[:, slot]])                                    # (T, k, Cm): the slots' queries
        masks = Fn.mask_logits(emb.contiguous(), mask_features).permute(1, 0, 2, 3)     # (k, T, h, w)
        out_hw = (video.get("height", img_size[0]), video.get("width", img_size[1]))
        masks = PP.resize2_gt0(masks, images.shape[-2:], img_size, out_hw)
        out = {"image_size": tuple(out_hw), "pred_scores": scores.tolist(), "pred_labels": labels.tolist(),
               "pred_masks": [m for m in masks], "pred_ids": slot.tolist(), "aligned_indices": idx}
        return PP.to_reference_format(out) if self.reference_outputs else out


@META_ARCH_REGISTRY.register()
class DVIS_Plus_online(_VideoBase):
    """Segmenter + referring tracker; masks come from the tracker (projected mask features)."""

    @classmethod
    def from_config(cls, cfg):
        """meta_architecture.py:502-588."""
        ret = cls._common_from_config(cfg)
        ret["tracker"], _ = cls._tracker_from_config(cfg)
        ret["max_iter_num"] = _get(_get(cfg, "SOLVER", {}), "MAX_ITER", 0)
        ret["use_cl"] = _get(cfg.MODEL.TRACKER, "USE_CL", False)
        return ret

    @torch.no_grad()
    @Fn.fp32_island
    def forward(self, batched_inputs):
        assert len(batched_inputs) == 1 and not self.training
        video = batched_inputs[0]
        self.keep = bool(video.get("keep", False))
        images, img_size = self.preprocess(video["image"])
        with self._x3_scope():
            # small windows replay the segmenter from a hipGraph captured per (frames, size, kernel switches); its outputs are
            # static buffers, consumed before this call returns
            if self._segmenter_graph_ok(images):
                embds, embds_nn, logits, mask_features = self._seg_graph(
                    (tuple(images.shape), getattr(self, "_x3_off", False), Fn.X3, Fn.X3_OFF), images)
            else:
                embds, embds_nn, logits, mask_features = self.segment(images)
        try:
            self._guard_verify(self._guard_snapshot())     # (the tracker's host-side assignment waits for the segmenter anyway)
        except Fn.X3RangeError as e:
            return self._x3_rerun(e, lambda: self.forward(batched_inputs))
        to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
        track = self.tracker(to_bctq(embds), mask_features.unsqueeze(0), resume=self.keep,
                             frame_embeds_no_norm=to_bctq(embds_nn), need_masks=False)
        cls, _ = PP.mean_logits(track["pred_logits"])
        dec = self.tracker.decoder_norm(track["pred_embds"][0].permute(1, 2, 0))   # (T, Q, C)
        emb = self.tracker.mask_embed(dec)
        proj = self.tracker.project_mask_features(mask_features)

        def mask_fn(idx):
            from . import functions as Fn
            e = emb if idx is None else emb[:, idx]
            return Fn.mask_logits(e.contiguous(), proj).permute(1, 0, 2, 3)
        if self.debug_stages is not None:
            self.debug_stages.update(mask_fn=mask_fn, cls=cls, aux=None)
        out_hw = (video.get("height", img_size[0]), video.get("width", img_size[1]))
        out = self._task_output(cls, None, mask_fn, img_size, out_hw, images.shape[-2:], len(images), video)
        return PP.to_reference_format(out) if self.reference_outputs else out


@META_ARCH_REGISTRY.register()
class DVIS_Plus_offline(_VideoBase):
    """Segmenter + referring tracker + temporal refiner (the north-star path)."""

    @classmethod
    def from_config(cls, cfg):
        """meta_architecture.py:1161-1262."""
        from .refiner import TemporalRefiner
        ret = cls._common_from_config(cfg)
        ret["tracker"], hidden = cls._tracker_from_config(cfg)
        mf = cfg.MODEL.MASK_FORMER
        ret["refiner"] = TemporalRefiner(hidden_channel=hidden, feedforward_channel=mf.DIM_FEEDFORWARD,
                                         num_head=mf.NHEADS, decoder_layer_num=cfg.MODEL.REFINER.DECODER_LAYERS,
                                         mask_dim=mf.HIDDEN_DIM, class_num=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
                                         windows=_get(mf.TEST, "WINDOW_SIZE", 3))
        ret["max_iter_num"] = _get(_get(cfg, "SOLVER", {}), "MAX_ITER", 0)
        ret["use_cl"] = _get(cfg.MODEL.REFINER, "USE_CL", False)
        return ret

    # ---- the clip as two phases: everything up to the per-frame queries is asynchronous and rank-local (phase A);
    # everything after needs the other ranks' queries, host-side assignment and the VPS statistics (phase B).
    @torch.no_grad()
    def _segment_phase(self, video, shift=0):
        """Phase A on the current stream: this rank's frames through backbone, pixel decoder and decoder.  No host sync,
        no collective.  `shift` rotates the block -> rank assignment (ClipShard.local_range)."""
        return self._segment_round([video], shift)[0]

    @torch.no_grad()
    def _segment_round(self, videos, shift=0, rotate=True):
        """Phase A of a round of clips (clip j sharded with rotation shift + j).  The segmenter treats frames as a
        batch, so this rank's frames of ALL clips of the round go through it in ONE call when their padded sizes agree
        (8 ranks x 30-frame clips: one 30-frame batch per round instead of eight 4-frame calls — 158.7 vs 183 frames/s
        per GPU in the small-batch measurement of DESIGN.md section 7); otherwise clip by clip."""
        metas, batches = [], []
        for j, video in enumerate(videos):
            frames = video["image"]
            T = len(frames)
            sh = shift + j if rotate else shift
            lo, hi = self.clip_shard.local_range(T, sh)
            images, img_size = self.preprocess(frames[lo:hi] if hi > lo else frames[:1])
            metas.append(dict(video=video, T=T, lo=lo, hi=hi, shift=sh, img_size=img_size,
                              padded=tuple(images.shape[-2:])))
            batches.append(images if hi > lo else images[:0])
        mask_dim = self.sem_seg_head.predictor.mask_embed.layers[-1].out_features
        # (one rank: every clip already is a full batch — separate calls keep a clip's segmenter bits independent of its
        # round mates; DVIS_ROUND_CLIPS, a development aid, forces the merged batch on a single GPU)
        merged = len(videos) > 1 and len({m["padded"] for m in metas}) == 1 and sum(len(b) for b in batches) > 0 \
            and (self.clip_shard.world > 1 or self.clip_shard.force or "DVIS_ROUND_CLIPS" in os.environ)

        def run(images):
            if len(images):
                return self.segment(images)
            mf = images.new_zeros((0, mask_dim, images.shape[-2] // 4, images.shape[-1] // 4))
            return (*self.decode(None, mf), mf)
        with self._x3_scope():
            if merged:
                outs = run(torch.cat(batches, 0))
                sizes = [len(b) for b in batches]
                parts = [o.split(sizes, 0) for o in outs]
                per_clip = [tuple(p[j] for p in parts) for j in range(len(videos))]
            else:
                per_clip = [run(b) for b in batches]
        guard = self._guard_snapshot()
        for m, (e, e_nn, lg, mf) in zip(metas, per_clip):
            m.update(embds=e, embds_nn=e_nn, logits=lg, mf=mf, guard=guard)
            if self.debug_stages is not None:
                self.debug_stages.update(mask_features=mf)
        return metas

    @property
    def _resume(self):
        """Does this clip continue the tracker state of the previous call?  The reference's offline model reads `keep`
        (meta_architecture.py:1301-1304) but its window loop only resumes for windows i != 0 (:1479-1486), so with
        TEST.WINDOW_INFERENCE (every shipped config) `keep` has no effect; its non-window branch passes
        resume=self.keep (:1329-1334).  Same here."""
        return bool(self.keep) and not self.window_inference

    def _track_core(self, embds, embds_nn):
        """Tracker + refiner over the T gathered frames of one clip: (mask_embed (1,T,Q,Cm), cls (Q,K+1), aux (Q,K+1))."""
        to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
        # one canonical memory layout, whatever the caller holds (a column slice of the all-gather's packed buffer, a copy
        # of it, a decoder output): torch's reductions pick their vectorisation from the strides, so the SAME values in another
        # layout give other last bits (1e-5 on the refined embeddings, tools/stream_shard_check.py) — and the replicated ranks
        # of north_star's split must agree bit for bit
        embds, embds_nn = embds.contiguous(), embds_nn.contiguous()
        track = self.tracker(to_bctq(embds), None, resume=self._resume, frame_embeds_no_norm=to_bctq(embds_nn),
                             need_masks=False)
        ref = self.refiner(track["pred_embds"], to_bctq(embds_nn), None, need_masks=False)
        cls, aux = PP.mean_logits(ref["pred_logits"], track["pred_logits"])
        if self.debug_stages is not None:      # tests / error budget: the floats of every stage, in the reference's layouts
            self.debug_stages.update(frame_embds=to_bctq(embds), frame_embds_no_norm=to_bctq(embds_nn),
                                     instance_embds=track["pred_embds"], online_logits=track["pred_logits"],
                                     refiner_embds=ref["pred_embds"], refiner_logits=ref["pred_logits"],
                                     refiner_mask_embed=ref["mask_embed"])
        return ref["mask_embed"], cls, aux

    def _finish_phase(self, st, mask_embed, cls, aux):
        """Masks of this rank's frames + post-processing (VPS segment areas are summed over the ranks)."""
        video, lo, hi = st["video"], st["lo"], st["hi"]
        emb_local = mask_embed[:, lo:hi]                                            # (1, t_local, Q, Cm)
        mf = st["mf"].unsqueeze(0)

        def mask_fn(idx):
            return self.refiner.predict_masks(emb_local, mf, idx)[0]                # (q', t_local, h, w)
        if self.debug_stages is not None:
            self.debug_stages.update(mask_fn=mask_fn, cls=cls, aux=aux)
        img_size = st["img_size"]
        out_hw = (video.get("height", img_size[0]), video.get("width", img_size[1]))
        out = self._task_output(cls, aux, mask_fn, img_size, out_hw, st["padded"], hi - lo, video)
        out["frame_ids"] = list(range(lo, hi))
        out["frame_range"] = (lo, hi)
        return out        # (forward() / stream() convert to the reference's format when reference_outputs is set)

    @torch.no_grad()
    def _track_phase(self, st):
        """Phase B on the current stream: ONE all-gather of the per-frame queries, tracker + refiner replicated on every
        rank, masks of this rank's frames, post-processing (VPS: one tiny all-reduce of the segment areas)."""
        self.keep = bool(st["video"].get("keep", False))
        embds, embds_nn, _ = self._gather_guarded(st)
        # Replicated tracker + refiner need NO broadcast: every rank holds the same gathered queries, the host assignment
        # is deterministic, and since round 3 phase B runs no library kernel (own deterministic GEMM, attention, add+LN)
        # — same bits on every rank, so the post-processing decisions agree.  north_star's "single all-gather" is
        # literally that (+ the few-hundred-byte VPS area sum).
        mask_embed, cls, aux = self._track_core(embds, embds_nn)
        self.clip_shard.check_replicas([mask_embed, cls, aux], "tracker / refiner results")   # DVIS_CHECK_REPLICAS=1 only
        return self._finish_phase(st, mask_embed, cls, aux)

    def _resume_of(self, st):
        return bool(st["video"].get("keep", False)) and not self.window_inference

    @torch.no_grad()
    def _track_phase_batched(self, sts):
        """Phase B of several clips of equal length with the tracker REPLICATED: one all-gather per clip as in
        _track_phase, then ONE tracker pass in which the clips' recurrences advance together (ReferringTracker_noiser with
        batch = number of clips: every op of the recurrence is row-wise or per (batch, head), the GEMMs' tile configuration is
        pinned to a single clip's — same bits per clip as alone), ONE refiner pass over the clips of the round (same argument),
        then masks and post-processing clip by clip."""
        to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
        self.keep = False
        gathered = [self._gather_guarded(st) for st in sts]
        embds = torch.cat([to_bctq(g[0]) for g in gathered], 0)                      # (clips, 2C, T, Q)
        embds_nn = torch.cat([to_bctq(g[1]) for g in gathered], 0)
        track = self.tracker(embds, None, resume=False, frame_embeds_no_norm=embds_nn, need_masks=False)
        ref = self.refiner(track["pred_embds"], embds_nn, None, need_masks=False)      # ... and ONE refiner pass
        self.clip_shard.check_replicas([ref["mask_embed"], ref["pred_logits"], track["pred_logits"]],
                                       "tracker / refiner results (batched pass)")
        outs = []
        for j, st in enumerate(sts):
            cls, aux = PP.mean_logits(ref["pred_logits"][j:j + 1], track["pred_logits"][j:j + 1])
            if self.debug_stages is not None:
                self.debug_stages.update(instance_embds=track["pred_embds"][j:j + 1],
                                         refiner_embds=ref["pred_embds"][j:j + 1])
            outs.append(self._finish_phase(st, ref["mask_embed"][j:j + 1], cls, aux))
        return outs

    @torch.no_grad()
    def _track_round(self, sts):
        """Phase B of a round of up to `world` clips with ONE OWNER per clip: the tracker's recurrence and the refiner
        are strictly sequential small kernels + host-side assignment (23 ms per 30-frame clip on MI355X) that frame
        sharding cannot split, so replicated on every rank they cap the strong-scaling speed-up at T_clip / 23 ms.  Here
        clip j of the round is tracked and refined by rank j alone — in parallel with the other ranks' clips — and one
        more all-gather hands every rank the refined mask embeddings + class logits of all clips of the round (3 MB per
        clip); masks and post-processing stay sharded by frame.  Collectives per round, same order on every rank: one
        all-gather per clip (per-frame queries), one all-gather of the results, then the post-processing reductions.
        The results come from a single rank, so they are bit-identical everywhere without the broadcast of
        _track_phase.  Clips that resume the tracker state of a previous call (`keep`) take the replicated path."""
        shard = self.clip_shard
        m = len(sts)
        if not self.owner_rounds or (shard.world == 1 and not shard.force) \
                or (not self.window_inference and any(bool(st["video"].get("keep", False)) for st in sts)):
            if m > 1 and len({st["T"] for st in sts}) == 1 and not any(self._resume_of(st) for st in sts):
                return self._track_phase_batched(sts)
            return [self._track_phase(st) for st in sts]
        assert m <= shard.world
        gathered = [self._gather_guarded(st) for st in sts]
        Q, K1 = self.num_queries, sts[0]["logits"].shape[-1]
        Cm = self.refiner.mask_embed.layers[-1].out_features
        C2 = self.tracker.decoder_norm.weight.shape[0]
        Tmax = max(st["T"] for st in sts)
        n_emb, n_cls, n_state = Tmax * Q * Cm, Q * K1, Q * C2
        pack = sts[0]["mf"].new_zeros(n_emb + 2 * n_cls + 3 * n_state)
        if shard.rank < m:
            self.keep = False
            mask_embed, cls, aux = self._track_core(gathered[shard.rank][0], gathered[shard.rank][1])
            pack[:mask_embed.numel()] = mask_embed.reshape(-1)
            pack[n_emb:n_emb + n_cls] = cls.reshape(-1)
            pack[n_emb + n_cls:n_emb + 2 * n_cls] = aux.reshape(-1)
            trk = self.tracker
            for i, t in enumerate((trk.last_outputs, trk.last_reference, trk.last_frame_embeds)):
                o = n_emb + 2 * n_cls + i * n_state
                pack[o:o + n_state] = t.reshape(-1)
        rows = shard.all_gather_rows(pack)                                           # (world, n_emb + 2 n_cls + state)
        # Every rank leaves the round with the tracker state of the round's LAST clip (its owner is rank m - 1): a later
        # clip that resumes (`keep`) takes the replicated path and must find the same predecessor state everywhere.
        st_row = rows[m - 1, n_emb + 2 * n_cls:]
        trk = self.tracker
        trk.last_outputs = st_row[:n_state].view(Q, 1, C2).clone()
        trk.last_reference = st_row[n_state:2 * n_state].view(Q, 1, C2).clone()
        trk.last_frame_embeds = st_row[2 * n_state:].view(Q, 1, C2).clone()
        outs = []
        for j, st in enumerate(sts):
            T = st["T"]
            outs.append(self._finish_phase(st, rows[j, :T * Q * Cm].view(1, T, Q, Cm),
                                           rows[j, n_emb:n_emb + n_cls].view(Q, K1).clone(),
                                           rows[j, n_emb + n_cls:n_emb + 2 * n_cls].view(Q, K1).clone()))
        return outs

    @torch.no_grad()
    def stream(self, videos):
        """Throughput mode for a sequence of clips: yields forward([v]) for every v in order.  Clips are taken in rounds
        of `world` (one clip per round on a single GPU); phase A of round i+1 is enqueued BEFORE phase B of round i
        blocks on its host-side steps (assignment chain, VPS statistics).  On the GPU phase B runs on a second stream,
        so the tracker's small, strictly sequential kernels fill in next to the next clips' backbone instead of owning
        the device, and with several ranks every clip of the round has its own tracker rank (_track_round).  Same
        results as calling forward clip by clip (same kernels, same order per clip; bit-identical from the per-frame
        queries onward: phase B is deterministic); per-clip latency is one round's phase A longer."""
        import itertools
        overlap = self.device.type == "cuda"
        main = torch.cuda.current_stream() if overlap else None
        if overlap and self._tracker_stream is None:
            self._tracker_stream = self._new_tracker_stream()
        side = self._tracker_stream if overlap else None
        # DVIS_ROUND_CLIPS: development aid — clips per round on a single GPU (exercises the merged segmenter batch)
        sharded_owner = self.owner_rounds and (self.clip_shard.world > 1 or self.clip_shard.force)
        per_round = int(os.environ.get("DVIS_ROUND_CLIPS", "0")) or (self.clip_shard.world if sharded_owner
                                                                     else self.tracker_batch)

        def phase_b(sts):
            """Always next to the following round's phase A: everything phase B enqueues is own code that never waits for
            another workgroup (attention, add+LayerNorm, mask contraction, post-processing kernels, and since round 3
            every GEMM: csrc/gemm.hip) or a torch element-wise / reduction kernel, so the library's stream-K GEMMs of
            phase A can never find themselves waiting for CUs held by a peer that waits for them (the two-stream stall
            of rounds 1-2).  tests/test_stream_gpu.py asserts that no GEMM / convolution library
            call is issued from here, and soaks T = 64 clips."""
            def track_round():
                try:
                    return self._track_round(sts)
                except Fn.X3RangeError as e:       # single GPU: the round's clips again, on the exact-fp32 kernels
                    return self._x3_rerun(e, lambda: [self._track_phase(self._segment_phase(st["video"])) for st in sts])
            if not overlap:
                return track_round()
            with torch.cuda.stream(side):
                for st in sts:
                    side.wait_event(st["done"])
                    for t in (st["embds"], st["embds_nn"], st["logits"], st["mf"]):
                        t.record_stream(side)                                       # allocated on the main stream
                outs = track_round()
                ready = torch.cuda.Event(enable_timing=self.stream_timing)
                ready.record(side)
            for out in outs:
                out["ready_event"] = ready
            return outs

        def hand_over(outs):
            """(consumer's thread) The outputs were produced (and allocated) on the side stream; the caller consumes them on
            ITS stream: order it behind phase B of the round and tell the allocator about the second stream."""
            if overlap:
                main.wait_event(outs[0]["ready_event"])
                for out in outs:
                    for v in out.values():
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(main)
            return [PP.to_reference_format(out) if self.reference_outputs else out for out in outs]

        # the split-f16 kernels of phase A are persistent (one workgroup per CU for the whole launch): leave 32 CUs (4 per XCD)
        # to phase B's small kernels on the side stream — measured 356 -> 365 - 370 frames/s (8 / 16 / 24 CUs lose, 40 - 64 gain less)
        # (set only WHILE a round's phase A is enqueued — the grid size is fixed at launch — and restored before this generator
        # yields: the caller's own kernels between two clips see the process-wide setting)
        self._stream_reserve = int(os.environ.get("DVIS_X3_RESERVE", "32")) if overlap else 0
        yield from self._stream_rounds(videos, per_round, sharded_owner, overlap, main, side, phase_b, hand_over)

    def _segment_round_reserved(self, chunk, shift, rotate):
        """_segment_round with the persistent split-f16 grids leaving `_stream_reserve` CUs to the side stream — for the stage
        `_reserve_scope` names, or for all of phase A."""
        r = getattr(self, "_stream_reserve", 0)
        if not r:
            return self._segment_round(chunk, shift=shift, rotate=rotate)
        if self._reserve_scope_name != "phase_a":
            self._stream_reserve_now = r
            try:
                return self._segment_round(chunk, shift=shift, rotate=rotate)
            finally:
                self._stream_reserve_now = 0
        prev = native.lib().dvis_x3_set_reserve(r)
        try:
            return self._segment_round(chunk, shift=shift, rotate=rotate)
        finally:
            native.lib().dvis_x3_set_reserve(prev)

    def _stream_rounds(self, videos, per_round, sharded_owner, overlap, main, side, phase_b, hand_over):
        import itertools
        if overlap and self.stream_thread:
            yield from self._stream_threaded(videos, per_round, sharded_owner, main, phase_b, hand_over)
            return

        it, prev, n = iter(videos), None, 0
        while True:
            chunk = list(itertools.islice(it, per_round))
            # Tracker-owner rounds: rotate the ragged split clip by clip (over a round of `world` clips every rank gets the
            # short blocks in turn: equal merged batches).  Tracker replicated (one clip per round, or `tracker_batch`
            # clips): no rotation — a clip takes ceil(T / world) frames of segmenter time whoever holds the short block,
            # and a fixed split means every rank keeps ONE batch shape (a new convolution shape costs a MIOpen solver
            # search, seconds).
            # (the fp32 island of forward(), entered per step: a context must not stay open across a `yield`)
            with Fn.no_autocast():
                sts = self._segment_round_reserved(chunk, n if sharded_owner else 0, sharded_owner) if chunk else []
            n += len(chunk)
            if overlap and sts:
                done = torch.cuda.Event()
                done.record(main)
                for st in sts:
                    st["done"] = done
            if prev is not None:
                with Fn.no_autocast():
                    outs = hand_over(phase_b(prev))
                yield from outs
            prev = sts or None
            if not sts:
                break
        if overlap:
            main.wait_stream(side)

    def _stream_threaded(self, videos, per_round, sharded_owner, main, phase_b, hand_over):
        """stream() with phase B on its own HOST thread (opt-in: `stream_thread`).  Phase B blocks the host for as long as
        the tracker's chain of small kernels and host-side assignments runs (~24 ms per 30-frame clip); with one host thread
        the next round's phase A can only be enqueued after that.  Here this thread only enqueues phase A (bounded: two rounds
        ahead) and hands finished rounds to the consumer; the worker runs phase B round by round in order — every collective
        of the pipeline is issued by it, in the same order on every rank.  The waits of phase B (event / stream
        synchronisation, the C assignment chain through ctypes) release the GIL.  Results are those of the single-threaded
        schedule: same kernels, same order per stream.  Measured (tools/rank_emulation.py, one rank of 8: 38.5 -> 37.4 ms per
        clip; owner rounds 19.2 -> 19.2): the host was not the limiter — phase B's kernels run 4-6x slower NEXT to phase A's
        than alone, whoever enqueues them — so this stays off."""
        import itertools
        import queue
        import threading
        q_in, q_out = queue.Queue(maxsize=2), queue.Queue()
        device = self.device

        def worker():
            torch.cuda.set_device(device)
            with torch.no_grad():
                while True:
                    sts = q_in.get()
                    if sts is None:
                        return
                    try:
                        q_out.put(("ok", phase_b(sts)))
                    except BaseException as e:      # noqa: BLE001 — re-raised in the consumer's thread
                        q_out.put(("err", e))
                        return
        th = threading.Thread(target=worker, name="dvis-phase-b", daemon=True)
        th.start()
        pending = 0

        def take(block):
            nonlocal pending
            kind, val = q_out.get(block=block)
            pending -= 1
            if kind == "err":
                pending = 0
                raise val
            return hand_over(val)
        try:
            it, n = iter(videos), 0
            while True:
                chunk = list(itertools.islice(it, per_round))
                if not chunk:
                    break
                with Fn.no_autocast():
                    sts = self._segment_round_reserved(chunk, n if sharded_owner else 0, sharded_owner)
                n += len(chunk)
                done = torch.cuda.Event()
                done.record(main)
                for st in sts:
                    st["done"] = done
                while True:                         # hand the round to the worker; meanwhile pass finished rounds on
                    try:
                        q_in.put(sts, timeout=0.002)
                        pending += 1
                        break
                    except queue.Full:
                        if not th.is_alive() and q_out.empty():
                            raise RuntimeError("stream(): the phase-B thread died")
                    while not q_out.empty():
                        yield from take(False)
                while not q_out.empty():
                    yield from take(False)
            while pending:
                yield from take(True)
        finally:
            try:
                while True:                         # (consumer stopped early / error: drop what is queued)
                    q_in.get_nowait()
            except queue.Empty:
                pass
            q_in.put(None)
            th.join()
            main.wait_stream(self._tracker_stream)

    @torch.no_grad()
    @Fn.fp32_island
    def forward(self, batched_inputs):
        assert len(batched_inputs) == 1 and not self.training
        video = batched_inputs[0]
        if self.pipeline_rounds <= 1:
            try:
                out = self._track_phase(self._segment_phase(video))
            except Fn.X3RangeError as e:
                out = self._x3_rerun(e, lambda: self._track_phase(self._segment_phase(video)))
            return PP.to_reference_format(out) if self.reference_outputs else out
        self.keep = bool(video.get("keep", False))
        frames = video["image"]
        T = len(frames)
        shard = self.clip_shard
        plan, k = shard.round_plan(T, self.pipeline_rounds)
        to_bctq = lambda z: z.permute(2, 0, 1).unsqueeze(0)
        overlap = len(plan) > 1 and self.device.type == "cuda"
        main = torch.cuda.current_stream() if overlap else None
        if overlap and self._tracker_stream is None:
            self._tracker_stream = self._new_tracker_stream()

        # backbone + pixel decoder once over all of this rank's frames (span-major order), at full batch efficiency;
        # only the decoder (cheap, batch-insensitive) is cut into spans
        local_ids = [i for (_, _, lo, hi) in plan for i in range(lo, hi)]
        contiguous = local_ids == list(range(local_ids[0], local_ids[0] + len(local_ids))) if local_ids else True
        if not local_ids:
            mine = frames[:0] if torch.is_tensor(frames) else []
        elif contiguous:
            mine = frames[local_ids[0]:local_ids[0] + len(local_ids)]
        else:
            mine = frames[torch.as_tensor(local_ids, device=frames.device)] if torch.is_tensor(frames) \
                else [frames[i] for i in local_ids]
        images, img_size = self.preprocess(mine if local_ids else frames[:1])
        if not local_ids:
            images = images[:0]
            pred = self.sem_seg_head.predictor
            multi_scale = None
            mask_features = images.new_zeros((0, pred.mask_embed.layers[-1].out_features, images.shape[-2] // 4,
                                              images.shape[-1] // 4))
        else:
            multi_scale, mask_features = self.encode(images)
        padded = images.shape[-2:]
        offsets, o = [], 0
        for (_, _, lo, hi) in plan:
            offsets.append((o, o + hi - lo))
            o += hi - lo

        def run_segmenter(c):
            """Span c's decoder on the main stream, then the (asynchronous) all-gather of its per-frame queries."""
            start, end, lo, hi = plan[c]
            a, b = offsets[c]
            whole = a == 0 and b == len(local_ids)
            ms = None if multi_scale is None else (multi_scale if whole else _slice_maps(multi_scale, a, b))
            e, e_nn, lg = self.decode(ms, mask_features[a:b])
            (e, e_nn, lg), work = shard.all_gather_frames([e, e_nn, lg], end - start, per=k, async_op=True)
            done = torch.cuda.Event() if overlap else None
            if overlap:
                done.record(main)
            return dict(embds=e, embds_nn=e_nn, logits=lg, work=work, done=done)

        def run_tracker(c, seg):
            """Round c of the recurrence; under `overlap` on the side stream, behind the span's gather."""
            if seg["work"] is not None:
                seg["work"].wait()
            return self.tracker(to_bctq(seg["embds"]), None, resume=self._resume or c > 0,
                                frame_embeds_no_norm=to_bctq(seg["embds_nn"]), need_masks=False)

        segs, tracks = [run_segmenter(0)], []
        if overlap:
            self._tracker_stream.wait_stream(main)
        for c in range(len(plan)):
            if c + 1 < len(plan):
                segs.append(run_segmenter(c + 1))       # enqueue the next span BEFORE blocking on this span's matching
            if overlap:
                with torch.cuda.stream(self._tracker_stream):
                    self._tracker_stream.wait_event(segs[c]["done"])
                    tracks.append(run_tracker(c, segs[c]))
            else:
                tracks.append(run_tracker(c, segs[c]))
        if overlap:
            main.wait_stream(self._tracker_stream)
        snap = self._guard_snapshot()                   # (span pipeline: raises; DVIS_X3=0 is the remedy named in the message)
        if snap is not None and (shard.world > 1 or shard.force):
            tag = shard.all_reduce_max(snap[3].to(torch.float32))      # same decision on every rank
            Fn.X3_GUARD.verify_gathered(tag, self.device, self)
        else:
            self._guard_verify(snap)
        cat = lambda xs, d: xs[0] if len(xs) == 1 else torch.cat(xs, d)
        embds_nn = cat([s_["embds_nn"] for s_ in segs], 0)
        track = {"pred_embds": cat([t["pred_embds"] for t in tracks], 2),
                 "pred_logits": cat([t["pred_logits"] for t in tracks], 1)}
        ref = self.refiner(track["pred_embds"], to_bctq(embds_nn), None, need_masks=False)
        cls, aux = PP.mean_logits(ref["pred_logits"], track["pred_logits"])
        if contiguous:
            emb_local = ref["mask_embed"][:, local_ids[0]:local_ids[0] + len(local_ids)] if local_ids \
                else ref["mask_embed"][:, :0]                                       # (1, t_local, Q, Cm)
        else:
            emb_local = ref["mask_embed"][:, torch.as_tensor(local_ids, device=self.device)]
        mf = mask_features.unsqueeze(0)

        def mask_fn(idx):
            return self.refiner.predict_masks(emb_local, mf, idx)[0]                # (q', t_local, h, w)
        out_hw = (video.get("height", img_size[0]), video.get("width", img_size[1]))
        out = self._task_output(cls, aux, mask_fn, img_size, out_hw, padded, len(local_ids), video)
        out["frame_ids"] = local_ids                                                # which frames of the clip the masks are
        out["frame_range"] = (local_ids[0], local_ids[-1] + 1) if local_ids and contiguous else None
        return PP.to_reference_format(out) if self.reference_outputs else out


def _slice_maps(maps, a, b):
    """Frames [a, b) of a list of (N, C, h, w) maps, keeping the token views of pixel_decoder.TokenMaps."""
    out = type(maps)(m[a:b] for m in maps)
    if getattr(maps, "tokens", None) is not None:
        out.tokens = [t[a:b] for t in maps.tokens]
    return out


def _make_backbone(backbone):
    """Built LAST by the builders so that the head / tracker / refiner weights of a given seed do not depend on the
    backbone choice."""
    if backbone == "r50":
        from .backbone import build_resnet50
        return build_resnet50()
    from .vit_adapter import D2VitAdapterDinoV2
    return D2VitAdapterDinoV2(backbone)


def build_dvis_plus_r50(mode="offline", *, num_classes=124, num_queries=100, n_things=58, task="vps", hidden_dim=256,
                        nheads=8, dim_feedforward=2048, dec_layers=10, enc_layers=6, tracker_layers=6,
                        refiner_layers=6, max_num=20, object_mask_threshold=0.8, overlap_threshold=0.8, seed=0,
                        segmenter_chunk=0, backbone="r50"):
    """DVIS++ R50 with the sizes of configs/dvis_Plus/VIPSeg/DVIS_Plus_{Online,Offline}_R50.yaml (HIDDEN_DIM 256,
    NHEADS 8, DIM_FEEDFORWARD 2048, DEC_LAYERS 10, 6 encoder / tracker / refiner layers, 100 queries, REID branch ->
    512-channel tracker/refiner) and deterministic random weights following the reference's init rules.
    backbone="vitl" / "vitb": the ViT-Adapter variants (configs/dvis_Plus/VIPSeg/vit_adapter/*.yaml swap the backbone
    and use 200 queries — pass num_queries=200)."""
    from .backbone import build_resnet50
    from .pixel_decoder import MSDeformAttnPixelDecoder, r50_input_shape
    from .refiner import TemporalRefiner
    from .tracker import ReferringTracker_noiser
    from .transformer_decoder import (VideoMultiScaleMaskedTransformerDecoder_dvisPlus,
                                      VideoMultiScaleMaskedTransformerDecoder_minvis)
    torch.manual_seed(seed)
    if backbone == "r50":
        in_shape = r50_input_shape()
    else:
        from .registry import ShapeSpec
        dim = {"vitl": 1024, "vitb": 768}[backbone]
        in_shape = {k: ShapeSpec(channels=dim, stride=s) for k, s in (("res2", 4), ("res3", 8), ("res4", 16), ("res5", 32))}
    pixel_decoder = MSDeformAttnPixelDecoder(
        in_shape, transformer_dropout=0.0, transformer_nheads=nheads, transformer_dim_feedforward=1024,
        transformer_enc_layers=enc_layers, conv_dim=hidden_dim, mask_dim=hidden_dim, norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    predictor = VideoMultiScaleMaskedTransformerDecoder_dvisPlus(
        hidden_dim, True, num_classes=num_classes, hidden_dim=hidden_dim, num_queries=num_queries, nheads=nheads,
        dim_feedforward=dim_feedforward, dec_layers=dec_layers - 1, pre_norm=False, mask_dim=hidden_dim,
        enforce_input_project=False, num_frames=1, num_reid_head_layers=3, reid_hidden_dim=hidden_dim)
    if mode == "minvis":     # configs/dvis_Plus/*/MinVIS_*.yaml: the per-frame decoder without the re-id branch, no tracker
        predictor = VideoMultiScaleMaskedTransformerDecoder_minvis(
            hidden_dim, True, num_classes=num_classes, hidden_dim=hidden_dim, num_queries=num_queries, nheads=nheads,
            dim_feedforward=dim_feedforward, dec_layers=dec_layers - 1, pre_norm=False, mask_dim=hidden_dim,
            enforce_input_project=False, num_frames=1)
        head = MaskFormerHead(num_classes=num_classes, pixel_decoder=pixel_decoder, transformer_predictor=predictor)
        return MinVIS(backbone=_make_backbone(backbone), sem_seg_head=head, num_queries=num_queries, task="vis",
                      segmenter_chunk=segmenter_chunk).eval()
    head = MaskFormerHead(num_classes=num_classes, pixel_decoder=pixel_decoder, transformer_predictor=predictor)
    tracker = ReferringTracker_noiser(hidden_channel=2 * hidden_dim, feedforward_channel=dim_feedforward,
                                      num_head=nheads, decoder_layer_num=tracker_layers, noise_mode="wa",
                                      mask_dim=hidden_dim, class_num=num_classes)
    kw = dict(backbone=_make_backbone(backbone), sem_seg_head=head, num_queries=num_queries,
              object_mask_threshold=object_mask_threshold, overlap_threshold=overlap_threshold, n_things=n_things,
              tracker=tracker, task=task, max_num=max_num, segmenter_chunk=segmenter_chunk)
    if mode == "online":
        return DVIS_Plus_online(**kw).eval()
    refiner = TemporalRefiner(hidden_channel=2 * hidden_dim, feedforward_channel=dim_feedforward, num_head=nheads,
                              decoder_layer_num=refiner_layers, mask_dim=hidden_dim, class_num=num_classes, windows=3)
    return DVIS_Plus_offline(refiner=refiner, **kw).eval()


build_dvis_plus = build_dvis_plus_r50     # backbone-agnostic name (backbone="r50" | "vitl" | "vitb")


@META_ARCH_REGISTRY.register()
class MaskFormer(nn.Module):
    """Image Mask2Former (BASELINE config #1) — inference half of mask2former/maskformer_model.py:167-380.
    ``forward([{"image": (3,H,W), "height", "width"}, ...])`` -> list of dicts with "sem_seg" (K,H,W),
    "panoptic_seg" (map int32, segments_info) and/or "instances" ({"pred_masks","scores","pred_classes"})."""

    @configurable
    def __init__(self, *, backbone, sem_seg_head, num_queries, object_mask_threshold=0.8, overlap_threshold=0.8,
                 thing_ids=(), size_divisibility=32, sem_seg_postprocess_before_inference=True,
                 pixel_mean=(123.675, 116.280, 103.530), pixel_std=(58.395, 57.120, 57.375), semantic_on=True,
                 panoptic_on=False, instance_on=False, test_topk_per_image=100, metadata=None, criterion=None):
        super().__init__()
        self.backbone, self.sem_seg_head, self.num_queries = backbone, sem_seg_head, num_queries
        self.object_mask_threshold, self.overlap_threshold = object_mask_threshold, overlap_threshold
        self.metadata = metadata
        ids = d2.thing_ids_from_metadata(metadata)
        self.thing_ids = set(int(t) for t in thing_ids) if ids is None else set(ids)
        self.size_divisibility = size_divisibility
        self.sem_seg_postprocess_before_inference = sem_seg_postprocess_before_inference
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)
        self.semantic_on, self.panoptic_on, self.instance_on = semantic_on, panoptic_on, instance_on
        self.test_topk_per_image = test_topk_per_image

    @property
    def device(self):
        return self.pixel_mean.device

    @classmethod
    def from_config(cls, cfg):
        """mask2former/maskformer_model.py:100-162, minus the training criterion."""
        mf = cfg.MODEL.MASK_FORMER
        backbone = d2.build_backbone(cfg)
        test = mf.TEST
        return {
            "backbone": backbone, "sem_seg_head": d2.build_sem_seg_head(cfg, backbone.output_shape()),
            "criterion": None, "num_queries": mf.NUM_OBJECT_QUERIES,
            "object_mask_threshold": test.OBJECT_MASK_THRESHOLD, "overlap_threshold": test.OVERLAP_THRESHOLD,
            "metadata": d2.dataset_metadata(cfg), "size_divisibility": mf.SIZE_DIVISIBILITY,
            "sem_seg_postprocess_before_inference": bool(_get(test, "SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE", False)
                                                         or _get(test, "PANOPTIC_ON", False)
                                                         or _get(test, "INSTANCE_ON", False)),
            "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD,
            "semantic_on": _get(test, "SEMANTIC_ON", True), "instance_on": _get(test, "INSTANCE_ON", False),
            "panoptic_on": _get(test, "PANOPTIC_ON", False),
            "test_topk_per_image": _get(_get(cfg, "TEST", {}), "DETECTIONS_PER_IMAGE", 100),
        }

    @staticmethod
    def sem_seg_postprocess(result, img_size, output_height, output_width):
        """detectron2.modeling.postprocessing.sem_seg_postprocess (un-vendored): crop the padding, resize."""
        result = result[:, :img_size[0], :img_size[1]].expand(1, -1, -1, -1)
        return torch.nn.functional.interpolate(result, size=(output_height, output_width), mode="bilinear",
                                               align_corners=False)[0]

    def semantic_inference(self, mask_cls, mask_pred):
        mask_cls = torch.softmax(mask_cls, dim=-1)[..., :-1]
        return torch.einsum("qc,qhw->chw", mask_cls, mask_pred.sigmoid())

    def panoptic_inference(self, mask_cls, mask_pred):
        scores, labels = torch.softmax(mask_cls, dim=-1).max(-1)
        mask_pred = mask_pred.sigmoid()
        keep = labels.ne(self.sem_seg_head.num_classes) & (scores > self.object_mask_threshold)
        cur_scores, cur_classes, cur_masks = scores[keep], labels[keep], mask_pred[keep]
        h, w = cur_masks.shape[-2:]
        panoptic_seg = torch.zeros((h, w), dtype=torch.int32, device=cur_masks.device)
        segments_info = []
        if cur_masks.shape[0] == 0:
            return panoptic_seg, segments_info
        cur_mask_ids = (cur_scores.view(-1, 1, 1) * cur_masks).argmax(0)
        K = cur_classes.shape[0]
        conf = cur_masks.gather(0, cur_mask_ids[None])[0] >= 0.5
        flat = cur_mask_ids.flatten()
        stats = torch.stack([torch.bincount(flat, minlength=K).double(), (cur_masks >= 0.5).flatten(1).sum(1).double(),
                             torch.bincount(flat, weights=conf.flatten().float(), minlength=K).double(),
                             cur_classes.double()]).cpu()                    # one device->host copy
        lut = torch.zeros(K, dtype=torch.int32)
        seg_id, stuff = 0, {}
        for k in range(K):
            area, orig, inter, cls_k = int(stats[0, k]), int(stats[1, k]), int(stats[2, k]), int(stats[3, k])
            isthing = cls_k in self.thing_ids
            if area > 0 and orig > 0 and inter > 0:
                if area / orig < self.overlap_threshold:
                    continue
                if not isthing:
                    if cls_k in stuff:
                        lut[k] = stuff[cls_k]
                        continue
                    stuff[cls_k] = seg_id + 1
                seg_id += 1
                lut[k] = seg_id
                segments_info.append({"id": seg_id, "isthing": bool(isthing), "category_id": cls_k})
        panoptic_seg = torch.where(conf, lut.to(conf.device)[cur_mask_ids], panoptic_seg)
        return panoptic_seg, segments_info

    def instance_inference(self, mask_cls, mask_pred):
        K = self.sem_seg_head.num_classes
        scores = torch.softmax(mask_cls, dim=-1)[:, :-1]
        labels = torch.arange(K, device=mask_cls.device).unsqueeze(0).repeat(self.num_queries, 1).flatten(0, 1)
        scores_per_image, topk = scores.flatten(0, 1).topk(self.test_topk_per_image, sorted=False)
        labels_per_image = labels[topk]
        mask_pred = mask_pred[topk // K]
        if self.panoptic_on:
            keep = torch.tensor([int(l) in self.thing_ids for l in labels_per_image.tolist()], dtype=torch.bool,
                                device=mask_pred.device)
            scores_per_image, labels_per_image, mask_pred = scores_per_image[keep], labels_per_image[keep], mask_pred[keep]
        pred_masks = (mask_pred > 0).float()
        mask_scores = (mask_pred.sigmoid().flatten(1) * pred_masks.flatten(1)).sum(1) / (pred_masks.flatten(1).sum(1) + 1e-6)
        return {"pred_masks": pred_masks, "scores": scores_per_image * mask_scores, "pred_classes": labels_per_image}

    @torch.no_grad()
    @Fn.fp32_island
    def forward(self, batched_inputs):
        assert not self.training, "dvis_plus_amd implements the inference path"
        sizes = [tuple(x["image"].shape[-2:]) for x in batched_inputs]
        d = self.size_divisibility
        Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
        Hp, Wp = ((Hm + d - 1) // d * d, (Wm + d - 1) // d * d) if d > 1 else (Hm, Wm)
        batch = torch.zeros((len(sizes), 3, Hp, Wp), dtype=torch.float32, device=self.device)
        for i, x in enumerate(batched_inputs):
            img = (x["image"].to(self.device, torch.float32) - self.pixel_mean) / self.pixel_std
            batch[i, :, :sizes[i][0], :sizes[i][1]] = img
        if getattr(self, "_x3_off", False):
            with Fn.x3_disabled():
                outputs = self.sem_seg_head(self.backbone(batch))
        else:
            outputs = self.sem_seg_head(self.backbone(batch))
            try:        # range guard of the split-f16 kernels (functions._X3RangeGuard); the image path synchronises here
                Fn.x3_range_verify(Fn.x3_range_snapshot(self.device), self)
            except Fn.X3RangeError as e:
                if Fn.X3_ON_OVERFLOW != "rerun":
                    raise
                import warnings
                warnings.warn(f"{e}  Re-running on the exact-fp32 kernels; this model stays on them.", RuntimeWarning)
                self._x3_off = True
                return self.forward(batched_inputs)
        mask_cls_results = outputs["pred_logits"]
        mask_pred_results = torch.nn.functional.interpolate(outputs["pred_masks"], size=(Hp, Wp), mode="bilinear",
                                                            align_corners=False)
        results = []
        for mask_cls, mask_pred, inp, image_size in zip(mask_cls_results, mask_pred_results, batched_inputs, sizes):
            height, width = inp.get("height", image_size[0]), inp.get("width", image_size[1])
            r = {}
            if self.sem_seg_postprocess_before_inference:
                mask_pred = self.sem_seg_postprocess(mask_pred, image_size, height, width)
            if self.semantic_on:
                sem = self.semantic_inference(mask_cls, mask_pred)
                if not self.sem_seg_postprocess_before_inference:
                    sem = self.sem_seg_postprocess(sem, image_size, height, width)
                r["sem_seg"] = sem
            if self.panoptic_on:
                r["panoptic_seg"] = self.panoptic_inference(mask_cls, mask_pred)
            if self.instance_on:
                r["instances"] = self.instance_inference(mask_cls, mask_pred)
            results.append(r)
        return results


def build_mask2former_r50(*, num_classes=133, num_queries=100, hidden_dim=256, nheads=8, dim_feedforward=2048,
                          dec_layers=10, enc_layers=6, seed=0, backbone="r50", **kw):
    """Image Mask2Former R50 (BASELINE config #1 sizes) with deterministic random weights."""
    from .backbone import build_resnet50
    from .pixel_decoder import MSDeformAttnPixelDecoder, r50_input_shape
    from .transformer_decoder import MultiScaleMaskedTransformerDecoder
    torch.manual_seed(seed)
    if backbone == "r50":
        in_shape = r50_input_shape()
    else:
        from .registry import ShapeSpec
        dim = {"vitl": 1024, "vitb": 768}[backbone]
        in_shape = {k: ShapeSpec(channels=dim, stride=s) for k, s in (("res2", 4), ("res3", 8), ("res4", 16), ("res5", 32))}
    pixel_decoder = MSDeformAttnPixelDecoder(
        in_shape, transformer_dropout=0.0, transformer_nheads=nheads, transformer_dim_feedforward=1024,
        transformer_enc_layers=enc_layers, conv_dim=hidden_dim, mask_dim=hidden_dim, norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4)
    predictor = MultiScaleMaskedTransformerDecoder(
        hidden_dim, True, num_classes=num_classes, hidden_dim=hidden_dim, num_queries=num_queries, nheads=nheads,
        dim_feedforward=dim_feedforward, dec_layers=dec_layers - 1, pre_norm=False, mask_dim=hidden_dim,
        enforce_input_project=False)
    head = MaskFormerHead(num_classes=num_classes, pixel_decoder=pixel_decoder, transformer_predictor=predictor)
    return MaskFormer(backbone=_make_backbone(backbone), sem_seg_head=head, num_queries=num_queries, **kw).eval()
