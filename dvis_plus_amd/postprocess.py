"""Video post-processing on the device — SURVEY.md §8 rows a12 (post_processing) and a13
(inference_video_vis / vps / vss), following dvis_Plus/meta_architecture.py:758-772, 818-979.

The reference runs these on CPU tensors (offline mode) with one ``.item()`` sync per candidate segment.  Here
everything stays on the GPU; the VPS segment bookkeeping needs three small per-segment statistics, fetched with a
single device->host copy, and the panoptic map is then written with one look-up-table gather.  Integer outputs
(argmax ids, segment ids, top-k labels, boolean masks) follow the reference's order of operations exactly.
"""
import torch
import torch.nn.functional as F


def get_instance_labels(pred_logits):
    labels = torch.argmax(F.softmax(pred_logits[0], dim=-1), dim=2)
    labels[labels == pred_logits.shape[-1] - 1] = -1
    return labels


def mean_logits(pred_logits, aux_logits=None):
    """post_processing(): class logits averaged over T.  (1,T,Q,K+1) -> (Q,K+1)."""
    out = pred_logits[0].mean(dim=0)
    return out, (None if aux_logits is None else aux_logits[0].mean(dim=0))


def _resize2(masks, first_resize_size, img_size, out_hw, sigmoid):
    if masks.shape[1] == 0:                      # a rank that holds no frame of this clip (T < world * frames_per_rank)
        return masks.new_zeros((masks.shape[0], 0, *out_hw))
    m = F.interpolate(masks, size=tuple(first_resize_size), mode="bilinear", align_corners=False)
    m = m[:, :, :img_size[0], :img_size[1]]
    if sigmoid:
        m = m.sigmoid()
    return F.interpolate(m, size=tuple(out_hw), mode="bilinear", align_corners=False)


def resize2_gt0(masks, first_resize_size, img_size, out_hw):
    """Boolean instance masks: both resizes + ``> 0`` (meta_architecture.py:843-853).  On the GPU one fused kernel that
    evaluates the resizes in torch's CPU operation order (the reference post-processes on the host); the torch ops
    otherwise (CPU tensors in the host-logic tests)."""
    if masks.is_cuda and masks.dtype == torch.float32 and masks.shape[1] > 0:
        from . import functions as Fn
        if masks.stride(3) != 1 or masks.stride(2) != masks.shape[3]:
            masks = masks.contiguous()
        return Fn.resize2_gt0(masks, first_resize_size, img_size, out_hw)
    return _resize2(masks, first_resize_size, img_size, out_hw, sigmoid=False) > 0.


def vis_select(pred_cls, num_classes, max_num, aux_pred_cls=None):
    """Top-k (query, class) pairs.  Returns (scores, labels, query index)."""
    scores = F.softmax(pred_cls, dim=-1)[:, :-1]
    if aux_pred_cls is not None:
        scores = torch.maximum(scores, F.softmax(aux_pred_cls, dim=-1)[:, :-1].to(scores))
    Q = pred_cls.shape[0]
    labels = torch.arange(num_classes, device=pred_cls.device).unsqueeze(0).repeat(Q, 1).flatten(0, 1)
    scores_per_image, topk = scores.flatten(0, 1).topk(max_num, sorted=False)
    return scores_per_image, labels[topk], topk // num_classes


def inference_video_vis(pred_cls, mask_fn, img_size, out_hw, first_resize_size, num_classes, max_num,
                        aux_pred_cls=None):
    """mask_fn(query_index) -> (q', T, h, w) mask logits of the selected queries only."""
    scores, labels, qidx = vis_select(pred_cls, num_classes, max_num, aux_pred_cls)
    masks = resize2_gt0(mask_fn(qidx), first_resize_size, img_size, out_hw)
    return {"image_size": tuple(out_hw), "pred_scores": scores, "pred_labels": labels, "pred_masks": masks,
            "pred_ids": qidx, "task": "vis"}


def vps_select(pred_cls, num_classes, object_mask_threshold, aux_pred_cls=None):
    probs = F.softmax(pred_cls, dim=-1)
    if aux_pred_cls is not None:
        probs[:, :-1] = torch.maximum(probs[:, :-1], F.softmax(aux_pred_cls, dim=-1)[:, :-1].to(probs))
    scores, labels = probs.max(-1)
    keep = labels.ne(num_classes) & (scores > object_mask_threshold)
    return scores, labels, keep


def inference_video_vps(pred_cls, mask_fn, img_size, out_hw, first_resize_size, num_classes, n_things,
                        object_mask_threshold, overlap_threshold, aux_pred_cls=None, num_frames=None, reduce_fn=None):
    """n_things: the thing classes — an int n (classes 0..n-1, VIPSeg's layout) or a collection of contiguous class ids
    (``metadata.thing_dataset_id_to_contiguous_id.values()``, meta_architecture.py:940).
    reduce_fn: sums a tensor over the ranks that hold the other frames of the clip (segment areas are per clip)."""
    thing_ids = frozenset(range(n_things)) if isinstance(n_things, int) else frozenset(int(t) for t in n_things)
    scores, labels, keep = vps_select(pred_cls, num_classes, object_mask_threshold, aux_pred_cls)
    ids = torch.nonzero(keep).flatten()                       # sync #1: how many queries survive
    dev = pred_cls.device
    if ids.numel() == 0:
        T = num_frames if num_frames is not None else 0
        return {"image_size": tuple(out_hw), "pred_masks": torch.zeros((T, *out_hw), dtype=torch.int32, device=dev),
                "segments_infos": [], "pred_ids": [], "task": "vps", "num_candidates": 0}
    cur_scores, cur_classes = scores[ids], labels[ids]
    K = ids.numel()
    logits = mask_fn(ids)                                                                     # (K', T, h, w)
    if logits.shape[1] == 0:
        # no frame of the clip on this rank: empty panoptic map, zero areas — but the same collective as everyone else
        cur_mask_ids = torch.zeros((0, *out_hw), dtype=torch.long, device=dev)
        conf = torch.zeros((0, *out_hw), dtype=torch.bool, device=dev)
        areas = torch.zeros((3, K), dtype=torch.float64, device=dev)
    elif logits.is_cuda and K <= 256:
        # one fused pass over the stride-4 logits (two-stage resize + sigmoid + weighted arg-max + areas)
        from . import functions as Fn
        cur_mask_ids, conf, areas = Fn.vps_argmax(logits, cur_scores, first_resize_size, img_size, out_hw)
        cur_mask_ids = cur_mask_ids.long()
        areas = areas.double()
    else:
        cur_masks = _resize2(logits, first_resize_size, img_size, out_hw, sigmoid=True)       # (K', T, H, W)
        cur_mask_ids = (cur_scores.view(-1, 1, 1, 1) * cur_masks).argmax(0)                   # (T, H, W)
        conf = cur_masks.gather(0, cur_mask_ids.unsqueeze(0))[0] >= 0.5                       # winner's own prob >= .5
        flat_ids = cur_mask_ids.flatten()
        mask_area = torch.bincount(flat_ids, minlength=K)
        inter = torch.bincount(flat_ids, weights=conf.flatten().to(torch.float32), minlength=K)
        original_area = (cur_masks >= 0.5).flatten(1).sum(1)
        areas = torch.stack([mask_area.double(), original_area.double(), inter.double()])
    if reduce_fn is not None:
        areas = reduce_fn(areas)
    stats = torch.cat([areas, cur_classes.double()[None], ids.double()[None]]).cpu()         # sync #2: one copy
    lut = torch.zeros(K, dtype=torch.int32)
    segments, out_ids, seg_id, stuff = [], [], 0, {}
    for k in range(K):
        area, orig, it = int(stats[0, k]), int(stats[1, k]), int(stats[2, k])
        cls_k = int(stats[3, k])
        isthing = cls_k in thing_ids
        if area > 0 and orig > 0 and it > 0:
            if area / orig < overlap_threshold:
                continue
            if not isthing:
                if cls_k in stuff:
                    lut[k] = stuff[cls_k]
                    continue
                stuff[cls_k] = seg_id + 1
            seg_id += 1
            lut[k] = seg_id
            segments.append({"id": seg_id, "isthing": bool(isthing), "category_id": cls_k})
            out_ids.append(int(stats[4, k]))
    panoptic = torch.where(conf, lut.to(dev)[cur_mask_ids], torch.zeros((), dtype=torch.int32, device=dev))
    return {"image_size": tuple(out_hw), "pred_masks": panoptic, "segments_infos": segments, "pred_ids": out_ids,
            "task": "vps", "num_candidates": K}


def inference_video_vss(pred_cls, mask_fn, img_size, out_hw, first_resize_size, aux_pred_cls=None, frame_chunk=8):
    mask_cls = F.softmax(pred_cls, dim=-1)[..., :-1]
    if aux_pred_cls is not None:
        mask_cls = torch.maximum(mask_cls, F.softmax(aux_pred_cls, dim=-1)[..., :-1].to(mask_cls))
    masks = mask_fn(None)                                                                     # (Q, T, h, w)
    if masks.shape[1] == 0:
        return {"image_size": tuple(out_hw), "pred_masks": torch.zeros((0, *out_hw), dtype=torch.long, device=masks.device),
                "task": "vss"}
    if masks.is_cuda and mask_cls.shape[1] <= 128:
        from . import functions as Fn
        if masks.stride(3) != 1 or masks.stride(2) != masks.shape[3]:
            masks = masks.contiguous()
        sem = Fn.vss_argmax(masks, mask_cls, first_resize_size, img_size, out_hw)             # one pass, no (C,T,H,W)
        return {"image_size": tuple(out_hw), "pred_masks": sem, "task": "vss"}
    outs = []
    own = masks.is_cuda and masks.dtype == torch.float32 and not torch.is_grad_enabled()
    if own:
        # > 128 classes (the one-pass kernel's limit): the class contraction still must not be a library GEMM — phase B of
        # stream() runs next to the segmenter's stream-K kernels (csrc/gemm.hip) — so it goes through dvis_gemm_nt as
        # scores (pixels, classes) = cur^T (pixels, Q) @ cls^T (classes, Q)^T, Q zero-padded to a multiple of 4
        from . import functions as Fn
        Q, C = mask_cls.shape
        Qp = (Q + 3) // 4 * 4
        w = torch.zeros((C, Qp), dtype=torch.float32, device=masks.device)
        w[:, :Q] = mask_cls.t()
    for s in range(0, masks.shape[1], frame_chunk):                                           # bound the 720p blow-up
        cur = _resize2(masks[:, s:s + frame_chunk], first_resize_size, img_size, out_hw, sigmoid=True)
        if own:
            a = torch.zeros((cur[0].numel(), Qp), dtype=torch.float32, device=cur.device)
            a[:, :Q] = cur.flatten(1).t()
            outs.append(Fn.gemm_nt(a, w).max(1)[1].view(cur.shape[1:]))
        else:
            outs.append(torch.einsum("qc,qthw->cthw", mask_cls, cur).max(0)[1])
    return {"image_size": tuple(out_hw), "pred_masks": torch.cat(outs, 0), "task": "vss"}


def to_reference_format(out):
    """Task dict in exactly the reference's output format (meta_architecture.py:603-626, :848-867, :944-950, :975-979) — what
    its evaluators consume unchanged (data_video/ytvis_eval.py:268-295: python floats / ints and per-instance (T, H, W)
    CPU masks that go through numpy + RLE; vps_eval.py:106-135 / vss_eval.py:92-93: `.numpy()` on the maps):
      vis: pred_scores list[float], pred_labels list[int], pred_ids list[int], pred_masks list of (T, H, W) bool CPU tensors;
      vps: pred_masks (T, H, W) int32 CPU tensor, segments_infos, pred_ids list[int];   vss: pred_masks (T, H, W) CPU.
    The product keeps everything on the device by default (the next consumer is usually another kernel; the 110 MB panoptic
    map of a 30-frame 720p clip costs ~5 ms of PCIe); models built the detectron2 way (`Cls(cfg)`) convert."""
    out = dict(out)
    task = out.get("task")
    to_list = lambda v: v.tolist() if torch.is_tensor(v) else [int(x) if not isinstance(x, float) else x for x in v]
    if task == "vis" or ("pred_scores" in out and task is None):
        masks = out["pred_masks"]
        if torch.is_tensor(masks):
            masks = [m for m in masks.cpu()]
        else:
            masks = [m.cpu() for m in masks]
        out.update(pred_scores=to_list(out["pred_scores"]), pred_labels=to_list(out["pred_labels"]),
                   pred_ids=to_list(out["pred_ids"]), pred_masks=masks)
    elif task == "vps":
        out.update(pred_masks=out["pred_masks"].cpu(), pred_ids=[int(i) for i in out["pred_ids"]])
    elif task == "vss":
        out.update(pred_masks=out["pred_masks"].cpu())
    out.pop("ready_event", None)
    return out
