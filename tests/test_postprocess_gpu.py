"""GPU parity: the fused panoptic arg-max kernel vs the reference's tensor-by-tensor sequence (torch ops, fp32)
and the golden fixture.  Integer maps must agree except where two candidates' weighted probabilities (or the 0.5
confidence threshold) are within fp32 rounding of each other."""
import pytest
import torch
import torch.nn.functional as F

from conftest import Golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_sequence(logits, scores, first, img, out_hw):
    m = F.interpolate(logits, size=first, mode="bilinear", align_corners=False)[:, :, :img[0], :img[1]].sigmoid()
    m = F.interpolate(m, size=out_hw, mode="bilinear", align_corners=False)
    weighted = scores.view(-1, 1, 1, 1) * m
    ids = weighted.argmax(0)
    conf = m.gather(0, ids[None])[0] >= 0.5
    K = logits.shape[0]
    areas = torch.stack([torch.bincount(ids.flatten(), minlength=K), (m >= 0.5).flatten(1).sum(1),
                         torch.bincount(ids.flatten(), weights=conf.flatten().float(), minlength=K).long()])
    top2 = weighted.topk(min(2, K), dim=0)[0]
    margin = top2[0] - top2[-1] if K > 1 else torch.ones_like(top2[0])
    return ids, conf, areas, margin, m


@pytest.mark.parametrize("K,T,h,w,first,img,out", [
    (20, 3, 46, 80, (184, 320), (180, 320), (180, 320)),      # identity second stage (bench case, scaled down)
    (7, 2, 10, 14, (40, 56), (37, 53), (30, 45)),             # both stages non-trivial, crop + down-size
    (1, 1, 5, 6, (20, 24), (20, 24), (41, 50)),               # single candidate, up-size
    (33, 2, 12, 20, (48, 80), (45, 77), (90, 160)),
])
def test_vps_argmax_vs_torch_sequence(K, T, h, w, first, img, out):
    from dvis_plus_amd.functions import vps_argmax
    g = torch.Generator().manual_seed(K * 10 + T)
    logits = (torch.randn(T, K, h, w, generator=g) * 3).to(DEV).permute(1, 0, 2, 3)   # (K,T,h,w) strided like mask_fn
    scores = torch.rand(K, generator=g).to(DEV) * 0.5 + 0.5
    ids, conf, areas = vps_argmax(logits, scores, first, img, out)
    r_ids, r_conf, r_areas, margin, m = _torch_sequence(logits.contiguous(), scores, first, img, out)
    clear = margin > 1e-5
    assert torch.equal(ids.long()[clear], r_ids[clear])
    assert (~clear).float().mean() < 1e-3
    best_prob = m.gather(0, r_ids[None])[0]
    sure = clear & ((best_prob - 0.5).abs() > 1e-5)
    assert torch.equal(conf[sure], r_conf[sure])
    tol = max(3, int(2e-4 * T * out[0] * out[1]))
    assert (areas.long() - r_areas).abs().max().item() <= tol
    assert int(areas[0].sum()) == T * out[0] * out[1]


def test_vps_fused_path_reproduces_golden_fixture():
    from dvis_plus_amd import postprocess as P
    g = Golden("g6_postprocess")
    i, o, cfg = g.ins, g.outs, g.meta["cfg"]
    logits, aux = P.mean_logits(i["pred_logits"].to(DEV), i["aux_logits"].to(DEV))
    masks = i["pred_masks"][0].to(DEV)
    fn = lambda idx: masks if idx is None else masks[idx]
    p = P.inference_video_vps(logits, fn, cfg["img_size"], cfg["out_hw"], cfg["first_resize"], cfg["K"],
                              cfg["n_things"], cfg["object_mask_threshold"], cfg["overlap_threshold"], aux,
                              num_frames=cfg["T"])
    assert [s["id"] for s in p["segments_infos"]] == o["vps_seg_id"].tolist()
    assert [s["category_id"] for s in p["segments_infos"]] == o["vps_seg_cat"].tolist()
    assert p["pred_ids"] == o["vps_ids"].tolist()
    assert (p["pred_masks"].cpu() == o["vps_masks"]).float().mean().item() > 0.999


@pytest.mark.parametrize("Q,C,T,h,w,first,img,out", [
    (100, 124, 2, 46, 80, (184, 320), (180, 320), (180, 320)),    # VIPSeg class count, identity second stage
    (13, 7, 2, 10, 14, (40, 56), (37, 53), (30, 45)),             # both stages non-trivial
    (40, 33, 1, 12, 20, (48, 80), (45, 77), (90, 160)),
    (5, 128, 1, 6, 8, (24, 32), (24, 32), (24, 32)),              # largest supported class count
])
def test_vss_argmax_vs_torch_sequence(Q, C, T, h, w, first, img, out):
    """dvis_vss_argmax == interpolate -> crop -> sigmoid -> interpolate -> einsum("qc,qthw->cthw") -> max(0)."""
    from dvis_plus_amd.functions import vss_argmax
    g = torch.Generator().manual_seed(Q + C)
    logits = (torch.randn(T, Q, h, w, generator=g) * 3).to(DEV).permute(1, 0, 2, 3)     # (Q,T,h,w) strided like mask_fn
    cls = torch.softmax(torch.randn(Q, C + 1, generator=g) * 2, -1)[:, :-1].to(DEV)
    got = vss_argmax(logits, cls, first, img, out)
    m = F.interpolate(logits.contiguous(), size=first, mode="bilinear", align_corners=False)[:, :, :img[0], :img[1]]
    m = F.interpolate(m.sigmoid(), size=out, mode="bilinear", align_corners=False)
    sem = torch.einsum("qc,qthw->cthw", cls.double(), m.double())
    want = sem.max(0)[1]
    top2 = sem.topk(min(2, C), dim=0)[0]
    clear = (top2[0] - top2[-1]) > 1e-5 if C > 1 else torch.ones_like(want, dtype=torch.bool)
    assert got.dtype == torch.int64 and got.shape == want.shape
    assert torch.equal(got[clear], want[clear])
    assert (~clear).float().mean() < 1e-3


def test_vss_fused_path_reproduces_golden_fixture():
    from dvis_plus_amd import postprocess as P
    g = Golden("g6_postprocess")
    i, o, cfg = g.ins, g.outs, g.meta["cfg"]
    logits, aux = P.mean_logits(i["pred_logits"].to(DEV), i["aux_logits"].to(DEV))
    masks = i["pred_masks"][0].to(DEV)
    s = P.inference_video_vss(logits, lambda idx: masks, cfg["img_size"], cfg["out_hw"], cfg["first_resize"], aux)
    assert (s["pred_masks"].cpu() == o["vss_masks"]).float().mean().item() > 0.999
