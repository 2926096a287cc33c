"""GPU parity of the fused post-processing kernels (csrc/postprocess.hip).

The reference post-processes on CPU tensors, so its integer outputs are defined by the arithmetic of torch's CPU
kernels.  The product evaluates that arithmetic operation for operation (csrc/torch_cpu_math.h), which these tests pin in
three layers:
  1. floats: dvis_resize2 == torch's CPU interpolate -> crop -> (sigmoid) -> interpolate, bit for bit (with the sigmoid:
     up to ~1e-4 of the floats by 1 ulp, where torch's own result depends on its thread count);
  2. decisions on the same logits: masks / arg-max ids / confidences / areas == the torch CPU sequence, torch.equal;
  3. the reference's own outputs: the golden fixtures g6_postprocess (small sizes) and g6_postprocess_large, torch.equal.
The semantic class sums are a (C x Q) GEMM whose summation order is the BLAS library's on the CPU and the MFMA's here:
there a pixel may differ only where the fp64 top-2 margin is below fp32 summation noise; the count is reported and bounded.
"""
import pytest
import torch
import torch.nn.functional as F

import intcmp
from conftest import Golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# K, T, (h, w), first, img, out
CASES = [
    (20, 3, (46, 80), (184, 320), (180, 320), (180, 320)),   # generic kernel, second stage copies, crop in y only
    (7, 2, (10, 14), (40, 56), (37, 53), (30, 45)),          # both outputs h + w <= 128; crop in x: scalar-exp tail columns
    (3, 18, (10, 14), (40, 56), (37, 53), (30, 45)),         # 18 frames: 16 vector-lane planes + 2 tail planes
    (1, 1, (5, 6), (20, 24), (20, 24), (41, 50)),            # single candidate, no crop, up-size
    (33, 2, (12, 20), (48, 80), (45, 77), (90, 160)),        # first stage small kernel (48 + 80 = 128), second generic
    (4, 2, (46, 80), (184, 320), (180, 300), (360, 600)),    # generic both, crop in x with a 12-column tail
    (5, 4, (23, 40), (92, 160), (90, 160), (45, 80)),        # down-size second stage, small kernel
]


def _logits(K, T, hw, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(T, K, *hw, generator=g) * 3).permute(1, 0, 2, 3)      # (K,T,h,w) strided like mask_fn's output


def _cpu_sequence(logits, first, img, out, sigmoid):
    m = F.interpolate(logits.contiguous(), size=first, mode="bilinear", align_corners=False)[:, :, :img[0], :img[1]]
    if sigmoid:
        m = m.sigmoid()
    return F.interpolate(m, size=out, mode="bilinear", align_corners=False)


@pytest.mark.parametrize("sigmoid", [False, True])
@pytest.mark.parametrize("K,T,hw,first,img,out", CASES)
def test_resize2_floats_bitwise_vs_torch_cpu(K, T, hw, first, img, out, sigmoid):
    from dvis_plus_amd.functions import resize2
    logits = _logits(K, T, hw, K * 100 + T)
    want = _cpu_sequence(logits, first, img, out, sigmoid)
    got = resize2(logits.to(DEV), first, img, out, sigmoid=sigmoid).cpu()
    diff = got.view(torch.int32) != want.view(torch.int32)
    n = int(diff.sum())
    intcmp._report(f"resize2 floats K={K} T={T} {hw}->{first}->{img}->{out} sigmoid={sigmoid}: {n} of {got.numel()} "
                   f"floats differ bitwise; max |d| {float((got - want).abs().max()):.2e}")
    if sigmoid:
        # torch's CPU sigmoid evaluates the last (n mod 32) elements of every contiguous run it hands to a thread with the
        # SCALAR exp (glibc) instead of the vector one (Sleef); where those runs end depends on the size of the tensor and
        # on the number of host threads (128 on the GPU box, 8 in the build container), so torch's own result is not unique
        # there.  The kernel follows the vector evaluation (and the scalar one for the tail columns of an x-cropped row,
        # which is independent of the thread count): a handful of floats may differ, each by one unit in the last place.
        assert n <= max(4, got.numel() // 10000) and float((got - want).abs().max()) <= 1.2e-7
    else:
        assert n == 0


@pytest.mark.parametrize("K,T,hw,first,img,out", CASES)
def test_resize2_gt0_equals_torch_cpu(K, T, hw, first, img, out):
    from dvis_plus_amd.functions import resize2_gt0
    logits = _logits(K, T, hw, K * 100 + T + 1)
    want = _cpu_sequence(logits, first, img, out, False) > 0
    got = resize2_gt0(logits.to(DEV), first, img, out)
    assert got.dtype == torch.bool
    intcmp.exact(got, want, f"resize2_gt0 K={K} T={T} {hw}->{first}->{img}->{out}")


@pytest.mark.parametrize("K,T,hw,first,img,out", CASES)
def test_vps_argmax_equals_torch_cpu_sequence(K, T, hw, first, img, out):
    from dvis_plus_amd.functions import vps_argmax
    logits = _logits(K, T, hw, K * 10 + T)
    scores = torch.rand(K, generator=torch.Generator().manual_seed(K)) * 0.5 + 0.5
    ids, conf, areas = vps_argmax(logits.to(DEV), scores.to(DEV), first, img, out)
    m = _cpu_sequence(logits, first, img, out, True)
    r_ids = (scores.view(-1, 1, 1, 1) * m).argmax(0)
    r_conf = m.gather(0, r_ids[None])[0] >= 0.5
    r_areas = torch.stack([torch.bincount(r_ids.flatten(), minlength=K), (m >= 0.5).flatten(1).sum(1),
                           torch.bincount(r_ids.flatten(), weights=r_conf.flatten().double(), minlength=K).long()])
    tag = f"K={K} T={T} {hw}->{first}->{img}->{out}"
    intcmp.exact(ids.long(), r_ids, "vps_argmax ids " + tag)
    intcmp.exact(conf, r_conf, "vps_argmax conf " + tag)
    intcmp.exact(areas.long(), r_areas, "vps_argmax areas " + tag)


def _golden_pp(name):
    from dvis_plus_amd import postprocess as P
    g = Golden(name)
    i, o, cfg = g.ins, g.outs, g.meta["cfg"]
    logits, aux = P.mean_logits(i["pred_logits"].to(DEV), i["aux_logits"].to(DEV))
    masks = i["pred_masks"][0].to(DEV)
    return P, o, cfg, logits, aux, masks


@pytest.mark.parametrize("name", ["g6_postprocess", "g6_postprocess_large"])
def test_vps_fused_path_reproduces_golden_fixture(name):
    P, o, cfg, logits, aux, masks = _golden_pp(name)
    p = P.inference_video_vps(logits, lambda idx: masks if idx is None else masks[idx], cfg["img_size"], cfg["out_hw"],
                              cfg["first_resize"], cfg["K"], cfg["n_things"], cfg["object_mask_threshold"],
                              cfg["overlap_threshold"], aux, num_frames=cfg["T"])
    assert [s["id"] for s in p["segments_infos"]] == o["vps_seg_id"].tolist()
    assert [s["category_id"] for s in p["segments_infos"]] == o["vps_seg_cat"].tolist()
    assert [s["isthing"] for s in p["segments_infos"]] == o["vps_seg_isthing"].tolist()
    assert p["pred_ids"] == o["vps_ids"].tolist()
    intcmp.exact(p["pred_masks"], o["vps_masks"], f"{name}: panoptic map vs the reference's")


@pytest.mark.parametrize("name", ["g6_postprocess", "g6_postprocess_large"])
def test_vis_fused_path_reproduces_golden_fixture(name):
    P, o, cfg, logits, aux, masks = _golden_pp(name)
    v = P.inference_video_vis(logits, lambda idx: masks[idx], cfg["img_size"], cfg["out_hw"], cfg["first_resize"],
                              cfg["K"], cfg["max_num"], aux)
    # topk(sorted=False): same set, device-dependent order -> align on (query, class)
    key_ref = o["vis_ids"] * 1000 + o["vis_labels"]
    key_out = v["pred_ids"].cpu() * 1000 + v["pred_labels"].cpu()
    a, b = key_ref.argsort(), key_out.argsort()
    assert torch.equal(key_ref[a], key_out[b])
    torch.testing.assert_close(v["pred_scores"].cpu()[b], o["vis_scores"][a], rtol=1e-5, atol=1e-6)
    intcmp.exact(v["pred_masks"].cpu()[b], o["vis_masks"][a], f"{name}: instance masks vs the reference's")


@pytest.mark.parametrize("name", ["g6_postprocess", "g6_postprocess_large"])
def test_vss_fused_path_reproduces_golden_fixture(name):
    P, o, cfg, logits, aux, masks = _golden_pp(name)
    s = P.inference_video_vss(logits, lambda idx: masks, cfg["img_size"], cfg["out_hw"], cfg["first_resize"], aux)
    # fp64 class sums from the CPU sequence: how close to a tie is a pixel that differs?
    m = _cpu_sequence(masks.cpu(), cfg["first_resize"], cfg["img_size"], cfg["out_hw"], True)
    cls = torch.maximum(F.softmax(logits.cpu(), -1)[..., :-1], F.softmax(aux.cpu(), -1)[..., :-1])
    margin = intcmp.argmax_margin(torch.einsum("qc,qthw->cthw", cls.double(), m.double()))
    intcmp.near_boundary(s["pred_masks"], o["vss_masks"], margin, 1e-5, f"{name}: semantic map vs the reference's",
                         max_count=2)


@pytest.mark.parametrize("Q,C,T,hw,first,img,out", [
    (100, 124, 2, (46, 80), (184, 320), (180, 320), (180, 320)),    # VIPSeg class count, MFMA form (second stage copies)
    (13, 7, 2, (10, 14), (40, 56), (37, 53), (30, 45)),             # both stages non-trivial
    (40, 33, 1, (12, 20), (48, 80), (45, 77), (90, 160)),
    (5, 128, 1, (6, 8), (24, 32), (24, 32), (24, 32)),              # largest supported class count
])
def test_vss_argmax_vs_torch_sequence(Q, C, T, hw, first, img, out):
    """dvis_vss_argmax == interpolate -> crop -> sigmoid -> interpolate -> einsum("qc,qthw->cthw") -> max(0); the
    probabilities are torch's bit for bit (layer 1), the class sums differ by summation order only."""
    from dvis_plus_amd.functions import vss_argmax
    g = torch.Generator().manual_seed(Q + C)
    logits = _logits(Q, T, hw, Q + C)
    cls = torch.softmax(torch.randn(Q, C + 1, generator=g) * 2, -1)[:, :-1]
    got = vss_argmax(logits.to(DEV), cls.to(DEV), first, img, out)
    m = _cpu_sequence(logits, first, img, out, True)
    sem = torch.einsum("qc,qthw->cthw", cls.double(), m.double())
    want = sem.max(0)[1]
    assert got.dtype == torch.int64 and got.shape == want.shape
    n = intcmp.near_boundary(got, want, intcmp.argmax_margin(sem), 2e-6 * float(sem.max()),
                             f"vss_argmax Q={Q} C={C} T={T} {hw}->{first}->{img}->{out} vs fp64 class sums", max_count=8)
    # and against torch's own fp32 einsum the count must be of the same order (both are one rounding pattern each)
    want32 = torch.einsum("qc,qthw->cthw", cls, m).max(0)[1]
    assert int((got.cpu() != want32).sum()) <= 8 + n
