"""BASELINE config #1 read literally: "Mask2Former R50 single 480p frame, 100 queries, PyTorch CPU MSDeformAttn fallback (plumbing,
no GPU)".  The product serves CPU TENSORS through its own torch formulations (dvis_plus_amd/cpu_ops.py, chosen by device — never
for a GPU tensor); checked here against the reference's golden vectors and, for the whole model, against the oracle."""
import glob
import os

import pytest
import torch

from conftest import GOLDEN, Golden

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g1_msda_*.npz")))


@pytest.mark.parametrize("name", CASES)
def test_ms_deform_attn_core_pytorch_matches_the_references_outputs(name):
    """g1: outputs of the imported reference's ms_deform_attn_core_pytorch (ms_deform_attn_func.py:52-72) incl. its test.py recipe."""
    from dvis_plus_amd.functions import ms_deform_attn_core_pytorch
    g = Golden(name)
    i, o = g.ins, g.outs
    out = ms_deform_attn_core_pytorch(i["value"], i["shapes"], i["loc"], i["w"])
    assert out.dtype == o["out"].dtype and out.shape == o["out"].shape
    tol = dict(rtol=1e-12, atol=1e-13) if out.dtype == torch.float64 else dict(rtol=1e-5, atol=1e-6)   # (other association of the L * P sum)
    torch.testing.assert_close(out, o["out"], **tol)


def test_pixel_decoder_on_cpu_tensors_matches_golden_g2():
    """g2: the reference's pixel decoder + a lone MSDeformAttn call on its CPU path vs the product modules on CPU tensors — the
    product's own torch formulations, no test stand-ins (tests/test_host_modules.py runs the same golden with the oracle's)."""
    from dvis_plus_amd.pixel_decoder import MSDeformAttnPixelDecoder
    from dvis_plus_amd.registry import ShapeSpec
    g = Golden("g2_pixel_decoder")
    chans = g.meta["cfg"]["chans"]
    strides = dict(res2=4, res3=8, res4=16, res5=32)
    pd = MSDeformAttnPixelDecoder({k: ShapeSpec(channels=chans[k], stride=strides[k]) for k in chans},
                                  transformer_dropout=0.0, transformer_nheads=2, transformer_dim_feedforward=64,
                                  transformer_enc_layers=2, conv_dim=32, mask_dim=16, norm="GN",
                                  transformer_in_features=["res3", "res4", "res5"], common_stride=4).eval()
    pd.load_state_dict(g.sd, strict=True)
    feats = {k[5:]: v for k, v in g.ins.items() if k.startswith("feat_")}
    with torch.no_grad():
        mf, out0, ms = pd.forward_features(feats)
        shapes = torch.tensor([(2, 3), (4, 6), (8, 12)])
        lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
        a = pd.transformer.encoder.layers[0].self_attn(g.ins["attn_query"], g.ins["attn_ref"], g.ins["attn_src"], shapes, lsi, None)
    tol = dict(rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(mf, g.outs["mask_features"], **tol)
    torch.testing.assert_close(out0, g.outs["out0"], **tol)
    for x, k in zip(ms, ("ms0", "ms1", "ms2")):
        torch.testing.assert_close(x, g.outs[k], **tol)
    torch.testing.assert_close(a, g.outs["attn_out"], **tol)


def test_decoder_on_cpu_tensors_matches_golden_g3_image():
    """g3_decoder_image: the reference's image decoder (mask2former_transformer_decoder.py:363-431) on its CPU path."""
    from dvis_plus_amd.transformer_decoder import MultiScaleMaskedTransformerDecoder
    g = Golden("g3_decoder_image")
    dec = MultiScaleMaskedTransformerDecoder(32, True, num_classes=7, hidden_dim=32, num_queries=6, nheads=2, dim_feedforward=64,
                                             dec_layers=3, pre_norm=False, mask_dim=16, enforce_input_project=False).eval()
    dec.load_state_dict(g.sd, strict=True)
    with torch.no_grad():
        out = dec([g.ins["x0"], g.ins["x1"], g.ins["x2"]], g.ins["mask_features"])
    torch.testing.assert_close(out["pred_logits"], g.outs["pred_logits"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(out["pred_masks"], g.outs["pred_masks"], rtol=1e-4, atol=2e-5)


def test_config1_one_480x640_frame_on_a_gpu_less_box_vs_oracle():
    """build_mask2former_r50() at its own sizes (133 classes, 100 queries, 6 encoder / 9 + 1 decoder layers), ONE 480 x 640 frame,
    CPU tensors end to end — no GPU, no libdvis_hip.so call — against the oracle (the reference's CPU / torch path) from the
    backbone outputs on."""
    from dvis_plus_amd.meta_architecture import build_mask2former_r50
    from oracle import dvis_torch as O
    from pipeline_parity import perturb_msda
    m = build_mask2former_r50(semantic_on=True)
    perturb_msda(m.sem_seg_head.pixel_decoder)
    img = torch.randint(0, 256, (3, 480, 640), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    sd["pixel_mean"], sd["pixel_std"] = m.pixel_mean.clone(), m.pixel_std.clone()
    with torch.no_grad():
        out = m([{"image": img, "height": 480, "width": 640}])[0]
        sem, _, _ = O.maskformer_image_forward(sd, lambda x: m.backbone(x), img, nheads=8, enc_layers=6, dec_layers=9, num_classes=133)
    assert out["sem_seg"].shape == (133, 480, 640) and out["sem_seg"].device.type == "cpu"
    torch.testing.assert_close(out["sem_seg"], sem, rtol=1e-3, atol=1e-3)
