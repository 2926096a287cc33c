"""ViT-Adapter backbone: the product module loads the reference's checkpoint keys strict=True and
reproduces the reference's outputs (tests/golden/g8_vit_adapter.npz).  CPU: host logic with the oracle's stand-ins for
the HIP ops; GPU: the real kernels (fp32-MFMA attention at head dim 32, fused single-level MSDeformAttn)."""
from functools import partial

import pytest
import torch

from conftest import Golden


def _build(cfg):
    from dvis_plus_amd.vit_adapter import DinoV2ViTAdapter, DinoVisionTransformer
    vit = DinoVisionTransformer(img_size=cfg["img_size"], patch_size=cfg["patch"], embed_dim=cfg["embed"],
                                depth=cfg["depth"], num_heads=cfg["heads"], mlp_ratio=4, init_values=0.5, ffn_layer="mlp",
                                block_chunks=0, qkv_bias=True, proj_bias=True, ffn_bias=True)
    return DinoV2ViTAdapter(vit_module=vit, pretrain_size=cfg["img_size"], conv_inplane=cfg["conv_inplane"],
                            n_points=cfg["n_points"], deform_num_heads=cfg["deform_heads"], init_values=1e-6,
                            interaction_indexes=cfg["interaction_indexes"], with_cffn=True,
                            cffn_ratio=cfg["cffn_ratio"], deform_ratio=0.5, add_vit_feature=True,
                            use_extra_extractor=True).eval()


def test_vit_adapter_host_logic_matches_reference(oracle_ops):
    g = Golden("g8_vit_adapter")
    m = _build(g.meta["cfg"])
    m.load_state_dict(g.sd, strict=True)                        # checkpoint surface: identical keys
    with torch.no_grad():
        f = m(g.ins["x"])
        tok, H, W = m.vit_module.prepare_tokens_with_masks(g.ins["x"], return_HW=True)
        b0 = m.vit_module.blocks[0](tok)
    assert [H, W] == g.meta["cfg"]["HW"]
    torch.testing.assert_close(tok, g.outs["tokens"], rtol=1e-5, atol=5e-6)
    torch.testing.assert_close(b0, g.outs["block0"], rtol=1e-4, atol=2e-5)
    for got, k in zip(f, ("f1", "f2", "f3", "f4")):
        torch.testing.assert_close(got, g.outs[k], rtol=1e-4, atol=5e-5)


def test_d2_wrapper_surface():
    from dvis_plus_amd.vit_adapter import D2VitAdapterDinoV2, get_adapter_args
    a = get_adapter_args("vitl")
    assert a["deform_num_heads"] == 16 and a["interaction_indexes"][-1] == [18, 23]
    assert a["vit_module"].embed_dim == 1024 and len(a["vit_module"].blocks) == 24
    del a
    m = D2VitAdapterDinoV2("vitb")
    assert set(m.output_shape()) == {"res2", "res3", "res4", "res5"} and m.size_divisibility == 32
    assert m.output_shape()["res5"].channels == 768 and m.output_shape()["res2"].stride == 4


@pytest.mark.gpu
def test_vit_adapter_gpu_matches_reference():
    g = Golden("g8_vit_adapter")
    m = _build(g.meta["cfg"])
    m.load_state_dict(g.sd, strict=True)
    m = m.to("cuda:0")
    x = g.ins["x"].to("cuda:0")
    with torch.no_grad():
        f = m(x)
        tok, H, W = m.vit_module.prepare_tokens_with_masks(x, return_HW=True)
        b0 = m.vit_module.blocks[0](tok)
    torch.testing.assert_close(b0.cpu(), g.outs["block0"], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(b0.cpu(), g.outs["block0"], rtol=2e-4, atol=5e-5)      # what is actually achieved
    for got, k in zip(f, ("f1", "f2", "f3", "f4")):
        torch.testing.assert_close(got.cpu(), g.outs[k], rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(got.cpu(), g.outs[k], rtol=5e-4, atol=1e-4)
