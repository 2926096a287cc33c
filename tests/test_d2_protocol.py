"""CPU: the detectron2 construction protocol (dvis_plus_amd/d2.py).

The reference's launcher builds its model with detectron2's ``build_model``:
``META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)`` (train_net_video.py:302), the meta-architecture's
``from_config`` calls ``build_backbone`` / ``build_sem_seg_head`` (dvis_Plus/meta_architecture.py:1162-1164), the head's
calls ``build_pixel_decoder`` / ``build_transformer_decoder`` (mask_former_head.py:88-116 ->
maskformer_transformer_decoder.py:16-27).  Here the same third-party stubs tests/golden/_ref_import.py uses stand in for
detectron2's registries, ``d2.install()`` writes this build's classes into them, and the model is built from the
REFERENCE'S OWN yaml (configs/dvis_Plus/VIPSeg/DVIS_Plus_Offline_R50.yaml with its _BASE_ chain — read directly when
/root/reference exists, from the committed resolved copy otherwise) exactly the way detectron2 would.
"""
import json
import os
import sys
import types

import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
REF_CFG = "/root/reference/DVIS_Plus/configs/dvis_Plus/VIPSeg"

VIPSEG_THING_IDS = [2, 4, 8, 10, 41, 43, 44, 46, 47, 48, 49, 50, 51, 52, 54, 55, 56, 60, 61, 62, 63, 64, 65, 72, 74, 76, 77, 78,
                    79, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 95, 96, 97, 99, 100, 101, 102, 106, 107, 108, 109, 114,
                    115, 116, 117, 118, 122, 123]


def _cfg(name, rel):
    from dvis_plus_amd.config import CfgNode, get_default_cfg
    cfg = get_default_cfg()
    fixture = CfgNode(json.load(open(os.path.join(GOLDEN, name + ".json"))))
    path = os.path.join(REF_CFG, rel)
    if os.path.exists(path):                                    # build container: the yaml itself
        cfg.merge_from_file(path)
        direct = get_default_cfg().merge(fixture)
        assert json.dumps(cfg, sort_keys=True, default=str) == json.dumps(direct, sort_keys=True, default=str), \
            "committed cfg fixture is stale w.r.t. the reference's yaml"
    else:
        cfg.merge(fixture)
    return cfg


@pytest.fixture
def d2_stubs(monkeypatch):
    """Stub registries under detectron2's / the reference's module names, pre-loaded with same-named "reference" classes."""
    import _ref_import as R                                      # only its stub classes; R.install() is never called
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        monkeypatch.setitem(sys.modules, name, m)
        mods[name] = m
        return m

    class _Meta:
        # VIPSeg as the reference registers it (dvis_Plus/data_video/datasets/vps.py:276-285): the 58 thing classes map
        # id -> id, interleaved over 0..123 (ids of the `isthing` entries of its category table)
        thing_dataset_id_to_contiguous_id = {i: i for i in VIPSEG_THING_IDS}

    class _Catalog:
        @staticmethod
        def get(name):
            return _Meta()
    dm = mod("detectron2.modeling", META_ARCH_REGISTRY=R._Registry("META_ARCH"),
             SEM_SEG_HEADS_REGISTRY=R._Registry("SEM_SEG_HEADS"), BACKBONE_REGISTRY=R._Registry("BACKBONE"))
    mod("detectron2", modeling=dm)
    mod("detectron2.data", MetadataCatalog=_Catalog)
    td = mod("mask2former.modeling.transformer_decoder.maskformer_transformer_decoder",
             TRANSFORMER_DECODER_REGISTRY=R._Registry("TRANSFORMER_MODULE"))
    for name in ("mask2former", "mask2former.modeling", "mask2former.modeling.transformer_decoder"):
        mod(name)
    # the reference's own classes registered first, as after `from dvis_Plus import ...` (train_net_video.py:48-62)
    for reg, names in ((dm.META_ARCH_REGISTRY, ["DVIS_Plus_offline", "DVIS_Plus_online", "MinVIS"]),
                       (dm.SEM_SEG_HEADS_REGISTRY, ["MaskFormerHead", "MSDeformAttnPixelDecoder"]),
                       (td.TRANSFORMER_DECODER_REGISTRY, ["VideoMultiScaleMaskedTransformerDecoder_dvisPlus"])):
        for n in names:
            reg.register(type(n, (), {"reference": True}))
    return dm, td


def test_configurable_protocol():
    from dvis_plus_amd.config import CfgNode
    from dvis_plus_amd.d2 import configurable

    class A:
        @configurable
        def __init__(self, x, *, y=1, z=2):
            self.x, self.y, self.z = x, y, z

        @classmethod
        def from_config(cls, cfg, x):
            return {"x": x, "y": cfg.MODEL.Y}
    cfg = CfgNode({"MODEL": {"Y": 7}})
    a = A(cfg, 3)
    assert (a.x, a.y, a.z) == (3, 7, 2)
    assert (A(cfg, x=4, z=9).x, A(cfg, x=4, z=9).z) == (4, 9)           # kwargs from_config does not declare override
    b = A(5, y=6)
    assert (b.x, b.y, b.z) == (5, 6, 2)                                  # explicit construction untouched


def test_build_model_like_detectron2_from_the_references_offline_yaml(d2_stubs):
    from dvis_plus_amd import d2
    dm, td = d2_stubs
    done = d2.install()
    assert ("META_ARCH", "DVIS_Plus_offline") in done and ("TRANSFORMER_MODULE",
                                                           "VideoMultiScaleMaskedTransformerDecoder_dvisPlus") in done
    cfg = _cfg("cfg_DVIS_Plus_Offline_R50", "DVIS_Plus_Offline_R50.yaml")
    arch = dm.META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    assert arch.__module__ == "dvis_plus_amd.meta_architecture"          # the reference's entry was replaced
    torch.manual_seed(0)
    model = arch(cfg)                                                    # detectron2: build_model(cfg)
    assert type(model).__name__ == "DVIS_Plus_offline" and model.task == "vps" and model.window_size == 3
    assert model.num_queries == 100 and model.sem_seg_head.num_classes == 124
    # video rule of the reference (dvis_Plus/meta_architecture.py:919): isthing = cls < len(thing table) — NOT membership
    # in the table's values (that is the image model's rule, maskformer_model.py:314); the oracle's `cls_k < n_things`
    assert model.thing_ids == frozenset(range(58)) and model.n_things == 58
    assert d2.thing_ids_from_metadata(model.metadata) == frozenset(VIPSEG_THING_IDS)       # image MaskFormer rule
    assert model.object_mask_threshold == 0.8 and model.overlap_threshold == 0.8
    assert model.reference_outputs is True          # built from cfg: outputs in the evaluators' format (lists, CPU tensors)
    pred = model.sem_seg_head.predictor
    assert type(pred).__name__ == "VideoMultiScaleMaskedTransformerDecoder_dvisPlus" and pred.num_layers == 9
    assert len(model.sem_seg_head.pixel_decoder.transformer.encoder.layers) == 6
    assert len(model.tracker.transformer_ffn_layers) == 6 and len(model.refiner.transformer_ffn_layers) == 6
    assert model.tracker.decoder_norm.weight.shape[0] == 512             # REID_BRANCH doubles the width
    # the head and the decoder can be built the reference's way on their own
    head = dm.SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, model.backbone.output_shape())
    assert type(head.pixel_decoder).__name__ == "MSDeformAttnPixelDecoder"
    dec = td.TRANSFORMER_DECODER_REGISTRY.get(cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME)(cfg, 256, True)
    assert dec.num_queries == 100

    # checkpoint layout: every key / shape of the reference's modules at this configuration, strictly
    shapes = json.load(open(os.path.join(GOLDEN, "ref_state_shapes_dvis_plus_r50.json")))
    ours = {k: list(v.shape) for k, v in model.state_dict().items()}
    for k, shp in shapes.items():
        assert ours.get(k) == shp, (k, ours.get(k), shp)
    extra = [k for k in ours if k not in shapes and not k.startswith("backbone.")]
    assert not extra, extra
    sd = {k: torch.full(shp, 0.5) for k, shp in shapes.items()}
    sd.update({k: v for k, v in model.state_dict().items() if k.startswith("backbone.")})
    model.load_state_dict(sd, strict=True)
    assert float(model.refiner.class_embed.weight.detach()[0, 0]) == 0.5


@pytest.mark.parametrize("name,rel,arch,backbone", [
    ("cfg_DVIS_Plus_Online_R50", "DVIS_Plus_Online_R50.yaml", "DVIS_Plus_online", "ResNet"),
    ("cfg_MinVIS_R50", "MinVIS_R50.yaml", "MinVIS", "ResNet"),
])
def test_other_reference_yamls_build(d2_stubs, name, rel, arch, backbone):
    from dvis_plus_amd import d2
    d2.install()
    cfg = _cfg(name, rel)
    model = d2_stubs[0].META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    assert type(model).__name__ == arch and type(model.backbone).__name__ == backbone
    assert (model.tracker is None) == (arch == "MinVIS") and model.refiner is None


def test_vit_adapter_yaml_selects_the_vit_backbone():
    """vit_adapter/*.yaml: BACKBONE.NAME D2VitAdapterDinoV2, 200 queries (built at ViT-B size to keep the test light)."""
    from dvis_plus_amd import d2
    cfg = _cfg("cfg_DVIS_Plus_Offline_VitAdapterL", "vit_adapter/DVIS_Plus_Offline_VitAdapterL.yaml")
    assert cfg.MODEL.BACKBONE.NAME == "D2VitAdapterDinoV2" and cfg.MODEL.MASK_FORMER.NUM_OBJECT_QUERIES == 200
    cfg.MODEL.VIT_ADAPTER.NAME = "vitb"
    bb = d2.build_backbone(cfg)
    assert type(bb).__name__ == "D2VitAdapterDinoV2" and bb.output_shape()["res5"].channels == 768
