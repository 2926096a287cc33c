"""Minimal yacs-style config for running without detectron2: yaml files with ``_BASE_`` inheritance, attribute access,
``KEY VALUE`` overrides, and the defaults of the keys that size the hot path (what the reference adds through
``add_maskformer2_config`` — mask2former/config.py:6-123 —, ``add_maskformer2_video_config`` —
mask2former_video/config.py:6 — and ``add_minvis_config / add_dvis_config`` — dvis_Plus/config.py:12-78).
Only keys read by ``dvis_plus_amd`` are defaulted; unknown keys in a yaml are kept as they are, so the reference's own
config files (e.g. configs/dvis_Plus/VIPSeg/DVIS_Plus_Offline_R50.yaml and its _BASE_ chain) load unchanged.
With detectron2 installed, its CfgNode works as well: every ``from_config`` here only uses attribute access.
"""
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else copy.deepcopy(v)
        return self

    def merge_from_file(self, path):
        return self.merge(load_yaml_with_base(path))

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0, "overrides come as KEY VALUE pairs"
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode())
            node[parts[-1]] = yaml.safe_load(val) if isinstance(val, str) else val
        return self


def load_yaml_with_base(path):
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    base = cfg.pop("_BASE_", None)
    if base is None:
        return cfg
    if not os.path.isabs(base):
        base = os.path.join(os.path.dirname(path), base)
    merged = CfgNode(load_yaml_with_base(base))
    return merged.merge(cfg)


def get_default_cfg():
    """Defaults of the keys the hot path reads (values as in the reference's add_*_config functions)."""
    return CfgNode({
        "MODEL": {
            "META_ARCHITECTURE": "DVIS_Plus_offline",
            "BACKBONE": {"NAME": "build_resnet_backbone"},
            "VIT_ADAPTER": {"NAME": "vitl", "VIT_WEIGHT": None, "FREEZE_VIT": True, "FINETUNE": False,
                            "FINETUNE_INDEXES": [0], "WITH_CP": False},
            "PIXEL_MEAN": [123.675, 116.280, 103.530],
            "PIXEL_STD": [58.395, 57.120, 57.375],
            "SEM_SEG_HEAD": {
                "NAME": "MaskFormerHead", "IN_FEATURES": ["res2", "res3", "res4", "res5"], "NUM_CLASSES": 124,
                "CONVS_DIM": 256, "MASK_DIM": 256, "NORM": "GN", "COMMON_STRIDE": 4,
                "PIXEL_DECODER_NAME": "MSDeformAttnPixelDecoder", "TRANSFORMER_ENC_LAYERS": 6,
                "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res3", "res4", "res5"],
            },
            "MASK_FORMER": {
                "TRANSFORMER_DECODER_NAME": "VideoMultiScaleMaskedTransformerDecoder_dvisPlus",
                "TRANSFORMER_IN_FEATURE": "multi_scale_pixel_decoder", "HIDDEN_DIM": 256, "NHEADS": 8,
                "DIM_FEEDFORWARD": 2048, "DEC_LAYERS": 10, "PRE_NORM": False, "ENFORCE_INPUT_PROJ": False,
                "NUM_OBJECT_QUERIES": 100, "DROPOUT": 0.0, "SIZE_DIVISIBILITY": 32, "REID_BRANCH": True,
                "REID_HIDDEN_DIM": 256, "NUM_REID_HEAD_LAYERS": 3,
                "TEST": {"OBJECT_MASK_THRESHOLD": 0.8, "OVERLAP_THRESHOLD": 0.8, "WINDOW_INFERENCE": True,
                         "WINDOW_SIZE": 3, "TASK": "vis", "MAX_NUM": 20},
            },
            "TRACKER": {"DECODER_LAYERS": 6, "NOISE_MODE": "none", "NOISE_RATIO": 0.5},
            "REFINER": {"DECODER_LAYERS": 6},
        },
        "INPUT": {"SAMPLING_FRAME_NUM": 1},
    })


def build_model(cfg, n_things=None):
    """cfg -> model, resolved through the registries by the yaml names (MODEL.META_ARCHITECTURE / BACKBONE.NAME /
    SEM_SEG_HEAD.NAME / SEM_SEG_HEAD.PIXEL_DECODER_NAME / MASK_FORMER.TRANSFORMER_DECODER_NAME) exactly like
    detectron2's ``build_model`` does for the reference: ``META_ARCH_REGISTRY.get(name)(cfg)``.
    n_things: thing classes 0..n-1 when no dataset metadata is available (standalone runs without detectron2)."""
    from . import d2
    d2_model = d2.build_model(cfg)
    if n_things is not None and hasattr(d2_model, "thing_ids") \
            and d2.thing_ids_from_metadata(d2_model.metadata) is None:     # no metadata, or one without a thing table (VSS)
        d2_model.thing_ids = frozenset(range(int(n_things)))
    return d2_model.eval()
