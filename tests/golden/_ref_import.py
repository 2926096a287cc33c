"""Import harness for the reference's Python modules (BUILD CONTAINER ONLY).

Used only by ``gen_golden.py`` to *generate* golden input/output vectors.  It
never travels usefully to the GPU box: ``/root/reference`` does not exist there
and nothing in ``tests/`` imports this file at test time.

What it does (recipe: SURVEY.md Appendix C):
  * pre-seeds ``sys.modules`` with an empty ``MultiScaleDeformableAttention`` so
    ``ops/functions/ms_deform_attn_func.py:22`` imports; the reference's
    ``MSDeformAttn.forward`` then lands in its own torch path
    (``ops/modules/ms_deform_attn.py:119-121``) = BASELINE config #1.
  * stubs ONLY un-vendored third-party names (detectron2 / fvcore), none of the
    reference's own code.
  * pre-creates empty parent packages with ``__path__`` into the reference so
    the dataset-registering ``__init__.py`` files never execute.
"""
import importlib
import sys
import types

import torch
from torch import nn
import torch.nn.functional as F

REF = "/root/reference/DVIS_Plus"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Registry:
    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._map[name]


class _Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d semantics: conv -> norm -> activation."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def _get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    assert norm == "GN"
    return nn.GroupNorm(32, out_channels)


class _ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


def _c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def _c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def _configurable(init_func=None, *, from_config=None):
    if init_func is not None:
        return init_func

    def deco(f):
        return f
    return deco


_done = False


def install():
    global _done
    if _done:
        return
    _done = True
    _mod("MultiScaleDeformableAttention")
    wi = _mod("fvcore.nn.weight_init", c2_xavier_fill=_c2_xavier_fill, c2_msra_fill=_c2_msra_fill)
    fn = _mod("fvcore.nn", weight_init=wi)
    _mod("fvcore", nn=fn)
    _mod("detectron2")
    _mod("detectron2.config", configurable=_configurable)
    _mod("detectron2.layers", Conv2d=_Conv2d, ShapeSpec=_ShapeSpec, get_norm=_get_norm)
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=_Registry)
    _mod("detectron2.modeling", SEM_SEG_HEADS_REGISTRY=_Registry("SEM_SEG_HEADS"),
         META_ARCH_REGISTRY=_Registry("META_ARCH"))

    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    pkg("mask2former", f"{REF}/mask2former")
    pkg("mask2former.modeling", f"{REF}/mask2former/modeling")
    pkg("mask2former.modeling.transformer_decoder", f"{REF}/mask2former/modeling/transformer_decoder")
    pkg("mask2former.modeling.pixel_decoder", f"{REF}/mask2former/modeling/pixel_decoder")
    ops = pkg("mask2former.modeling.pixel_decoder.ops", f"{REF}/mask2former/modeling/pixel_decoder/ops")
    fpk = pkg("mask2former.modeling.pixel_decoder.ops.functions",
              f"{REF}/mask2former/modeling/pixel_decoder/ops/functions")
    mpk = pkg("mask2former.modeling.pixel_decoder.ops.modules",
              f"{REF}/mask2former/modeling/pixel_decoder/ops/modules")
    pkg("mask2former_video", f"{REF}/mask2former_video")
    pkg("mask2former_video.modeling", f"{REF}/mask2former_video/modeling")
    pkg("mask2former_video.modeling.transformer_decoder",
        f"{REF}/mask2former_video/modeling/transformer_decoder")
    pkg("dvis_Plus", f"{REF}/dvis_Plus")

    func = importlib.import_module("mask2former.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    fpk.MSDeformAttnFunction = func.MSDeformAttnFunction
    fpk.ms_deform_attn_func = func
    modm = importlib.import_module("mask2former.modeling.pixel_decoder.ops.modules.ms_deform_attn")
    mpk.MSDeformAttn = modm.MSDeformAttn
    ops.functions, ops.modules = fpk, mpk


def ref(name):
    install()
    return importlib.import_module(name)


def ref_vit_adapter():
    """Import the reference's vendored ViT-Adapter (mask2former/modeling/backbones_vitAdapter/adapter.py).
    Extra stubs: timm.models.layers.{trunc_normal_, DropPath} and detectron2.modeling.{BACKBONE_REGISTRY, Backbone}
    (un-vendored third-party names; DropPath is the identity in eval mode, the only mode used here)."""
    install()

    class _DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            assert not self.training or self.drop_prob == 0.0
            return x

    tl = _mod("timm.models.layers", trunc_normal_=nn.init.trunc_normal_, DropPath=_DropPath)
    tm = _mod("timm.models", layers=tl)
    _mod("timm", models=tm)
    d2m = sys.modules["detectron2.modeling"]
    d2m.BACKBONE_REGISTRY = _Registry("BACKBONE")
    d2m.Backbone = nn.Module
    d2m.ShapeSpec = _ShapeSpec
    if "mask2former.modeling.backbones_vitAdapter" not in sys.modules:
        m = types.ModuleType("mask2former.modeling.backbones_vitAdapter")
        m.__path__ = [f"{REF}/mask2former/modeling/backbones_vitAdapter"]
        sys.modules["mask2former.modeling.backbones_vitAdapter"] = m
    return importlib.import_module("mask2former.modeling.backbones_vitAdapter.adapter")


def ref_meta():
    """Import dvis_Plus.meta_architecture for its *inference* methods only.

    Extra stubs: un-vendored detectron2 names, plus placeholders for the
    reference's TRAINING-ONLY criterion/matcher modules (out of scope, never
    called) so the module-level imports at meta_architecture.py:13-15 resolve.
    """
    install()
    if "dvis_Plus.meta_architecture" in sys.modules:
        return sys.modules["dvis_Plus.meta_architecture"]

    class _Any:
        def __init__(self, *a, **k):
            pass

    _mod("detectron2.data", MetadataCatalog=_Any)
    dm = sys.modules["detectron2.modeling"]
    dm.build_backbone = None
    dm.build_sem_seg_head = None
    _mod("detectron2.modeling.backbone", Backbone=_Any)
    _mod("detectron2.structures", Boxes=_Any, ImageList=_Any, Instances=_Any, BitMasks=_Any)
    _mod("mask2former_video.modeling.criterion", VideoSetCriterion=_Any)
    _mod("mask2former_video.modeling.matcher", VideoHungarianMatcher=_Any,
         VideoHungarianMatcher_Consistent=_Any)
    m = types.ModuleType("mask2former_video.utils")
    m.__path__ = [f"{REF}/mask2former_video/utils"]
    sys.modules["mask2former_video.utils"] = m
    return importlib.import_module("dvis_Plus.meta_architecture")
