"""The detectron2 construction protocol, so that the reference's callers build these modules the way they build the
reference's own:

    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)            # detectron2 build_model, train_net_video.py:302
    head  = SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, shape)  # detectron2 build_sem_seg_head
    dec   = TRANSFORMER_DECODER_REGISTRY.get(name)(cfg, in_channels, mask_classification)
                                                     # mask2former/modeling/transformer_decoder/maskformer_transformer_decoder.py:16-27

i.e. every registered class accepts EITHER its explicit keyword arguments OR a config node (+ the positional extras of its
``from_config``).  The reference gets that from detectron2's ``@configurable``; detectron2 is un-vendored, so the same
protocol is implemented here (``configurable``), independent of whether detectron2 is importable — it only needs a
config object with attribute access (detectron2's CfgNode, yacs, or dvis_plus_amd.config.CfgNode).

``install()`` puts the classes into detectron2's / the reference's registries under the reference's names, REPLACING the
reference's own entries when its packages were imported first (train_net_video.py:48-62): one added line in the
launcher selects this build for every yaml of the reference.
"""
import functools
import inspect

from .registry import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY, TRANSFORMER_DECODER_REGISTRY)


def _is_cfg(x):
    """A config node: attribute access to MODEL (detectron2 / yacs CfgNode, omegaconf, dvis_plus_amd.config.CfgNode)."""
    if x is None or isinstance(x, (str, bytes, int, float, list, tuple)):
        return False
    try:
        return hasattr(x, "MODEL") or (isinstance(x, dict) and "MODEL" in x)
    except Exception:
        return False


def _called_with_cfg(*args, **kwargs):
    if args and _is_cfg(args[0]):
        return True
    return _is_cfg(kwargs.get("cfg"))


def _args_from_config(from_config, *args, **kwargs):
    """Call cls.from_config(cfg, ...) with the arguments it declares; keyword arguments it does not declare override the
    values it returns (detectron2 semantics: explicit kwargs win over the config)."""
    params = inspect.signature(from_config).parameters
    if list(params)[:1] != ["cfg"]:
        raise TypeError(f"{from_config.__qualname__} must take 'cfg' as its first argument")
    takes_any = any(p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) for p in params.values())
    if takes_any:
        return from_config(*args, **kwargs)
    extra = {k: kwargs.pop(k) for k in list(kwargs) if k not in params}
    ret = from_config(*args, **kwargs)
    ret.update(extra)
    return ret


def configurable(init_func):
    """Decorator for ``__init__``: ``Cls(cfg, *extras)`` -> ``Cls(**Cls.from_config(cfg, *extras))``; explicit keyword
    construction is untouched."""
    if init_func.__name__ != "__init__":
        raise TypeError("@configurable decorates __init__")

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        if _called_with_cfg(*args, **kwargs):
            try:
                from_config = type(self).from_config
            except AttributeError as e:
                raise AttributeError(f"{type(self).__name__} needs a from_config classmethod to be built from a config") from e
            init_func(self, **_args_from_config(from_config, *args, **kwargs))
        else:
            init_func(self, *args, **kwargs)
    return wrapped


def _populate():
    """Import the modules whose classes register themselves (idempotent)."""
    from . import backbone, meta_architecture, pixel_decoder, transformer_decoder, vit_adapter  # noqa: F401


# ---- the builders the reference's from_config functions call ------------------------------------------------------
def build_backbone(cfg, input_shape=None):
    """detectron2.modeling.build_backbone: ``BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)``.
    Names served here: ``build_resnet_backbone`` (R50, the fused inference ResNet of backbone.py) and
    ``D2VitAdapterDinoV2`` (vit_adapter.py); anything else is detectron2's to build when it is importable."""
    _populate()
    name = cfg.MODEL.BACKBONE.NAME
    if name in BACKBONE_REGISTRY:
        return BACKBONE_REGISTRY.get(name)(cfg, input_shape)
    try:
        from detectron2.modeling import build_backbone as d2_build
    except Exception:
        raise KeyError(f"backbone '{name}' is not provided by dvis_plus_amd and detectron2 is not importable") from None
    return d2_build(cfg) if input_shape is None else d2_build(cfg, input_shape)


def build_sem_seg_head(cfg, input_shape):
    """detectron2.modeling.build_sem_seg_head."""
    _populate()
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)


def build_pixel_decoder(cfg, input_shape):
    """mask2former/modeling/pixel_decoder/fpn.py:21-34."""
    _populate()
    name = cfg.MODEL.SEM_SEG_HEAD.PIXEL_DECODER_NAME
    model = SEM_SEG_HEADS_REGISTRY.get(name)(cfg, input_shape)
    if not callable(getattr(model, "forward_features", None)):
        raise ValueError(f"Only SEM_SEG_HEADS with a forward_features method can be used as pixel decoder ({name})")
    return model


def build_transformer_decoder(cfg, in_channels, mask_classification=True):
    """mask2former/modeling/transformer_decoder/maskformer_transformer_decoder.py:22-27."""
    _populate()
    name = cfg.MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME
    return TRANSFORMER_DECODER_REGISTRY.get(name)(cfg, in_channels, mask_classification)


def build_model(cfg):
    """detectron2.modeling.build_model (minus ``.to(cfg.MODEL.DEVICE)``, left to the caller)."""
    _populate()
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)


def thing_ids_from_metadata(metadata, video=False):
    """The contiguous class ids post-processing treats as things; None when the metadata has no
    ``thing_dataset_id_to_contiguous_id`` table.  The reference uses TWO rules:
      * image MaskFormer (mask2former/maskformer_model.py:314, :363): ``cls in table.values()``;
      * the video meta-architectures (dvis_Plus/meta_architecture.py:919): ``cls < len(table)`` — whatever the table's
        values are.  The reference registers VIPSeg with id -> id (dvis_Plus/data_video/datasets/vps.py:276-285), so the
        58 thing ids are interleaved over 0..123 and the two rules disagree there; ``video=True`` is the second one."""
    table = getattr(metadata, "thing_dataset_id_to_contiguous_id", None)
    if table is None:
        return None
    if video:
        return frozenset(range(len(table)))
    return frozenset(int(v) for v in (table.values() if hasattr(table, "values") else table))


def dataset_metadata(cfg):
    """``MetadataCatalog.get(cfg.DATASETS.TRAIN[0])`` when detectron2 is importable, else None."""
    try:
        from detectron2.data import MetadataCatalog
        names = cfg.DATASETS.TRAIN
        return MetadataCatalog.get(names[0]) if len(names) else None
    except Exception:
        return None


def _force_register(registry, obj):
    """Register `obj` under its class name, replacing an existing entry (fvcore's Registry keeps them in _obj_map)."""
    name = obj.__name__
    table = getattr(registry, "_obj_map", None)
    if isinstance(table, dict):
        table[name] = obj
        return
    table = getattr(registry, "_map", None)          # minimal registries (tests, dvis_plus_amd.registry)
    if isinstance(table, dict):
        table[name] = obj
        return
    registry.register(obj)


def install():
    """Register every dvis_plus_amd class in detectron2's META_ARCH / SEM_SEG_HEADS / BACKBONE registries and in the
    reference's TRANSFORMER_DECODER_REGISTRY (when those packages are importable), replacing same-named entries.
    Returns the list of (registry name, class name) pairs that were installed."""
    import importlib
    _populate()
    done = []
    targets = []
    try:
        dm = importlib.import_module("detectron2.modeling")
        targets += [(getattr(dm, "META_ARCH_REGISTRY", None), META_ARCH_REGISTRY),
                    (getattr(dm, "SEM_SEG_HEADS_REGISTRY", None), SEM_SEG_HEADS_REGISTRY),
                    (getattr(dm, "BACKBONE_REGISTRY", None), BACKBONE_REGISTRY)]
    except Exception:
        pass
    try:
        mt = importlib.import_module("mask2former.modeling.transformer_decoder.maskformer_transformer_decoder")
        targets.append((getattr(mt, "TRANSFORMER_DECODER_REGISTRY", None), TRANSFORMER_DECODER_REGISTRY))
    except Exception:
        pass
    for theirs, ours in targets:
        if theirs is None:
            continue
        for name, obj in ours.items():
            _force_register(theirs, obj)
            done.append((getattr(theirs, "_name", "?"), name))
    return done
