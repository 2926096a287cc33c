"""Name -> class registries with the reference's registry names.

When detectron2 is importable, classes are ALSO registered into its ``SEM_SEG_HEADS_REGISTRY`` /
``META_ARCH_REGISTRY`` and into the ``TRANSFORMER_DECODER_REGISTRY`` name the reference defines at
mask2former/modeling/transformer_decoder/maskformer_transformer_decoder.py:16, so yaml keys
``MODEL.META_ARCHITECTURE`` / ``SEM_SEG_HEAD.PIXEL_DECODER_NAME`` / ``MASK_FORMER.TRANSFORMER_DECODER_NAME``
select these implementations from ``train_net_video.py`` unchanged.  Without detectron2
(this image) the local registries serve the standalone runner.
"""
from dataclasses import dataclass
from typing import Optional


@dataclass
class ShapeSpec:
    channels: Optional[int] = None
    height: Optional[int] = None
    width: Optional[int] = None
    stride: Optional[int] = None


class Registry:
    def __init__(self, name, mirror=None):
        self._name, self._map, self._mirror = name, {}, mirror

    def register(self, obj=None):
        def add(o):
            if o.__name__ in self._map:
                raise KeyError(f"{o.__name__} already registered in {self._name}")
            self._map[o.__name__] = o
            if self._mirror is not None:
                try:
                    self._mirror.register(o)
                except Exception:   # name already taken by the reference's own class: ours stays local
                    pass
            return o
        return add if obj is None else add(obj)

    def get(self, name):
        if name not in self._map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._map[name]

    def __contains__(self, name):
        return name in self._map


def _d2(name):
    try:
        import detectron2.modeling as dm
        return getattr(dm, name)
    except Exception:
        return None


SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS", _d2("SEM_SEG_HEADS_REGISTRY"))
META_ARCH_REGISTRY = Registry("META_ARCH", _d2("META_ARCH_REGISTRY"))
TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")
