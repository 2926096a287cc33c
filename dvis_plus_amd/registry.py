"""Name -> class registries with the reference's registry names.

``META_ARCH`` / ``SEM_SEG_HEADS`` / ``BACKBONE`` are detectron2's registries (un-vendored), ``TRANSFORMER_MODULE`` is the
one the reference defines at mask2former/modeling/transformer_decoder/maskformer_transformer_decoder.py:16.  These local
registries always resolve the yaml keys ``MODEL.META_ARCHITECTURE`` / ``SEM_SEG_HEAD.NAME`` / ``SEM_SEG_HEAD.PIXEL_DECODER_NAME``
/ ``MASK_FORMER.TRANSFORMER_DECODER_NAME`` / ``BACKBONE.NAME`` to this build's classes (``dvis_plus_amd.d2.build_model``);
``dvis_plus_amd.d2.install()`` additionally writes them into detectron2's and the reference's own registry objects when
those are importable, so ``train_net_video.py`` selects them.
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
    def __init__(self, name):
        self._name, self._map = name, {}

    def register(self, obj=None):
        def add(o):
            if o.__name__ in self._map:
                raise KeyError(f"{o.__name__} already registered in {self._name}")
            self._map[o.__name__] = o
            return o
        return add if obj is None else add(obj)

    def get(self, name):
        if name not in self._map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._map[name]

    def items(self):
        return list(self._map.items())

    def __contains__(self, name):
        return name in self._map


SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")
