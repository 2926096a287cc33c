"""Test infrastructure: a 4-level average-pool + 3x3-convolution pyramid standing in for the R50 backbone in the a12 composition
goldens / tests (the R50 itself is un-vendored detectron2 code, "parity unpinned"; what g10 pins is everything AROUND the
backbone: window loop, state hand-off, post-processing).  Same module on the reference side (golden generation), the
oracle side (any callable images -> {res2..res5}) and the product side (needs output_shape() / size_divisibility)."""
import torch
from torch import nn

CHANS = dict(res2=8, res3=12, res4=16, res5=20)
STRIDES = dict(res2=4, res3=8, res4=16, res5=32)


class ToyBackbone(nn.Module):
    size_divisibility = 32

    def __init__(self, chans=None):
        super().__init__()
        self.chans = dict(chans or CHANS)
        self.convs = nn.ModuleDict({k: nn.Conv2d(3, c, kernel_size=3, padding=1) for k, c in self.chans.items()})

    def forward(self, x):
        pool = torch.nn.functional.avg_pool2d
        return {k: torch.tanh(conv(pool(x, STRIDES[k]))) for k, conv in self.convs.items()}

    def output_shape(self):
        from dvis_plus_amd.registry import ShapeSpec
        return {k: ShapeSpec(channels=c, stride=STRIDES[k]) for k, c in self.chans.items()}
