"""ResNet-50 front-end (detectron2 ``build_resnet_backbone`` semantics: STRIDE_IN_1X1 False, FrozenBN, outputs
res2..res5 at strides 4/8/16/32 with 256/512/1024/2048 channels — configs/…/Base-*.yaml:2-16).

detectron2 is an un-vendored third-party dependency of the reference, so this module's numerics are NOT pinned by
any reference source ("parity unpinned"); parity of the hot path is asserted from the
backbone OUTPUTS onward.  Parameter names follow detectron2's checkpoints (stem.conv1.{weight,norm.*},
resN.M.{conv1,conv2,conv3,shortcut}.{weight,norm.*}).  Convolutions run on MIOpen through torch — plain library
calls; frozen batch-norm is folded into the convolution weights once at first use (inference).
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import functions as Fn
from .registry import BACKBONE_REGISTRY, ShapeSpec


class FrozenBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


class ConvBN(nn.Conv2d):
    """bias-free conv followed by FrozenBN (`.norm`), evaluated as ONE conv with folded weights."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=False)
        self.norm = FrozenBatchNorm2d(cout)
        self._folded = None
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")

    def folded(self):
        key = (self.weight._version, self.norm.weight._version, self.norm.running_var._version, self.weight.device,
               self.weight.is_contiguous(memory_format=torch.channels_last))
        if self._folded is None or self._folded[0] != key:
            scale, shift = self.norm.scale_shift()
            w = (self.weight.detach() * scale.reshape(-1, 1, 1, 1))
            if key[-1]:
                w = w.contiguous(memory_format=torch.channels_last)
            self._folded = (key, w, shift.detach().contiguous())
        return self._folded[1], self._folded[2]

    def forward(self, x, res=None, relu=False):
        """conv (MIOpen, no bias) then ONE fused pass: + folded-BN shift (+ residual) (+ ReLU)."""
        w, b = self.folded()
        plain = self.dilation == (1, 1) and self.groups == 1      # the own kernels (and their library fallbacks) are dense, un-dilated
        if plain and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0):
            return Fn.conv1x1_bias_act(x, w, b, res, relu)
        if plain and self.kernel_size == (1, 1) and self.stride == (2, 2) and self.padding == (0, 0) and not key_is_channels_last(w) \
                and _res_ok(res, x, w.shape[0], 2):
            if Fn.conv1x1_x3_ok(x, w, 2, res):       # (any map size: small test maps included — no library convolution in phase A)
                return Fn.conv1x1_x3(x, w, b, res, relu, stride=2)
            if Fn.conv1x1s2_supported(x, w):
                return Fn.conv1x1_mfma(x, w, b, res, relu, stride=2)     # the down-sampling shortcut, read in place
        if plain and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1) and res is None and x.is_cuda \
                and not key_is_channels_last(w):
            return Fn.conv3x3_bias_act(x, w, b, relu)            # own Winograd kernel where the shape is served
        if plain and self.kernel_size == (3, 3) and self.stride == (2, 2) and self.padding == (1, 1) and res is None and x.is_cuda \
                and not key_is_channels_last(w):
            return Fn.conv3x3s2_bias_act(x, w, b, relu)          # own direct kernel where it beats the library
        return Fn.bias_act_(F.conv2d(x, w, None, self.stride, self.padding, self.dilation, self.groups), b, res, relu)


def _res_ok(res, x, Co, stride):
    """None, or a contiguous float32 residual of exactly the strided convolution's output shape."""
    if res is None:
        return True
    N, _, H, W = x.shape
    return res.is_contiguous() and res.dtype == torch.float32 and \
        tuple(res.shape) == (N, Co, (H + stride - 1) // stride, (W + stride - 1) // stride)


def key_is_channels_last(w):
    return w.dim() == 4 and not w.is_contiguous() and w.is_contiguous(memory_format=torch.channels_last)


class BasicStem(nn.Module):
    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.conv1 = ConvBN(in_channels, out_channels, 7, stride=2, padding=3)

    def forward(self, x):
        """conv then ONE pass: folded-BN shift + ReLU + 3x3/2 max-pool (the 1.8 GB stem output is read once)."""
        c = self.conv1
        w, b = c.folded()
        if c.kernel_size == (7, 7) and c.stride == (2, 2) and c.padding == (3, 3) and x.is_cuda and not key_is_channels_last(w):
            return Fn.bias_relu_maxpool(Fn.conv7x7s2_stem(x, w), b)          # own direct kernel where the shape is served
        return Fn.bias_relu_maxpool(F.conv2d(x, w, None, c.stride, c.padding), b)


class BottleneckBlock(nn.Module):
    def __init__(self, cin, cout, bottleneck, stride):
        super().__init__()
        self.shortcut = ConvBN(cin, cout, 1, stride=stride) if cin != cout else None
        self.conv1 = ConvBN(cin, bottleneck, 1, stride=1)                       # STRIDE_IN_1X1 False:
        self.conv2 = ConvBN(bottleneck, bottleneck, 3, stride=stride, padding=1)  # stride on the 3x3
        self.conv3 = ConvBN(bottleneck, cout, 1)

    def _plain(self):
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        return all(c.dilation == (1, 1) and c.groups == 1 for c in (c1, c2, c3)) and c1.kernel_size == (1, 1) and c1.stride == (1, 1) \
            and c1.padding == (0, 0) and c2.kernel_size == (3, 3) and c2.padding == (1, 1) and c2.stride[0] == c2.stride[1] \
            and c2.stride[0] in (1, 2) and c3.kernel_size == (1, 1) and c3.stride == (1, 1) and c3.padding == (0, 0)

    def _forward_images(self, x):
        """The maps INSIDE the bottleneck as operand images (csrc/conv1x1_x3.hip, IMGIN / IMGOUT): conv1's epilogue splits its
        output once, the nine taps of conv2 and conv3's 1x1 read fragments.  None when the block / shapes are not served."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not torch.is_grad_enabled() and self._plain()):
            return None
        N, C, H, W = x.shape
        (w1, b1), (w2, b2), (w3, b3) = self.conv1.folded(), self.conv2.folded(), self.conv3.folded()
        if any(key_is_channels_last(w) for w in (w1, w2, w3)):
            return None
        M, s = w1.shape[0], self.conv2.stride[0]
        OH, OW = (H + s - 1) // s, (W + s - 1) // s
        if M < 128 or not (Fn.x3_images_ok(N, C, M, H, W, x.device) and Fn.x3_images_ok(N, M, M, H, W, x.device, 9, s)):
            return None
        sc = self.shortcut
        if sc is None:
            if s != 1 or not Fn.x3_images_ok(N, M, w3.shape[0], OH, OW, x.device):
                return None
            a1 = Fn.conv_x3_image(x, w1, b1, None, relu=True, out_image=True)
            a2 = Fn.conv_x3_image(a1, w2, b2, None, relu=True, out_image=True)
            return Fn.conv_x3_image(a2, w3, b3, x, relu=True)
        # projection shortcut: conv1 -> image -> conv2 (fp32 map out) -> conv3 + shortcut as one accumulation (the fp32-map kernel)
        if not (sc.kernel_size == (1, 1) and sc.padding == (0, 0) and sc.stride[0] == sc.stride[1] == s and sc.dilation == (1, 1) and sc.groups == 1):
            return None
        ws, bs = sc.folded()
        if key_is_channels_last(ws):
            return None
        a1 = Fn.conv_x3_image(x, w1, b1, None, relu=True, out_image=True)
        a2 = Fn.conv_x3_image(a1, w2, b2, None, relu=True, stride=s)
        if not Fn.conv1x1_x3_dual_ok(a2, w3, x, ws, s):
            return None
        return Fn.conv1x1_x3_dual(a2, w3, b3, x, ws, bs, relu=True, stride2=s)

    def forward(self, x):
        y = self._forward_images(x)
        if y is not None:
            return y
        out = self.conv1(x, relu=True)
        out = self.conv2(out, relu=True)
        sc, c3 = self.shortcut, self.conv3
        if sc is not None and sc.kernel_size == (1, 1) and sc.padding == (0, 0) and sc.stride[0] == sc.stride[1] \
                and sc.dilation == (1, 1) and sc.groups == 1 and c3.kernel_size == (1, 1) and c3.stride == (1, 1) \
                and c3.dilation == (1, 1) and c3.groups == 1 and x.is_cuda:
            (w3, b3), (ws, bs) = c3.folded(), sc.folded()
            if not key_is_channels_last(w3) and not key_is_channels_last(ws) and Fn.conv1x1_x3_dual_ok(out, w3, x, ws, sc.stride[0]):
                # conv3 and the projection shortcut as ONE accumulation over [conv2's channels | the block input's channels]: the
                # shortcut's map (256 channels at the stage's resolution) is neither written nor read back
                return Fn.conv1x1_x3_dual(out, w3, b3, x, ws, bs, relu=True, stride2=sc.stride[0])
        shortcut = x if sc is None else sc(x)
        return c3(out, res=shortcut, relu=True)


class ResNet(nn.Module):
    def __init__(self, depths=(3, 4, 6, 3), out_features=("res2", "res3", "res4", "res5")):
        super().__init__()
        self.stem = BasicStem()
        cin, self._out_features = 64, list(out_features)
        self.stage_names = []
        for i, (n, bott, cout) in enumerate(zip(depths, (64, 128, 256, 512), (256, 512, 1024, 2048))):
            blocks = []
            for b in range(n):
                blocks.append(BottleneckBlock(cin, cout, bott, stride=(1 if i == 0 or b > 0 else 2)))
                cin = cout
            name = f"res{i + 2}"
            self.add_module(name, nn.Sequential(*blocks))
            self.stage_names.append(name)
        self.size_divisibility = 0
        self._out_channels = dict(zip(self.stage_names, (256, 512, 1024, 2048)))

    def output_shape(self):
        return {k: ShapeSpec(channels=self._out_channels[k], stride=2 ** int(k[3:]))
                for k in self._out_features}

    @staticmethod
    def _chain_blocks(stage):
        """The stage's bottlenecks as folded (weight, shift) dicts for Fn.bneck_stage_x3, or None when a block is not the plain
        stride-1 1x1 / 3x3 / 1x1 form."""
        blocks = []
        for b in stage:
            convs = [b.conv1, b.conv2, b.conv3] + ([b.shortcut] if b.shortcut is not None else [])
            if any(c.stride != (1, 1) or c.dilation != (1, 1) or c.groups != 1 for c in convs) or b.conv2.padding != (1, 1) \
                    or any(c.padding != (0, 0) for c in (b.conv1, b.conv3)):
                return None
            (w1, b1), (w2, b2), (w3, b3) = b.conv1.folded(), b.conv2.folded(), b.conv3.folded()
            ws, bs = b.shortcut.folded() if b.shortcut is not None else (None, None)
            if any(key_is_channels_last(w) for w in (w1, w2, w3) + ((ws,) if ws is not None else ())):
                return None
            blocks.append(dict(w1=w1, b1=b1, w2=w2, b2=b2, w3=w3, b3=b3, ws=ws, bs=bs))
        return blocks

    @Fn.fp32_island
    def forward(self, x):
        out = {}
        x = self.stem(Fn.f32(x))
        for name in self.stage_names:
            stage = getattr(self, name)
            blocks = self._chain_blocks(stage) if name == "res2" and x.is_cuda and not torch.is_grad_enabled() else None
            if blocks is not None and Fn.bneck_stage_x3_ok(x, blocks):
                # res2 as a chain: one launch per bottleneck (conv2 -> conv3 + shortcut -> the next block's conv1), the 64-channel
                # maps between them as operand images (csrc/bneck_x3.hip)
                x = Fn.bneck_stage_x3(x, blocks)
            else:
                x = stage(x)
            if name in self._out_features:
                out[name] = x
        return out


def build_resnet50():
    return ResNet((3, 4, 6, 3))


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape=None):
    """detectron2's ``build_resnet_backbone`` for the configuration every R50 yaml of the reference uses
    (configs/dvis_Plus/*/Base-*.yaml: DEPTH 50, STRIDE_IN_1X1 False, FrozenBN, res2..res5).  Other ResNet settings are
    detectron2's to build."""
    r = cfg.MODEL.get("RESNETS", {}) if hasattr(cfg.MODEL, "get") else getattr(cfg.MODEL, "RESNETS", {})
    get = (lambda k, d: r.get(k, d)) if hasattr(r, "get") else (lambda k, d: getattr(r, k, d))
    if int(get("DEPTH", 50)) != 50 or bool(get("STRIDE_IN_1X1", False)) or get("NORM", "FrozenBN") != "FrozenBN":
        raise NotImplementedError("dvis_plus_amd serves ResNet-50 with FrozenBN and STRIDE_IN_1X1 False "
                                  "(the reference's R50 configs); use detectron2's builder for other ResNets")
    return ResNet((3, 4, 6, 3), tuple(get("OUT_FEATURES", ["res2", "res3", "res4", "res5"])))
