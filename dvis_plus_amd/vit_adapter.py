"""DINOv2 ViT + ViT-Adapter backbone (inference) — SURVEY.md §8 row f-4, BASELINE config #5.

Mirrors the reference's vendored backbone: same constructor arguments, parameter names / `state_dict` keys and outputs
  DinoVisionTransformer, vit_large         mask2former/modeling/backbones_vitAdapter/backbones.py:36-260, 300-311
  PatchEmbed / Block / Attention / Mlp     .../layers/{patch_embed,block,attention,mlp,layer_scale}.py
  DinoV2ViTAdapter (+ SpatialPriorModule, InteractionBlockWithCls_Efficient, Extractor, ConvFFN, DWConv)
                                           .../adapter.py:62-137, 262-385, 422-586
  D2VitAdapterDinoV2 (detectron2 backbone wrapper: dict res2..res5, 1024 channels each)        .../adapter.py:589-651

What runs where: the token attention (3681 tokens x 16 heads x d=64 at 720p) is the repo's fp32-MFMA attention kernel on
strided views of the fused qkv projection (the reference materialises the (B,16,N,N) probabilities when xformers is
absent, attention.py:49-61); the extractors call the MSDeformAttn op through its fused single-level path
(D=64, L=1, P=4); projections / MLPs are library GEMMs; convolutions MIOpen.  LayerScale is folded into the
projection that precedes it once per weight version (like FrozenBN in backbone.py).  Eval only: drop-path is the
identity, SyncBatchNorm is batch-norm with running statistics.
"""
import math
import os
from functools import partial

import torch
import torch.nn.functional as F
from torch import nn

from . import functions as Fn
from .pixel_decoder import MSDeformAttn
from .registry import BACKBONE_REGISTRY, ShapeSpec

LayerNorm = partial(nn.LayerNorm, eps=1e-6)


# ------------------------------------------------------------------------------------------------ DINOv2 ViT
class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def _flat_weight(self):
        """The stride-p p x p convolution as a linear layer over flattened patches: (D, 3 p p), an own tensor (not a view: the
        packed split-f16 image is cached per weight tensor), refreshed when the parameter changes."""
        w = self.proj.weight
        key = (w._version, w.data_ptr(), w.device)
        if getattr(self, "_flat", None) is None or self._flat[0] != key:
            self._flat = (key, w.detach().reshape(w.shape[0], -1).clone())
        return self._flat[1]

    def forward(self, x, return_HW=False):
        B, Cin, H, W = x.shape
        ph, pw = self.patch_size
        assert H % ph == 0 and W % pw == 0, "image size must be a multiple of the patch"
        if x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not torch.is_autocast_enabled() \
                and self.proj.stride == self.proj.kernel_size and self.proj.padding == (0, 0) and (Cin * ph * pw) % 4 == 0:
            # non-overlapping patches: ONE gather into (B h w, 3 p p) rows and a tall GEMM (split-f16 tiled kernel from K = 512 on)
            # whose output already is the token matrix — no library convolution, no flatten / transpose copy
            h, w = H // ph, W // pw
            rows = x.view(B, Cin, h, ph, w, pw).permute(0, 2, 4, 1, 3, 5).reshape(B * h * w, Cin * ph * pw)
            t = Fn.linear(rows, self._flat_weight(), self.proj.bias, tall=True).view(B, h * w, -1)
            return (t, h, w) if return_HW else t
        x = self.proj(x)
        H, W = x.shape[-2:]
        x = x.flatten(2).transpose(1, 2)
        return (x, H, W) if return_HW else x


class LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Folded:
    """out-projection with the LayerScale gamma that follows it folded in: gamma * (W a + b) = (gamma W) a + gamma b."""

    def __init__(self):
        self._cache = None

    def get(self, lin, ls):
        g = ls.gamma if isinstance(ls, LayerScale) else None
        key = (lin.weight._version, lin.bias._version if lin.bias is not None else -1,
               g._version if g is not None else -1, lin.weight.device)
        if self._cache is None or self._cache[0] != key:
            w, b = lin.weight.detach(), None if lin.bias is None else lin.bias.detach()
            if g is not None:
                w = w * g.detach()[:, None]
                b = None if b is None else b * g.detach()
            self._cache = (key, w.contiguous(), b)
        return self._cache[1], self._cache[2]


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)

    def core(self, x):
        """(B, N, C) -> attention output before the out-projection, (B, N, C)."""
        B, N, C = x.shape
        if Fn.x3_qkv_attention_ok(x, self.qkv.weight, self.num_heads):
            # the projection's epilogue writes the attention kernel's split-f16 operands: no fp32 qkv tensor, no pack pass
            return Fn.x3_qkv_attention(x, self.qkv.weight, self.qkv.bias, self.num_heads)
        qkv = Fn.linear(x, self.qkv.weight, self.qkv.bias, tall=True)                       # (B, N, 3C)
        v = qkv.transpose(0, 1)                                                    # (N, B, 3C) view: rows strided
        out = torch.empty((B, N, C), dtype=x.dtype, device=x.device)
        Fn.attention(v[..., :C], v[..., C:2 * C], v[..., 2 * C:], self.num_heads, out=out.transpose(0, 1))
        return out

    def forward(self, x):
        return Fn.linear(self.core(x), self.proj.weight, self.proj.bias, tall=True)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, bias=True):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features, bias=bias)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features, bias=bias)

    def hidden(self, x):
        return Fn.linear(x, self.fc1.weight, self.fc1.bias, tall=True, act="gelu")            # exact (erf) GELU as nn.GELU()

    def forward(self, x):
        return Fn.linear(self.hidden(x), self.fc2.weight, self.fc2.bias, tall=True)


class Block(nn.Module):
    """Pre-norm block with LayerScale (block.py:36-104, eval branch)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, proj_bias=True, ffn_bias=True, init_values=None):
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), bias=ffn_bias)
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self._f1, self._f2 = _Folded(), _Folded()

    def row_images_ok(self, x):
        """norm -> GEMM -> ... with row images between the layers (functions.RowImage): every normalisation / GELU / attention output
        is written once, pre-split, as the next GEMM's row operand."""
        if x.dim() != 3 or not (Fn.layer_norm_rows_image_ok(x, self.norm1) and self.norm2.weight is not None and self.norm2.bias is not None):
            return False
        B, L, C = x.shape
        a, m = self.attn, self.mlp
        H = m.fc1.weight.shape[0]
        return (C >= Fn.X3_TILE_MIN_K and C % 256 == 0 and H % 256 == 0 and m.fc2.weight.shape == (C, H) and a.proj.weight.shape == (C, C)
                and m.fc1.bias is not None and Fn.x3_qkv_attention_ok(Fn.RowImage(None, x.shape, 0, 0), a.qkv.weight, a.num_heads))

    def forward(self, x):
        if self.row_images_ok(x):
            a, m = self.attn, self.mlp
            h = Fn.x3_qkv_attention(Fn.layer_norm_rows_image(x, self.norm1), a.qkv.weight, a.qkv.bias, a.num_heads, out_image=True)
            w, b = self._f1.get(a.proj, self.ls1)
            x = Fn.x3_tile_linear(h, w, b, residual=x)
            h = Fn.x3_tile_linear(Fn.layer_norm_rows_image(x, self.norm2), m.fc1.weight, m.fc1.bias, act="gelu")
            w, b = self._f2.get(m.fc2, self.ls2)
            return Fn.x3_tile_linear(h, w, b, residual=x)
        w, b = self._f1.get(self.attn.proj, self.ls1)
        x = Fn.linear(self.attn.core(Fn.add_layer_norm(x, None, self.norm1)), w, b, tall=True, residual=x)
        w, b = self._f2.get(self.mlp.fc2, self.ls2)
        return Fn.linear(self.mlp.hidden(Fn.add_layer_norm(x, None, self.norm2)), w, b, tall=True, residual=x)


class DinoVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, ffn_bias=True, proj_bias=True, drop_path_rate=0.0, init_values=None, ffn_layer="mlp",
                 block_chunks=0):
        super().__init__()
        if ffn_layer != "mlp" or block_chunks != 0:
            raise NotImplementedError("the DVIS++ ViT-Adapter configs use ffn_layer='mlp', block_chunks=0 (backbones.py:403-412)")
        self.configs_dict = dict(img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth,
                                 num_heads=num_heads, mlp_ratio=mlp_ratio, drop_path_rate=drop_path_rate,
                                 init_values=init_values)
        self.norm_layer = LayerNorm
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens, self.n_blocks, self.num_heads, self.patch_size = 1, depth, num_heads, patch_size
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, proj_bias, ffn_bias, init_values)
                                     for _ in range(depth)])
        self.norm = LayerNorm(embed_dim)
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        self._pos_cache = {}

    def interpolate_pos_encoding(self, npatch, w, h):
        """backbones.py:176-202; (w, h) are the image height / width as the reference names them.  Cached per size."""
        N = self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        key = (w, h, self.pos_embed._version, self.pos_embed.device)
        if key not in self._pos_cache:
            pe = self.pos_embed.detach().float()
            dim = pe.shape[-1]
            w0, h0 = w // self.patch_size + 0.1, h // self.patch_size + 0.1
            s = int(math.sqrt(N))
            pp = F.interpolate(pe[:, 1:].reshape(1, s, s, dim).permute(0, 3, 1, 2),
                               scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
            assert int(w0) == pp.shape[-2] and int(h0) == pp.shape[-1]
            self._pos_cache = {key: torch.cat((pe[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)}
        return self._pos_cache[key]

    def prepare_tokens_with_masks(self, x, masks=None, return_HW=False):
        assert masks is None, "masked tokens are a DINOv2 pre-training feature, unused by the adapter"
        w, h = x.shape[2:]
        t, H, W = self.patch_embed(x, return_HW=True)
        t = torch.cat((self.cls_token.expand(t.shape[0], -1, -1), t), dim=1)
        t = t + self.interpolate_pos_encoding(t.shape[1] - 1, w, h)
        return (t, H, W) if return_HW else t


def vit_large(patch_size=16, **kw):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4, **kw)


def vit_base(patch_size=16, **kw):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, **kw)


def get_models(name="vitl"):
    """backbones.py:400-418 (weights are loaded through the model's state_dict, not here)."""
    return {"vitl": vit_large, "vitb": vit_base}[name](img_size=592, patch_size=16, init_values=1.0e-05, ffn_layer="mlp",
                                                       block_chunks=0, qkv_bias=True, proj_bias=True, ffn_bias=True)


# ------------------------------------------------------------------------------------------------ adapter
def get_reference_points(spatial_shapes, device):
    out = []
    for H_, W_ in spatial_shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
        out.append(torch.stack((rx.reshape(-1)[None] / W_, ry.reshape(-1)[None] / H_), -1))
    return torch.cat(out, 1)[:, :, None]


class DWConv(nn.Module):
    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def levels(self, N, H, W):
        return [(H * 2, W * 2), (H, W), (H // 2, W // 2)]

    def fused_ok(self, x, H, W):
        return (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not torch.is_autocast_enabled()
                and x.is_contiguous() and x.shape[2] % 4 == 0 and H % 2 == 0 and W % 2 == 0 and x.shape[1] == 21 * (H // 2) * (W // 2)
                and os.environ.get("DVIS_DWCONV_TOKENS", "1") != "0")

    def forward(self, x, H, W):
        B, N, C = x.shape
        n = N // 21
        outs = []
        for lo, hi, (h_, w_) in ((0, 16 * n, (H * 2, W * 2)), (16 * n, 20 * n, (H, W)), (20 * n, N, (H // 2, W // 2))):
            m = x[:, lo:hi].transpose(1, 2).reshape(B, C, h_, w_)
            outs.append(self.dwconv(m).flatten(2).transpose(1, 2))
        return torch.cat(outs, dim=1)


class ConvFFN(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.dwconv = DWConv(hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x, H, W, residual=None):
        if isinstance(x, Fn.RowImage):          # (the normalisation in front wrote fc1's row operand: Extractor.forward)
            x = Fn.x3_tile_linear(x, self.fc1.weight, self.fc1.bias)
        else:
            x = Fn.linear(x, self.fc1.weight, self.fc1.bias, tall=True)
        if self.dwconv.fused_ok(x, H, W):
            # depthwise 3x3 + bias + GELU on the token tensor itself, level by level (no NCHW round trip, no concatenation)
            x = Fn.dwconv3x3_tokens(x, self.dwconv.levels(x.shape[1], H, W), self.dwconv.dwconv.weight, self.dwconv.dwconv.bias, gelu=True)
        else:
            x = F.gelu(self.dwconv(x, H, W))
        return Fn.linear(x, self.fc2.weight, self.fc2.bias, tall=True, residual=residual)


class Extractor(nn.Module):
    def __init__(self, dim, num_heads=6, n_points=4, n_levels=1, deform_ratio=1.0, with_cffn=True, cffn_ratio=0.25):
        super().__init__()
        self.query_norm, self.feat_norm = LayerNorm(dim), LayerNorm(dim)
        self.attn = MSDeformAttn(d_model=dim, n_levels=n_levels, n_heads=num_heads, n_points=n_points, ratio=deform_ratio)
        self.with_cffn = with_cffn
        if with_cffn:
            self.ffn = ConvFFN(in_features=dim, hidden_features=int(dim * cffn_ratio))
            self.ffn_norm = LayerNorm(dim)

    def forward(self, query, reference_points, feat, spatial_shapes, level_start_index, H, W, shapes_py=None):
        query = self.attn(Fn.add_layer_norm(query, None, self.query_norm), reference_points,
                          Fn.add_layer_norm(feat, None, self.feat_norm), spatial_shapes, level_start_index, None,
                          post=(query, None))                      # query + attn: the add in the output projection's epilogue
        if self.with_cffn:
            fc1 = self.ffn.fc1
            if (Fn.layer_norm_rows_image_ok(query, self.ffn_norm) and fc1.weight.shape[1] >= Fn.X3_TILE_MIN_K and fc1.weight._base is None
                    and bool(Fn.native.lib().dvis_x3_tile_supported(fc1.weight.shape[0], fc1.weight.shape[1]))):
                normed = Fn.layer_norm_rows_image(query, self.ffn_norm)
            else:
                normed = Fn.add_layer_norm(query, None, self.ffn_norm)
            query = self.ffn(normed, H, W, residual=query)
        return query


class InteractionBlockWithCls_Efficient(nn.Module):
    """ViT blocks of one stage, then the extractor(s) — adapter.py:262-321 (no injector in this variant)."""

    def __init__(self, dim, num_heads=6, n_points=4, with_cffn=True, cffn_ratio=0.25, deform_ratio=1.0,
                 extra_extractor=False):
        super().__init__()
        mk = lambda: Extractor(dim=dim, n_levels=1, num_heads=num_heads, n_points=n_points, deform_ratio=deform_ratio,
                               with_cffn=with_cffn, cffn_ratio=cffn_ratio)
        self.extractor = mk()
        self.extra_extractors = nn.Sequential(mk(), mk()) if extra_extractor else None

    def forward(self, x, c, cls, blocks, deform_inputs2, H, W):
        x = torch.cat((cls, x), dim=1)
        for blk in blocks:
            x = blk(x)
        cls, x = x[:, :1], x[:, 1:].contiguous()
        ref, shapes, lsi = deform_inputs2
        c = self.extractor(c, ref, x, shapes, lsi, H, W)
        if self.extra_extractors is not None:
            for ex in self.extra_extractors:
                c = ex(c, ref, x, shapes, lsi, H, W)
        return x, c, cls


class SpatialPriorModule(nn.Module):
    def __init__(self, inplanes=64, embed_dim=384):
        super().__init__()
        cbr = lambda ci, co, s: [nn.Conv2d(ci, co, kernel_size=3, stride=s, padding=1, bias=False), nn.BatchNorm2d(co),
                                 nn.ReLU(inplace=True)]
        self.stem = nn.Sequential(*cbr(3, inplanes, 2), *cbr(inplanes, inplanes, 1), *cbr(inplanes, inplanes, 1),
                                  nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
        self.conv2 = nn.Sequential(*cbr(inplanes, 2 * inplanes, 2))
        self.conv3 = nn.Sequential(*cbr(2 * inplanes, 4 * inplanes, 2))
        self.conv4 = nn.Sequential(*cbr(4 * inplanes, 4 * inplanes, 2))
        self.fc1 = nn.Conv2d(inplanes, embed_dim, kernel_size=1)
        self.fc2 = nn.Conv2d(2 * inplanes, embed_dim, kernel_size=1)
        self.fc3 = nn.Conv2d(4 * inplanes, embed_dim, kernel_size=1)
        self.fc4 = nn.Conv2d(4 * inplanes, embed_dim, kernel_size=1)

    def forward(self, x):
        c1 = self.stem(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        tok = lambda t: t.flatten(2).transpose(1, 2)
        return self.fc1(c1), tok(self.fc2(c2)), tok(self.fc3(c3)), tok(self.fc4(c4))

    # ---- the same module on the repo's own convolution kernels (round 6: no MIOpen / hipBLASLt call left in the stem)
    def _folded(self, conv, bn, embed7=False):
        """(weight, bias) of `conv` with the eval-mode BatchNorm that follows folded in (y = s (W x) + (beta - mean s), s = gamma /
        sqrt(var + eps)), cached per parameter version.  embed7: the 3 x 3 / stride 2 / padding 1 kernel laid into the centre of a
        7 x 7 / padding 3 one (the same convolution) — the shape the direct stem kernel serves (csrc/conv7x7s2.hip)."""
        key = (conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
               conv.weight.device, embed7)
        cache = self.__dict__.setdefault("_fold_cache", {})
        ent = cache.get(id(conv))
        if ent is None or ent[0] != key:
            with torch.no_grad():
                s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
                w = (conv.weight.double() * s.view(-1, 1, 1, 1)).float()
                b = (bn.bias.double() - bn.running_mean.double() * s).float().contiguous()
                if embed7:
                    w7 = w.new_zeros(w.shape[0], w.shape[1], 7, 7)
                    w7[:, :, 2:5, 2:5] = w
                    w = w7
                cache[id(conv)] = ent = (key, w.contiguous(), b)
        return ent[1], ent[2]

    def own_ok(self, x):
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
        return bool(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_grad_enabled() and not self.training
                    and not torch.is_autocast_enabled() and x.shape[2] % 32 == 0 and x.shape[3] % 32 == 0
                    and all(b.track_running_stats and b.running_mean is not None and b.weight is not None for b in bns)
                    and os.environ.get("DVIS_SPM_OWN", "1") != "0")

    def forward_own(self, x, level_embed):
        """-> (c1 (B, D, H/4, W/4), c (B, n2 + n3 + n4, D) = cat(tok(c2) + le[0], tok(c3) + le[1], tok(c4) + le[2]), n2, n3): the
        module's four outputs as the adapter consumes them (adapter.py:558-565), every convolution on an own kernel with the
        BatchNorm folded, ReLU (and the max-pool) in the epilogues, the level embeddings in the projections' biases and the three
        token blocks written straight into one buffer (no torch.cat)."""
        st = self.stem
        w, b = self._folded(st[0], st[1], embed7=True)
        y = Fn.conv7x7s2_stem(x, w, b, relu=True)                                   # 3 -> 64, stride 2
        w, b = self._folded(st[3], st[4])
        y = Fn.conv3x3_bias_act(y, w, b, relu=True)
        w, b = self._folded(st[6], st[7])
        c1 = Fn.bias_relu_maxpool(Fn.conv3x3_bias_act(y, w, None, relu=False), b)   # + shift, ReLU, 3 x 3 / 2 max-pool in one pass
        c2 = Fn.conv3x3s2_bias_act(c1, *self._folded(self.conv2[0], self.conv2[1]), relu=True)
        c3 = Fn.conv3x3s2_bias_act(c2, *self._folded(self.conv3[0], self.conv3[1]), relu=True)
        c4 = Fn.conv3x3s2_bias_act(c3, *self._folded(self.conv4[0], self.conv4[1]), relu=True)
        f1 = Fn.conv1x1_bias_act(c1, self.fc1.weight, self.fc1.bias)
        le = level_embed.detach()
        maps = [Fn.conv1x1_bias_act(c, fc.weight, fc.bias.detach() + le[i]) for i, (c, fc) in
                enumerate(((c2, self.fc2), (c3, self.fc3), (c4, self.fc4)))]
        return f1, Fn.maps_to_tokens(maps), maps[0].shape[2] * maps[0].shape[3], maps[1].shape[2] * maps[1].shape[3]


class DinoV2ViTAdapter(nn.Module):
    def __init__(self, vit_module=None, pretrain_size=224, conv_inplane=64, n_points=4, deform_num_heads=6,
                 init_values=0.0, interaction_indexes=None, with_cffn=True, cffn_ratio=0.25, deform_ratio=1.0,
                 add_vit_feature=True, use_extra_extractor=True, with_cp=False, freeze_backbone=False, finetune=False,
                 finetune_indexes=(0,)):
        super().__init__()
        self.pretrain_size = (pretrain_size, pretrain_size)
        self.interaction_indexes, self.add_vit_feature = interaction_indexes, add_vit_feature
        dim = vit_module.embed_dim
        self.level_embed = nn.Parameter(torch.zeros(3, dim))
        self.spm = SpatialPriorModule(inplanes=conv_inplane, embed_dim=dim)
        self.interactions = nn.Sequential(*[
            InteractionBlockWithCls_Efficient(dim=dim, num_heads=deform_num_heads, n_points=n_points, with_cffn=with_cffn,
                                              cffn_ratio=cffn_ratio, deform_ratio=deform_ratio,
                                              extra_extractor=(i == len(interaction_indexes) - 1 and use_extra_extractor))
            for i in range(len(interaction_indexes))])
        self.up = nn.ConvTranspose2d(dim, dim, 2, 2)
        self.norm1, self.norm2, self.norm3, self.norm4 = (nn.BatchNorm2d(dim) for _ in range(4))
        for m in list(self.up.modules()) + list(self.spm.modules()) + list(self.interactions.modules()):
            self._init_weights(m)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.normal_(self.level_embed)
        self.vit_module = vit_module
        self._deform_cache = {}

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                m.bias.data.zero_()

    def _deform_inputs(self, h, w, device):
        """deform_inputs2 of adapter.py:40-59: 3-scale query reference points against the single stride-16 value map."""
        key = (h, w, device)
        if key not in self._deform_cache:
            shapes = torch.as_tensor([(h // 16, w // 16)], dtype=torch.long, device=device)
            lsi = shapes.new_zeros((1,))
            ref = get_reference_points([(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)], device).contiguous()
            self._deform_cache = {key: (ref, shapes, lsi)}
        return self._deform_cache[key]

    @Fn.fp32_island
    def forward(self, x):
        if self.training:
            raise NotImplementedError("dvis_plus_amd implements the backbone's inference path")
        x = Fn.f32(x)
        bs, _, h, w = x.shape
        d2 = self._deform_inputs(h, w, x.device)
        if self.spm.own_ok(x):
            c1, c, n2, n3 = self.spm.forward_own(x, self.level_embed)
        else:
            c1, c2, c3, c4 = self.spm(x)
            n2, n3 = c2.shape[1], c3.shape[1]
            c = torch.cat([c2 + self.level_embed[0], c3 + self.level_embed[1], c4 + self.level_embed[2]], dim=1)
        t, H, W = self.vit_module.prepare_tokens_with_masks(x, masks=None, return_HW=True)
        dim = t.shape[-1]
        cls, t = t[:, :1], t[:, 1:]
        outs, x1_tok, outs_tok = [], None, []
        for i, layer in enumerate(self.interactions):
            lo, hi = self.interaction_indexes[i]
            t, c, cls = layer(t, c, cls, self.vit_module.blocks[lo:hi + 1], d2, H, W)
            outs_tok.append(t)
            if i == 0:
                x1_tok = t
            outs.append(t.transpose(1, 2).reshape(bs, dim, H, W) if i > 0 or not self._res2_fused_ok(c1, t) else None)
        c2_tok, c3, c4 = c[:, :n2], c[:, n2:n2 + n3], c[:, n2 + n3:]
        c2 = c2_tok.transpose(1, 2).reshape(bs, dim, H * 2, W * 2)
        c3 = c3.transpose(1, 2).reshape(bs, dim, H, W)
        c4 = c4.transpose(1, 2).reshape(bs, dim, H // 2, W // 2)
        f1 = None
        if outs[0] is None:
            # stride-4 output in two launches: `up` as a GEMM over the stride-8 tokens (BatchNorm scale folded into its weights),
            # then pixel shuffle + SPM feature + 4x-upsampled ViT feature + BatchNorm shift in one pass (Fn.adapter_res2)
            w_l, scale, shift = self._res2_folded()
            g = Fn.linear(c2_tok.reshape(bs * n2, dim), w_l, None, tall=True)
            f1 = Fn.adapter_res2(g, c1, x1_tok if self.add_vit_feature else None, scale, shift, 2 * H, 2 * W)
        else:
            c1 = self.up(c2) + c1
        if f1 is not None and self.add_vit_feature and self._tail_fused_ok(c, H, W):
            # stride 8 / 16 / 32 outputs without a library kernel (round 6): eval BatchNorm is a per-channel affine, so
            #   norm2(c2 + up2(x2)) = (s c2 + t) + up2(s x2)            -> Fn.upsample_add with the lateral's affine applied on the fly
            #   norm3(c3 + x3)      = s (c3 + x3) + t                   (token-major, then one tiled transpose)
            #   norm4(c4 + down2(x4)) with down2 = the mean of every 2 x 2 block (what bilinear, align_corners=False samples at 1/2)
            t2, t3, t4 = (tok.contiguous() for tok in (outs_tok[1], outs_tok[2], outs_tok[3]))
            (s2, b2), (s3, b3), (s4, b4) = (self._bn_affine(bn) for bn in (self.norm2, self.norm3, self.norm4))
            c_all = c.contiguous()
            lat2 = Fn.tokens_to_map(c_all, 0, 2 * H, 2 * W)
            top2 = Fn.tokens_to_map(t2 * s2, 0, H, W)
            f2 = Fn.upsample_add(lat2, top2, (s2.repeat(bs).contiguous(), b2.repeat(bs).contiguous()))
            f3 = Fn.tokens_to_map(((c_all[:, n2:n2 + n3] + t3) * s3 + b3).contiguous(), 0, H, W)
            x4d = t4.view(bs, H // 2, 2, W // 2, 2, dim).mean(dim=(2, 4)).reshape(bs, -1, dim)
            f4 = Fn.tokens_to_map(((c_all[:, n2 + n3:] + x4d) * s4 + b4).contiguous(), 0, H // 2, W // 2)
            return [f1, f2, f3, f4]
        if self.add_vit_feature:
            x1, x2, x3, x4 = outs
            if f1 is None:
                c1 = c1 + F.interpolate(x1, scale_factor=4, mode="bilinear", align_corners=False)
            c2 = c2 + F.interpolate(x2, scale_factor=2, mode="bilinear", align_corners=False)
            c3 = c3 + x3
            c4 = c4 + F.interpolate(x4, scale_factor=0.5, mode="bilinear", align_corners=False)
        return [self.norm1(c1) if f1 is None else f1, self.norm2(c2), self.norm3(c3), self.norm4(c4)]

    def _tail_fused_ok(self, c, H, W):
        bns = (self.norm2, self.norm3, self.norm4)
        return (c.is_cuda and c.dtype == torch.float32 and not torch.is_grad_enabled() and H % 2 == 0 and W % 2 == 0
                and all(isinstance(b, nn.BatchNorm2d) and b.track_running_stats and b.running_mean is not None and b.weight is not None
                        for b in bns) and os.environ.get("DVIS_ADAPTER_TAIL", "1") != "0")

    def _bn_affine(self, bn):
        """Eval-mode BatchNorm as (scale, shift) per channel, fp64 arithmetic rounded once, cached per parameter version."""
        key = tuple(p._version for p in (bn.weight, bn.bias, bn.running_mean, bn.running_var)) + (bn.weight.device,)
        cache = self.__dict__.setdefault("_bn_cache", {})
        ent = cache.get(id(bn))
        if ent is None or ent[0] != key:
            with torch.no_grad():
                s = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
                cache[id(bn)] = ent = (key, s.float().contiguous(), (bn.bias.double() - bn.running_mean.double() * s).float().contiguous())
        return ent[1], ent[2]

    def _res2_fused_ok(self, c1, t):
        bn = self.norm1
        return (c1.is_cuda and not torch.is_grad_enabled() and not torch.is_autocast_enabled() and c1.dtype == torch.float32
                and t.dtype == torch.float32 and c1.is_contiguous() and c1.shape[1] % 64 == 0 and c1.shape[2] % 4 == 0
                and c1.shape[3] % 4 == 0 and c1.shape[0] * (c1.shape[2] // 2) <= 65535 and isinstance(bn, nn.BatchNorm2d)
                and bn.track_running_stats and bn.running_mean is not None and os.environ.get("DVIS_ADAPTER_RES2", "1") != "0")

    def _res2_folded(self):
        """(W_l, scale, shift): ConvTranspose2d(C, C, 2, 2) as a linear layer over the stride-8 tokens, rows ordered (dy, dx, co),
        with norm1's eval affine folded in: W_l[(dy, dx, co), ci] = s[co] * up.weight[ci, co, dy, dx]; shift = s * up.bias + beta -
        mean * s.  Cached per parameter version."""
        up, bn = self.up, self.norm1
        key = tuple(p._version for p in (up.weight, up.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)) + (up.weight.device,)
        if getattr(self, "_res2_cache", None) is None or self._res2_cache[0] != key:
            with torch.no_grad():
                s = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
                sh = s * up.bias.double() + bn.bias.double() - bn.running_mean.double() * s
                C = up.weight.shape[1]
                w_l = (up.weight.double() * s.view(1, C, 1, 1)).permute(2, 3, 1, 0).reshape(4 * C, up.weight.shape[0])
                self._res2_cache = (key, w_l.float().contiguous(), s.float().contiguous(), sh.float().contiguous())
        return self._res2_cache[1:]


def get_adapter_args(name="vitl"):
    """adapter.py:387-433."""
    vit = get_models(name)
    return dict(vit_module=vit, pretrain_size=vit.configs_dict["img_size"], init_values=1e-6, conv_inplane=64, n_points=4,
                deform_num_heads={"vitl": 16, "vitb": 12}[name], with_cffn=True, cffn_ratio=0.25, deform_ratio=0.5,
                add_vit_feature=True, use_extra_extractor=True,
                interaction_indexes={"vitl": [[0, 5], [6, 11], [12, 17], [18, 23]],
                                     "vitb": [[0, 2], [3, 5], [6, 8], [9, 11]]}[name])


@BACKBONE_REGISTRY.register()
class D2VitAdapterDinoV2(DinoV2ViTAdapter):
    """detectron2 backbone surface: forward -> {res2..res5}, output_shape(), size_divisibility (adapter.py:589-651).
    Built like the reference's, ``D2VitAdapterDinoV2(cfg, input_shape)`` (size from cfg.MODEL.VIT_ADAPTER.NAME; the
    weight-file / freezing / checkpointing keys are training-time settings), or directly by size name."""

    def __init__(self, name="vitl", input_shape=None, **overrides):
        if not isinstance(name, str):                       # a config node
            name = name.MODEL.VIT_ADAPTER.NAME
        args = get_adapter_args(name)
        args.update(overrides)
        super().__init__(**args)
        self._out_features = ["res2", "res3", "res4", "res5"]
        self._out_feature_strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}

    def forward(self, x):
        assert x.dim() == 4
        return dict(zip(self._out_features, super().forward(x)))

    def output_shape(self):
        return {k: ShapeSpec(channels=self.vit_module.embed_dim, stride=s) for k, s in self._out_feature_strides.items()}

    @property
    def size_divisibility(self):
        return 32
