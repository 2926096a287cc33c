"""CPU restatement (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py) of the reference's DINOv2 ViT + ViT-Adapter
backbone, eval mode, keyed by the reference's state_dict names:

  DinoVisionTransformer          mask2former/modeling/backbones_vitAdapter/backbones.py:36-260
    PatchEmbed                   .../layers/patch_embed.py:26-80
    NestedTensorBlock / Block    .../layers/block.py:36-104   (pre-norm, LayerScale, eval branch :101-103)
    Attention (no xformers)      .../layers/attention.py:29-61
    Mlp                          .../layers/mlp.py:17-41       (exact-erf GELU)
  DinoV2ViTAdapter.forward       .../adapter.py:528-586
    deform_inputs                .../adapter.py:40-59
    SpatialPriorModule           .../adapter.py:323-385        (SyncBatchNorm in eval = batch_norm with running stats)
    InteractionBlockWithCls_Efficient (extractors only)        .../adapter.py:262-321
    Extractor / ConvFFN / DWConv .../adapter.py:62-137

Pinned against tests/golden/g8_vit_adapter.npz (generated from the imported reference, tests/golden/gen_golden.py).
"""
import math

import torch
import torch.nn.functional as F

from .dvis_torch import linear, ms_deform_attn_module

LN_EPS = 1e-6      # norm_layer = partial(nn.LayerNorm, eps=1e-6), backbones.py:110, adapter.py:109


def ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, 1e-5)


def interpolate_pos_encoding(pos_embed, npatch, w, h, patch):
    """backbones.py:176-202 — note the reference names the image HEIGHT `w` and the width `h` (:205)."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    class_pos, patch_pos = pos_embed[:, 0], pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    w0, h0 = w // patch + 0.1, h // patch + 0.1
    s = int(math.sqrt(N))
    patch_pos = F.interpolate(patch_pos.reshape(1, s, s, dim).permute(0, 3, 1, 2),
                              scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    return torch.cat((class_pos.unsqueeze(0), patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def prepare_tokens(sd, x, p="vit_module."):
    """prepare_tokens_with_masks(x, None, return_HW=True), backbones.py:204-219 -> (B, 1 + HW, C), H, W."""
    W_, b_ = sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"]
    patch = W_.shape[-1]
    t = F.conv2d(x, W_, b_, stride=patch)
    H, W = t.shape[-2:]
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd[p + "cls_token"].expand(t.shape[0], -1, -1), t), dim=1)
    return t + interpolate_pos_encoding(sd[p + "pos_embed"], t.shape[1] - 1, x.shape[2], x.shape[3], patch), H, W


def vit_block(sd, p, x, heads):
    B, N, C = x.shape
    qkv = linear(sd, p + ".attn.qkv", ln(sd, p + ".norm1", x)).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    a = linear(sd, p + ".attn.proj", (a @ v).transpose(1, 2).reshape(B, N, C))
    x = x + a * sd[p + ".ls1.gamma"]
    h = linear(sd, p + ".mlp.fc2", F.gelu(linear(sd, p + ".mlp.fc1", ln(sd, p + ".norm2", x))))
    return x + h * sd[p + ".ls2.gamma"]


def reference_points(shapes):
    """get_reference_points, adapter.py:24-37 -> (1, sum HW, 1, 2) pixel centres in (x, y)."""
    out = []
    for H_, W_ in shapes:
        ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
        out.append(torch.stack((rx.reshape(-1)[None] / W_, ry.reshape(-1)[None] / H_), -1))
    return torch.cat(out, 1)[:, :, None]


def dwconv(sd, p, x, H, W):
    """DWConv.forward, adapter.py:87-98: the 21n tokens are the stride-8 / 16 / 32 maps (16n + 4n + n)."""
    B, N, C = x.shape
    n = N // 21
    w, b = sd[p + ".dwconv.weight"], sd[p + ".dwconv.bias"]
    outs = []
    for sl, (h_, w_) in ((slice(0, 16 * n), (H * 2, W * 2)), (slice(16 * n, 20 * n), (H, W)),
                         (slice(20 * n, N), (H // 2, W // 2))):
        m = x[:, sl].transpose(1, 2).reshape(B, C, h_, w_)
        outs.append(F.conv2d(m, w, b, 1, 1, 1, C).flatten(2).transpose(1, 2))
    return torch.cat(outs, dim=1)


def extractor(sd, p, c, ref, feat, shape, H, W, heads, points):
    """Extractor.forward, adapter.py:119-137 (with_cffn=True, drop_path = identity in eval)."""
    attn = ms_deform_attn_module(sd, p + ".attn", ln(sd, p + ".query_norm", c), ref, ln(sd, p + ".feat_norm", feat),
                                 [shape], heads, points)
    c = c + attn
    h = linear(sd, p + ".ffn.fc1", ln(sd, p + ".ffn_norm", c))
    h = F.gelu(dwconv(sd, p + ".ffn.dwconv", h, H, W))
    return c + linear(sd, p + ".ffn.fc2", h)


def spm(sd, x, p="spm."):
    def cbr(x, conv, norm, stride):
        return F.relu(bn(sd, p + norm, F.conv2d(x, sd[p + conv + ".weight"], None, stride, 1)))
    c1 = cbr(x, "stem.0", "stem.1", 2)
    c1 = cbr(c1, "stem.3", "stem.4", 1)
    c1 = cbr(c1, "stem.6", "stem.7", 1)
    c1 = F.max_pool2d(c1, 3, 2, 1)
    c2 = cbr(c1, "conv2.0", "conv2.1", 2)
    c3 = cbr(c2, "conv3.0", "conv3.1", 2)
    c4 = cbr(c3, "conv4.0", "conv4.1", 2)
    fc = lambda t, n: F.conv2d(t, sd[p + n + ".weight"], sd[p + n + ".bias"])
    c1, c2, c3, c4 = fc(c1, "fc1"), fc(c2, "fc2"), fc(c3, "fc3"), fc(c4, "fc4")
    tok = lambda t: t.flatten(2).transpose(1, 2)
    return c1, tok(c2), tok(c3), tok(c4)


def vit_adapter_forward(sd, x, *, heads, deform_heads, interaction_indexes, n_points=4, stages=None):
    """DinoV2ViTAdapter.forward (eval) -> [f1, f2, f3, f4] at strides 4 / 8 / 16 / 32."""
    bs, _, h, w = x.shape
    ref = reference_points([(h // 8, w // 8), (h // 16, w // 16), (h // 32, w // 32)])
    c1, c2, c3, c4 = spm(sd, x)
    n2, n3 = c2.shape[1], c3.shape[1]
    c = torch.cat([c2 + sd["level_embed"][0], c3 + sd["level_embed"][1], c4 + sd["level_embed"][2]], dim=1)
    t, H, W = prepare_tokens(sd, x)
    if stages is not None:
        stages["tokens"] = t
    dim = t.shape[-1]
    outs = []
    for i, (lo, hi) in enumerate(interaction_indexes):
        for b in range(lo, hi + 1):
            t = vit_block(sd, f"vit_module.blocks.{b}", t, heads)
            if stages is not None and b == 0:
                stages["block0"] = t
        xs = t[:, 1:]
        p = f"interactions.{i}"
        c = extractor(sd, p + ".extractor", c, ref, xs, (H, W), H, W, deform_heads, n_points)
        k = 0
        while f"{p}.extra_extractors.{k}.query_norm.weight" in sd:
            c = extractor(sd, f"{p}.extra_extractors.{k}", c, ref, xs, (H, W), H, W, deform_heads, n_points)
            k += 1
        outs.append(xs.transpose(1, 2).reshape(bs, dim, H, W))
    c2, c3, c4 = c[:, :n2], c[:, n2:n2 + n3], c[:, n2 + n3:]
    c2 = c2.transpose(1, 2).reshape(bs, dim, H * 2, W * 2)
    c3 = c3.transpose(1, 2).reshape(bs, dim, H, W)
    c4 = c4.transpose(1, 2).reshape(bs, dim, H // 2, W // 2)
    c1 = F.conv_transpose2d(c2, sd["up.weight"], sd["up.bias"], 2) + c1
    x1, x2, x3, x4 = outs                                              # add_vit_feature=True
    c1 = c1 + F.interpolate(x1, scale_factor=4, mode="bilinear", align_corners=False)
    c2 = c2 + F.interpolate(x2, scale_factor=2, mode="bilinear", align_corners=False)
    c3 = c3 + x3
    c4 = c4 + F.interpolate(x4, scale_factor=0.5, mode="bilinear", align_corners=False)
    return [bn(sd, "norm1", c1), bn(sd, "norm2", c2), bn(sd, "norm3", c3), bn(sd, "norm4", c4)]
