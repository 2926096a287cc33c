"""Event-timed breakdown of the pixel decoder / backbone building blocks at the BASELINE shape (dev tool)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dvis_plus_amd.meta_architecture import build_dvis_plus_r50  # noqa: E402

dev = torch.device("cuda", 0)
m = build_dvis_plus_r50("offline").to(dev)
pd = m.sem_seg_head.pixel_decoder
N = 30


def t(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    feats = {"res2": torch.randn(N, 256, 184, 320, device=dev), "res3": torch.randn(N, 512, 92, 160, device=dev),
             "res4": torch.randn(N, 1024, 46, 80, device=dev), "res5": torch.randn(N, 2048, 23, 40, device=dev)}
    print(f"pixel decoder total        {t(lambda: pd.forward_features(feats)):8.2f} ms")
    for i, f in enumerate(["res5", "res4", "res3"]):
        print(f"input_proj[{f}] conv+GN    {t(lambda: pd.input_proj[i](feats[f])):8.2f} ms")
    S, C = 19320, 256
    src = torch.randn(N, S, C, device=dev)
    pos = torch.randn(1, S, C, device=dev)
    layer = pd.transformer.encoder.layers[0]
    shapes_py = [(23, 40), (46, 80), (92, 160)]
    ss, lsi = pd.transformer._shape_tensors(shapes_py, dev)
    ref = pd.transformer.encoder.reference_points_unpadded(shapes_py, dev)
    print(f"encoder layer              {t(lambda: layer(src, pos, ref, ss, lsi, None, shapes_py=shapes_py)):8.2f} ms (x6)")
    at = layer.self_attn
    print(f"  src + pos                {t(lambda: src + pos):8.2f} ms")
    print(f"  value_proj               {t(lambda: at.value_proj(src)):8.2f} ms")
    w, b = at._fused_projection()
    print(f"  offsets|logits proj      {t(lambda: F.linear(src.view(-1, C), w, b)):8.2f} ms")
    print(f"  self_attn (all)          {t(lambda: at(src, ref, src, ss, lsi, None, spatial_shapes_py=shapes_py)):8.2f} ms")
    print(f"  output_proj              {t(lambda: at.output_proj(src)):8.2f} ms")
    print(f"  add + LayerNorm          {t(lambda: layer.norm1(src + src)):8.2f} ms")
    print(f"  linear1                  {t(lambda: layer.linear1(src)):8.2f} ms")
    h = layer.linear1(src)
    print(f"  relu                     {t(lambda: F.relu(h)):8.2f} ms")
    print(f"  linear1+relu (addmm_act) {t(lambda: torch._addmm_activation(layer.linear1.bias, src.view(-1, C), layer.linear1.weight.t())):8.2f} ms")
    print(f"  linear2                  {t(lambda: layer.linear2(h)):8.2f} ms")
    x2 = feats["res2"]
    print(f"FPN lateral 1x1+GN         {t(lambda: pd.lateral_convs[0](x2)):8.2f} ms")
    y = torch.randn(N, 256, 184, 320, device=dev)
    print(f"FPN 3x3 conv+GN+relu       {t(lambda: pd.output_convs[0](y)):8.2f} ms")
    print(f"  3x3 conv only            {t(lambda: F.conv2d(y, pd.output_convs[0].weight, None, 1, 1)):8.2f} ms")
    ycl = y.contiguous(memory_format=torch.channels_last)
    wcl = pd.output_convs[0].weight.detach().contiguous(memory_format=torch.channels_last)
    print(f"  3x3 conv channels_last   {t(lambda: F.conv2d(ycl, wcl, None, 1, 1)):8.2f} ms")
    print(f"  GroupNorm                {t(lambda: pd.output_convs[0].norm(y)):8.2f} ms")
    print(f"  upsample + add           {t(lambda: y + F.interpolate(feats['res3'][:, :256], size=(184, 320), mode='bilinear', align_corners=False)):8.2f} ms")
    print(f"mask_features 1x1          {t(lambda: pd.mask_features(y)):8.2f} ms")
    # backbone pieces
    bb = m.backbone
    img = torch.randn(N, 3, 736, 1280, device=dev)
    print(f"backbone total             {t(lambda: bb(img)):8.2f} ms")
    print(f"  stem                     {t(lambda: bb.stem(img)):8.2f} ms")
    x = bb.stem(img)
    for name in bb.stage_names:
        st = getattr(bb, name)
        print(f"  {name}                     {t(lambda: st(x)):8.2f} ms")
        x = st(x)
    imgcl = img.contiguous(memory_format=torch.channels_last)
    bbcl = build_dvis_plus_r50("offline").backbone.to(dev).to(memory_format=torch.channels_last)
    print(f"backbone channels_last     {t(lambda: bbcl(imgcl)):8.2f} ms")
    blk = bb.res2[1]
    xr = torch.randn(N, 256, 184, 320, device=dev)
    print(f"  res2 block               {t(lambda: blk(xr)):8.2f} ms")
    print(f"    conv1 (1x1 256->64)    {t(lambda: blk.conv1(xr)):8.2f} ms")
    h1 = blk.conv1(xr)
    print(f"    relu_ on 64ch          {t(lambda: F.relu_(h1)):8.2f} ms")
    print(f"    conv2 (3x3 64->64)     {t(lambda: blk.conv2(h1)):8.2f} ms")
    print(f"    conv3 (1x1 64->256)    {t(lambda: blk.conv3(h1)):8.2f} ms")
    print(f"    add + relu 256ch       {t(lambda: F.relu_(xr + xr)):8.2f} ms")
    try:
        w1, b1 = blk.conv1.folded()
        print(f"    miopen_convolution_relu conv1  {t(lambda: torch.ops.aten.miopen_convolution_relu(xr, w1, b1, [1, 1], [0, 0], [1, 1], 1)):8.2f} ms")
        w3, b3 = blk.conv3.folded()
        print(f"    miopen_convolution_add_relu c3 {t(lambda: torch.ops.aten.miopen_convolution_add_relu(h1, w3, xr, 1.0, b3, [1, 1], [0, 0], [1, 1], 1)):8.2f} ms")
    except Exception as e:  # noqa: BLE001
        print("    miopen fused ops unavailable:", repr(e)[:200])
