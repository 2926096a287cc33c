"""csrc/gemm_x3.hip at the encoder's shapes: error against fp64 (next to the fp32 library GEMM's on the same data) and time.
    python tools/x3_time.py [frames]
Per layer of the deformable encoder (30 frames x 19 320 tokens): value_proj 256 -> 256, offsets | logits 256 -> 288,
output_proj + residual + LayerNorm, FFN 256 -> 1024 -> 256 + residual + LayerNorm."""
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, ".")
from dvis_plus_amd import functions as Fn  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rel(c, ref, scale):
    return float(((c.double() - ref).abs() / scale).max())


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    S = 19320
    M = S * frames
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(frames, S, 256, device=DEV, generator=g)
    x[0, :64] *= torch.logspace(-6, 2, 64, device=DEV)[:, None]          # rows of very different magnitude
    res = torch.randn(frames, S, 256, device=DEV, generator=g)
    pos = torch.randn(1, S, 256, device=DEV, generator=g)
    R = 4096                                                              # rows checked against fp64
    sel = torch.cat([torch.arange(64, R // 2), torch.arange(M - R // 2, M)]).to(DEV)
    tiny = torch.arange(0, 64, device=DEV)                                # the rows scaled by 1e-6 ... 1e2
    print(f"M = {M} tokens ({frames} frames); errors = max |c - fp64| / (sum|a||w| + |bias|) over {R - 64} rows of unit scale; "
          f"'scaled rows' = the 64 rows multiplied by 1e-6 ... 1e2: elements below 2^-3 / 2^xexp = 0.0078 keep an ABSOLUTE precision "
          f"of 2^-25 / 2^xexp = 1.9e-9 (f16 subnormal low term), so relative to such a row's own scale the error grows as the row shrinks")
    for N in (256, 288, 128, 192):
        lin = nn.Linear(256, N).to(DEV)
        nn.init.xavier_uniform_(lin.weight)
        nn.init.normal_(lin.bias)
        xs = x.view(M, 256)[sel]
        ref = xs.double() @ lin.weight.double().t() + lin.bias.double()
        scale = xs.double().abs() @ lin.weight.double().abs().t() + lin.bias.double().abs()
        for relu in (False, True):
            got = Fn.x3_linear(x, lin.weight, lin.bias, relu=relu).view(M, N)[sel]
            lib = F.linear(xs, lin.weight, lin.bias)
            r = ref.clamp_min(0) if relu else ref
            print(f"  linear 256 -> {N} relu={int(relu)}: x3 {rel(got, r, scale):.2e}   fp32 library {rel(F.relu(lib) if relu else lib, r, scale):.2e}")
        xt = x.view(M, 256)[tiny]
        reft = xt.double() @ lin.weight.double().t() + lin.bias.double()
        gott = Fn.x3_linear(x, lin.weight, lin.bias).view(M, N)[tiny]
        libt = F.linear(xt, lin.weight, lin.bias)
        scalet = xt.double().abs() @ lin.weight.double().abs().t() + lin.bias.double().abs()
        print(f"    scaled rows: max |c - fp64| absolute: x3 {float((gott.double() - reft).abs().max()):.2e}   fp32 library "
              f"{float((libt.double() - reft).abs().max()):.2e};  relative to the row's own sum|a||w| + |bias|: x3 "
              f"{rel(gott, reft, scalet):.2e}   fp32 library {rel(libt, reft, scalet):.2e}")
        t = timeit(lambda: Fn.x3_linear(x, lin.weight, lin.bias))
        tl = timeit(lambda: F.linear(x, lin.weight, lin.bias))
        fl = 2.0 * M * 256 * N
        gb = 4.0 * M * (256 + N)
        print(f"    x3 {t:.3f} ms ({fl / t / 1e9:.0f} TF fp32-equivalent, {gb / t / 1e6:.0f} GB/s)   library fp32 {tl:.3f} ms ({fl / tl / 1e9:.0f} TF)")
    # output_proj + residual + LayerNorm (+ pos)
    lin = nn.Linear(256, 256).to(DEV)
    nn.init.xavier_uniform_(lin.weight)
    norm = nn.LayerNorm(256).to(DEV)
    nn.init.normal_(norm.weight, 1.0, 0.2)
    nn.init.normal_(norm.bias, 0.0, 0.2)
    xs, rs = x.view(M, 256)[sel], res.view(M, 256)[sel]
    ps = pos.view(S, 256)[sel % S]
    ref = F.layer_norm(rs.double() + xs.double() @ lin.weight.double().t() + lin.bias.double(), (256,), norm.weight.double(),
                       norm.bias.double(), norm.eps)
    out, out2 = Fn.x3_linear_ln(x, lin.weight, lin.bias, res, norm, pos=pos)
    lib = norm(rs + F.linear(xs, lin.weight, lin.bias))
    print(f"  linear + res + LN: x3 max abs err {float((out.view(M, 256)[sel].double() - ref).abs().max()):.2e}, "
          f"out + pos {float((out2.view(M, 256)[sel].double() - ref - ps.double()).abs().max()):.2e}; fp32 torch {float((lib.double() - ref).abs().max()):.2e}")
    t = timeit(lambda: Fn.x3_linear_ln(x, lin.weight, lin.bias, res, norm, pos=pos))
    t1 = timeit(lambda: Fn.x3_linear_ln(x, lin.weight, lin.bias, res, norm))
    tl = timeit(lambda: Fn.add_layer_norm(F.linear(x, lin.weight, lin.bias), res, norm, pos=pos))
    print(f"    x3 {t:.3f} ms with out + pos, {t1:.3f} ms without; library GEMM + add_layernorm kernel {tl:.3f} ms")
    # FFN
    l1, l2 = nn.Linear(256, 1024).to(DEV), nn.Linear(1024, 256).to(DEV)
    nn.init.xavier_uniform_(l1.weight)
    nn.init.xavier_uniform_(l2.weight)
    nn.init.normal_(l1.bias, 0.0, 0.5)
    nn.init.normal_(l2.bias, 0.0, 0.5)
    h = F.relu(xs.double() @ l1.weight.double().t() + l1.bias.double())
    ref = F.layer_norm(xs.double() + h @ l2.weight.double().t() + l2.bias.double(), (256,), norm.weight.double(), norm.bias.double(),
                       norm.eps)
    out, out2 = Fn.x3_ffn_ln(x, l1, l2, norm, pos=pos)
    lib = norm(xs + l2(F.relu(l1(xs))))
    print(f"  FFN + res + LN: x3 max abs err {float((out.view(M, 256)[sel].double() - ref).abs().max()):.2e}, "
          f"out + pos {float((out2.view(M, 256)[sel].double() - ref - ps.double()).abs().max()):.2e}; fp32 torch {float((lib.double() - ref).abs().max()):.2e}")
    t = timeit(lambda: Fn.x3_ffn_ln(x, l1, l2, norm, pos=pos), 5)
    t1 = timeit(lambda: Fn.x3_ffn_ln(x, l1, l2, norm), 5)
    tl = timeit(lambda: Fn.add_layer_norm(Fn.linear(Fn.linear_relu(x, l1), l2.weight, l2.bias), x, norm, pos=pos), 5)
    fl = 2.0 * M * 256 * 1024 * 2
    print(f"    x3 {t:.3f} ms with out + pos ({fl / t / 1e9:.0f} TF fp32-equivalent), {t1:.3f} ms without; "
          f"library GEMMs + add_layernorm kernel {tl:.3f} ms ({fl / tl / 1e9:.0f} TF)")
    # run-to-run bits
    a, b = Fn.x3_ffn_ln(x, l1, l2, norm), Fn.x3_ffn_ln(x, l1, l2, norm)
    print("  FFN bit-reproducible:", bool(torch.equal(a, b)))
    # a ragged tail and a tiny M
    for m in (1, 33, 127, 129, 300):
        xm = x.view(M, 256)[:m].contiguous()
        o = Fn.x3_ffn_ln(xm, l1, l2, norm)
        r = norm(xm + l2(F.relu(l1(xm))))
        print(f"  M = {m}: FFN max abs diff vs torch fp32 {float((o - r).abs().max()):.2e}")


def conv_main():
    """the R50's compute-bound 1x1 layers at 30 frames of 736 x 1280: csrc/conv1x1_x3.hip vs csrc/conv1x1_mfma.hip (exact fp32)"""
    shapes = [(512, 128, 92, 160, 1), (128, 512, 92, 160, 1), (256, 512, 184, 320, 2), (1024, 256, 46, 80, 1), (256, 1024, 46, 80, 1),
              (512, 1024, 92, 160, 2), (2048, 512, 23, 40, 1), (512, 2048, 23, 40, 1), (1024, 2048, 46, 80, 2), (512, 256, 92, 160, 1),
              (1024, 512, 46, 80, 1),
              # the layers csrc/conv1x1.hip (weights resident in LDS, exact fp32) claims: res2 conv1 / conv3, res3.0 conv1
              (256, 64, 184, 320, 1), (256, 128, 184, 320, 1), (64, 256, 184, 320, 1), (64, 64, 184, 320, 1)]
    for Ci, Co, H, W, stride in shapes:
        x = torch.randn(30, Ci, H, W, device=DEV)
        w = torch.randn(Co, Ci, 1, 1, device=DEV) * (2.0 / Ci) ** 0.5
        b = torch.randn(Co, device=DEV)
        OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
        r = torch.randn(30, Co, OH, OW, device=DEV) if stride == 1 and Co > Ci else None
        if not Fn.conv1x1_x3_ok(x, w, stride, r):
            print(f"  conv {Ci} -> {Co} {H}x{W} s{stride}: not served")
            continue
        t = timeit(lambda: Fn.conv1x1_x3(x, w, b, r, True, stride))
        tm = timeit(lambda: Fn.conv1x1_mfma(x, w, b, r, True, stride)) if Ci % 128 == 0 else float("nan")
        fl = 2.0 * 30 * OH * OW * Ci * Co
        gb = 4.0 * 30 * (H * W * Ci / (stride * stride) + OH * OW * Co * (2 if r is not None else 1))
        line = (f"  conv {Ci:4d} -> {Co:4d} {H}x{W} s{stride} res={int(r is not None)}: x3 {t:.3f} ms ({fl / t / 1e9:.0f} TF fp32-eq, "
                f"{gb / t / 1e6:.0f} GB/s)   exact-fp32 MFMA kernel {tm:.3f} ms ({fl / tm / 1e9:.0f} TF)")
        if stride == 1 and Fn.native.lib().dvis_conv1x1_supported(Ci, Co, H * W):
            Fn.X3 = False                                                   # -> csrc/conv1x1.hip (weights resident in LDS)
            tb = timeit(lambda: Fn.conv1x1_bias_act(x, w, b, r, True))
            Fn.X3 = True
            line += f"   exact-fp32 LDS-weights kernel {tb:.3f} ms ({gb / tb / 1e6:.0f} GB/s)"
        print(line)


def conv3_main():
    """the R50's 3x3 layers and the FPN output convolution at 30 frames of 736 x 1280: nine-tap split-f16 kernel vs the exact-fp32
    Winograd / direct stride-2 kernels"""
    shapes = [(64, 64, 184, 320, 1), (128, 128, 184, 320, 2), (128, 128, 92, 160, 1), (256, 256, 92, 160, 2), (256, 256, 46, 80, 1),
              (512, 512, 46, 80, 2), (512, 512, 23, 40, 1), (256, 256, 184, 320, 1)]
    for Ci, Co, H, W, stride in shapes:
        x = torch.randn(30, Ci, H, W, device=DEV)
        w = torch.randn(Co, Ci, 3, 3, device=DEV) * (2.0 / (9 * Ci)) ** 0.5
        b = torch.randn(Co, device=DEV)
        OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
        if not Fn.conv3x3_x3_ok(x, w, stride):
            print(f"  conv3x3 {Ci} -> {Co} {H}x{W} s{stride}: not served")
            continue
        t = timeit(lambda: Fn.conv3x3_x3(x, w, b, None, True, stride))
        Fn.X3 = False                                       # the dispatchers would hand these layers to the x3 kernel again
        te = timeit(lambda: Fn.conv3x3_bias_act(x, w, b, True)) if stride == 1 else timeit(lambda: Fn.conv3x3s2_bias_act(x, w, b, True))
        Fn.X3 = True
        fl = 2.0 * 30 * OH * OW * Ci * Co * 9
        print(f"  conv3x3 {Ci:4d} -> {Co:4d} {H}x{W} s{stride}: x3 nine taps {t:.3f} ms ({fl / t / 1e9:.0f} TF fp32-eq direct)"
              f"   exact-fp32 {'Winograd' if stride == 1 else 'direct'} kernel {te:.3f} ms ({fl / te / 1e9:.0f} TF direct-eq)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "conv3":
        with torch.no_grad():
            conv3_main()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "conv":
        with torch.no_grad():
            conv_main()
        sys.exit(0)
    main()
