// The arithmetic of torch's CPU kernels, operation for operation — gfx950 device functions.
//
// The reference runs its video post-processing on CPU tensors (dvis_Plus/meta_architecture.py:1473, :802-804 move
// everything to the host; inference_video_{vis,vps,vss} :818-979 then call F.interpolate / sigmoid / argmax there), and
// BASELINE.json asks for bit-exact index / argmax outputs.  An argmax is only as reproducible as the floats it
// compares, so the two bilinear resizes and the sigmoid are evaluated here in exactly the order, with exactly the
// fused-multiply-adds, of the torch CPU build the reference runs on (torch 2.10, AVX512 kernels, GCC -ffp-contract=fast).
// Pinned empirically against torch itself (0 differing bits on 10^6..10^7 values per case; the probes are summarised in
// DESIGN.md section 5.1 and re-checked by tests/test_postprocess_gpu.py against torch's CPU ops on the GPU box's host):
//
//  upsample_bilinear2d, align_corners=False (aten/native/cpu/UpSampleKernel.cpp)
//    source index   src = fma(scale, dst + 0.5, -0.5), clamped at 0; scale = (float)in / out;  in == out -> copy
//    i0 = min((int)src, in - 1), i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1
//    output height + width  > 128  (generic separable kernel):
//        tA = fma(a, lx0, b * lx1); tB = fma(c, lx0, d * lx1); out = fma(tA, ly0, tB * ly1)
//    output height + width <= 128  ("channels-last" kernel on the (N, C=frames) planes; w_ij = ly_i * lx_j):
//        channel < 16 * (C / 16)  (vector lanes):  fma(w00, a, fma(w01, b, fma(w11, d, w10 * c)))
//        remaining channels       (scalar tail):   fma(w11, d, fma(w10, c, fma(w00, a, w01 * b)))
//  sigmoid (aten/native/cpu/UnaryOpsKernel.cpp): 1 / (1 + exp(0 - x)) with
//    vector lanes: Sleef_expf16_u10 (two-step Cody-Waite reduction, degree-5 polynomial, all FMA)
//    scalar tail (the last n % 32 elements of each contiguous run): glibc expf (correctly rounded for our purposes)
#pragma once
#include <hip/hip_runtime.h>

namespace tcpu {

struct Tap {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Tap make_tap(int dst, float scale, int in, int out) {
#pragma clang fp contract(off)
  Tap t;
  if (in == out) {        // "scale_factor = 1, simply copy"
    t.i0 = t.i1 = dst;
    t.l0 = 1.f;
    t.l1 = 0.f;
    return t;
  }
  float src = __builtin_fmaf(scale, (float)dst + 0.5f, -0.5f);
  src = src < 0.f ? 0.f : src;
  t.i0 = min((int)src, in - 1);
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.l1 = fminf(fmaxf(src - (float)t.i0, 0.f), 1.f);
  t.l0 = 1.f - t.l1;
  return t;
}

enum { kGeneric = 0, kSmallTail = 1, kSmallVec = 2 };

// which of torch's three evaluation orders applies to plane `channel` of `channels` for this output size
__device__ __forceinline__ int bilinear_kind(int out_h, int out_w, int channel, int channels) {
  if (out_h + out_w > 128) return kGeneric;
  return channel < (channels & ~15) ? kSmallVec : kSmallTail;
}

__device__ __forceinline__ float bilinear(float a, float b, float c, float d, const Tap &ty, const Tap &tx, int kind) {
#pragma clang fp contract(off)
  if (kind == kGeneric) {
    const float tA = __builtin_fmaf(a, tx.l0, b * tx.l1);
    const float tB = __builtin_fmaf(c, tx.l0, d * tx.l1);
    return __builtin_fmaf(tA, ty.l0, tB * ty.l1);
  }
  const float w00 = ty.l0 * tx.l0, w01 = ty.l0 * tx.l1, w10 = ty.l1 * tx.l0, w11 = ty.l1 * tx.l1;
  if (kind == kSmallVec) return __builtin_fmaf(w00, a, __builtin_fmaf(w01, b, __builtin_fmaf(w11, d, w10 * c)));
  return __builtin_fmaf(w11, d, __builtin_fmaf(w10, c, __builtin_fmaf(w00, a, w01 * b)));
}

// Sleef_expf_u10 (purecfma form used by the AVX512 build)
__device__ __forceinline__ float sleef_expf(float d) {
#pragma clang fp contract(off)
  const float R_LN2f = 1.442695040888963407359924681001892137426645954152985934135449406931f;
  const float L2Uf = 0.693145751953125f, L2Lf = 1.428606765330187045e-06f;
  const float qf = rintf(d * R_LN2f);
  const int q = (int)qf;
  float s = __builtin_fmaf(qf, -L2Uf, d);
  s = __builtin_fmaf(qf, -L2Lf, s);
  float u = 0.000198527617612853646278381f;
  u = __builtin_fmaf(u, s, 0.00139304355252534151077271f);
  u = __builtin_fmaf(u, s, 0.00833336077630519866943359f);
  u = __builtin_fmaf(u, s, 0.0416664853692054748535156f);
  u = __builtin_fmaf(u, s, 0.166666671633720397949219f);
  u = __builtin_fmaf(u, s, 0.5f);
  u = 1.0f + __builtin_fmaf(s * s, u, s);
  const int q1 = q >> 1, q2 = q - q1;                     // vldexp2: u * 2^(q>>1) * 2^(q - (q>>1))
  u = u * __int_as_float((q1 + 127) << 23);
  u = u * __int_as_float((q2 + 127) << 23);
  if (d < -104.f) u = 0.f;
  if (d > 100.f) u = INFINITY;
  return u;
}

// torch.sigmoid on a float CPU tensor; `scalar_tail`: the element sits in the last n % 32 of its contiguous run
__device__ __forceinline__ float sigmoid(float x, bool scalar_tail) {
#pragma clang fp contract(off)
  const float nx = 0.f - x;
  const float e = scalar_tail ? (float)exp((double)nx) : sleef_expf(nx);
  return 1.f / (1.f + e);
}

// Both stages for one candidate map: stride-4 logits (h, w) -> (first_h, first_w) -> crop (img_h, img_w) [-> sigmoid]
// -> (out_h, out_w).  Geometry that does not depend on the candidate is precomputed per output pixel in Pix.
struct Geometry {
  int h, w, first_h, first_w, img_h, img_w, out_h, out_w;
  float s1y, s1x, s2y, s2x;
  int tail_x;             // first column of the cropped row that torch's sigmoid evaluates with the scalar exp
  __device__ __forceinline__ Geometry(int h_, int w_, int fh, int fw, int ih, int iw, int oh, int ow)
      : h(h_), w(w_), first_h(fh), first_w(fw), img_h(ih), img_w(iw), out_h(oh), out_w(ow) {
    s1y = (float)h / (float)first_h;
    s1x = (float)w / (float)first_w;
    s2y = (float)img_h / (float)out_h;
    s2x = (float)img_w / (float)out_w;
    // a crop in x leaves rows of img_w floats as the contiguous runs of the sigmoid's TensorIterator loop
    tail_x = img_w != first_w ? (img_w & ~31) : 0x7fffffff;
  }
};

// first-stage value at (yy, xx) of the padded-size image, then optionally the sigmoid
template <bool SIGMOID>
__device__ __forceinline__ float stage1(const float *__restrict__ lg, const Geometry &g, int yy, int xx, int kind1) {
  const Tap ty = make_tap(yy, g.s1y, g.h, g.first_h), tx = make_tap(xx, g.s1x, g.w, g.first_w);
  const float a = lg[ty.i0 * g.w + tx.i0], b = lg[ty.i0 * g.w + tx.i1];
  const float c = lg[ty.i1 * g.w + tx.i0], d = lg[ty.i1 * g.w + tx.i1];
  const float v = bilinear(a, b, c, d, ty, tx, kind1);
  return SIGMOID ? sigmoid(v, xx >= g.tail_x) : v;
}

template <bool SIGMOID>
__device__ __forceinline__ float two_stage(const float *__restrict__ lg, const Geometry &g, int Y, int X, const Tap &ty2,
                                           const Tap &tx2, int kind1, int kind2) {
  if (g.img_h == g.out_h && g.img_w == g.out_w) return stage1<SIGMOID>(lg, g, Y, X, kind1);     // second stage copies
  const float a = stage1<SIGMOID>(lg, g, ty2.i0, tx2.i0, kind1), b = stage1<SIGMOID>(lg, g, ty2.i0, tx2.i1, kind1);
  const float c = stage1<SIGMOID>(lg, g, ty2.i1, tx2.i0, kind1), d = stage1<SIGMOID>(lg, g, ty2.i1, tx2.i1, kind1);
  return bilinear(a, b, c, d, ty2, tx2, kind2);
}

}  // namespace tcpu
