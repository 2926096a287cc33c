// 7x7 / stride 2 / pad 3 convolution with 3 input and 64 output channels, NCHW fp32, direct on the fp32 matrix cores (round 3):
// the ResNet stem (detectron2 BasicStem.conv1).  MIOpen runs it as a VALU Winograd at 56 TFLOP/s (2.4 ms per
// 30 frames of 736 x 1280).  Same skeleton as csrc/conv3x3s2.hip, with ONE stage: the contraction has only 3 * 7 * 7 = 147 rows
//     out[k][pixel] = sum_{c, ky, kx} W[k][c][ky][kx] * x[c][2 oy - 3 + ky][2 ox - 3 + kx],      row = (7 c + ky) * 7 + kx.
//   * workgroup = 64 consecutive output pixels x the 64 output channels, 8 waves; LDS holds the 148 x 64 im2col block (37 KB).
//   * wave w fills the rows of the (c, ky) groups w, w + 8, w + 16: per pixel two 16-byte loads of input row 2 oy - 3 + ky from
//     column max(2 ox - 3, 0) — a load never starts in front of its row; the two leftmost pixels shift by 3 / 1 in registers,
//     the rightmost masks columns W, W + 1 — and 7 LDS writes.
//   * wave w = (16 output channels, half of the 37 k-steps); the halves meet through LDS; weights come packed in the MFMA A
//     layout (dvis_conv7x7s2_pack), 20 floats per lane.
// A workgroup loads, then contracts: the overlap comes from the 3 - 4 workgroups a CU holds.  Fixed accumulation order.
#include "dvis_common.h"

namespace {

constexpr int kPix = 64, kRowsPad = 148, kSteps = 37;
constexpr unsigned kOOB = 0x80000000u;

struct StemArgs {
  const float *x, *uf, *bias;
  float *y;
  int N, H, W, OH, OW, relu, nsp;
  long long pixels;
};

__global__ __launch_bounds__(512, 2) void conv7x7s2_kernel(const StemArgs a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int kb16 = wv & 3, half = wv >> 2;
  const int sp = blockIdx.x;
  const long long p0 = (long long)sp * kPix;
  const int per_img = a.OH * a.OW;
  const int n0 = (int)(p0 / per_img);
  const long long plane = (long long)a.H * a.W, img = plane * 3;
  const int n_here = min(2, a.N - n0);
  const __amdgpu_buffer_rsrc_t rx = dvis_make_rsrc_uniform(a.x + (long long)n0 * img, (unsigned)(n_here * img * 4));
  const __amdgpu_buffer_rsrc_t ru = dvis_make_rsrc_uniform(a.uf, (unsigned)(4 * 2 * 64 * 20 * 4));

  // weights of this wave: 20 floats per lane (k-steps 19 half .. of its 16 output channels), requested first
  dvis_f4 u[5];
#pragma unroll
  for (int q = 0; q < 5; ++q)
    u[q] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(ru, (unsigned)lane * 16u + 1024u * q,
                                                                             (unsigned)(kb16 * 2 + half) * 5120u, 0));

  // ---- fill role: lane = output pixel
  const long long p = p0 + lane;
  const bool pv = p < a.pixels;
  const int pi = pv ? (int)(p - (long long)n0 * per_img) : 0;
  const int nn = pi / per_img, r = pi - nn * per_img;
  const int oy = r / a.OW, ox = r - oy * a.OW;
  const int start = max(2 * ox - 3, 0), sh = start - (2 * ox - 3);            // sh = 3, 1 or 0
  const unsigned m0 = sh == 0 ? ~0u : 0u, m1 = sh == 1 ? ~0u : 0u, m3 = sh == 3 ? ~0u : 0u;
  const unsigned mr = 2 * ox + 2 >= a.W ? 0u : ~0u;                           // columns 2 ox + 2, 2 ox + 3 (kx = 5, 6) inside?
  dvis_f4 q[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int grp = wv + 8 * t;                                               // (c, ky) group, wave-uniform
    const int c = grp / 7, ky = grp - 7 * c;
    const int yy = 2 * oy - 3 + ky;
    const bool ok = pv && grp < 21 && yy >= 0 && yy < a.H;
    const unsigned off = ok ? (unsigned)(((long long)nn * img + c * plane + (long long)yy * a.W + start) * 4) : kOOB;
    q[t][0] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    q[t][1] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off + 16u : kOOB, 0, 0));
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int grp = wv + 8 * t;
    if (grp >= 21) break;                                                     // wave-uniform
    unsigned v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(q[t][0][e]), v[4 + e] = __float_as_uint(q[t][1][e]);
    float *vw = lds + (grp * 7) * kPix + lane;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {   // tap kx = column 2 ox - 3 + kx = loaded element kx - sh
      unsigned e = v[kx] & m0;
      if (kx >= 1) e |= v[kx - 1] & m1;
      if (kx >= 3) e |= v[kx - 3] & m3;
      if (kx >= 5) e &= mr;
      vw[kx * kPix] = __uint_as_float(e);
    }
  }
  if (wv == 0) lds[147 * kPix + lane] = 0.f;                                   // the padding row of the last k-step
  __syncthreads();

  // ---- contraction: k-steps [19 half, 19 half + 19) of 37 (the second half has 18)
  dvis_f4 acc[4];
#pragma unroll
  for (int tb = 0; tb < 4; ++tb) acc[tb] = dvis_f4{0.f, 0.f, 0.f, 0.f};
  const dvis_f4 *vr = reinterpret_cast<const dvis_f4 *>(lds + (half * 76 + g) * kPix + 4 * j);
  dvis_f4 b[2];
  b[0] = vr[0];
#pragma unroll
  for (int s = 0; s < 19; ++s) {
    if (s == 18 && half) break;                                               // wave-uniform
    if (s + 1 < 19) b[(s + 1) & 1] = vr[min(s + 1, half ? 17 : 18) * kPix];
    const float av = u[s >> 2][s & 3];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[s & 1][tb], acc[tb], 0, 0, 0);
  }

  // ---- the halves' partial sums through LDS; half h stores accumulator tiles 2 h, 2 h + 1 (pixels 4 j + tb)
  __syncthreads();
  {
    float *ex = lds + (((1 - half) * 4 + kb16) * 2) * 4 * 64 + lane;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) ex[(t2 * 4 + rr) * 64] = half ? acc[t2][rr] : acc[2 + t2][rr];
  }
  __syncthreads();
  const float *ex = lds + ((half * 4 + kb16) * 2) * 4 * 64 + lane;
  const int k0 = kb16 * 16 + 4 * g;
  const long long pa = p0 + 4 * j + 2 * half;
  if (pa >= a.pixels) return;
  const int n = (int)(pa / per_img), ro = (int)(pa - (long long)n * per_img);
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    float o[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const float mine = half ? acc[2 + t2][rr] : acc[t2][rr], other = ex[(t2 * 4 + rr) * 64];
      const float v = (half ? other + mine : mine + other) + (a.bias ? a.bias[k0 + rr] : 0.f);
      o[t2] = a.relu ? fmaxf(v, 0.f) : v;
    }
    *reinterpret_cast<float2 *>(a.y + ((long long)n * 64 + k0 + rr) * per_img + ro) = make_float2(o[0], o[1]);
  }
}

// uf[kb16][half][q (5)][lane = 16 g + i][e (4)] = W[k = 16 kb16 + i][row = 4 (19 half + 4 q + e) + g] (0 for row >= 147 or a
// k-step past the half's end), row = (7 c + ky) * 7 + kx = the flat index of the (3, 7, 7) filter
__global__ void conv7x7s2_pack_kernel(const float *__restrict__ w, float *__restrict__ uf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4 * 2 * 5 * 64 * 4) return;
  int t = idx;
  const int e = t & 3; t >>= 2;
  const int ln = t & 63; t >>= 6;
  const int q = t % 5; t /= 5;
  const int half = t & 1, kb16 = t >> 1;
  const int s = 4 * q + e, g = ln >> 4, i16 = ln & 15;
  const int step = 19 * half + s, row = 4 * step + g;
  float v = 0.f;
  if (s < (half ? 18 : 19) && row < 147) v = w[(kb16 * 16 + i16) * 147 + row];
  uf[idx] = v;
}

}  // namespace

DVIS_EXPORT int dvis_conv7x7s2_supported(int C, int K, int H, int W) {
  if (C != 3 || K != 64 || H < 8 || W < 8 || (W & 1) || (H & 1)) return 0;
  if ((long long)(H / 2) * (W / 2) < kPix || 2ll * 3 * H * W * 4 >= (1ll << 31)) return 0;
  return 1;
}

DVIS_EXPORT int dvis_conv7x7s2_pack(const float *w, float *uf, void *stream) {
  DVIS_REQUIRE(w && uf, "conv7x7s2_pack: null pointer");
  hipLaunchKernelGGL(conv7x7s2_pack_kernel, dim3(40), dim3(256), 0, (hipStream_t)stream, w, uf);
  return dvis_check_launch("dvis_conv7x7s2_pack");
}

DVIS_EXPORT int dvis_conv7x7s2(const float *x, const float *uf, const float *bias, float *y, int N, int H, int W, int relu,
                               void *stream) {
  DVIS_REQUIRE(N >= 0, "conv7x7s2: bad batch");
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(x && uf && y, "conv7x7s2: null pointer");
  DVIS_REQUIRE(dvis_conv7x7s2_supported(3, 64, H, W), "conv7x7s2: unsupported size H=%d W=%d (dvis_conv7x7s2_supported)", H, W);
  DVIS_REQUIRE((((uintptr_t)x | (uintptr_t)uf | (uintptr_t)y) & 15) == 0, "conv7x7s2: 16-byte aligned tensors");
  StemArgs a;
  a.x = x, a.uf = uf, a.bias = bias, a.y = y;
  a.N = N, a.H = H, a.W = W, a.relu = relu, a.OH = H / 2, a.OW = W / 2;
  a.pixels = (long long)N * a.OH * a.OW;
  const long long nsp = (a.pixels + kPix - 1) / kPix;
  DVIS_REQUIRE(nsp < (1ll << 31), "conv7x7s2: grid too large");
  a.nsp = (int)nsp;
  hipLaunchKernelGGL(conv7x7s2_kernel, dim3((unsigned)nsp), dim3(512), kRowsPad * kPix * sizeof(float), (hipStream_t)stream, a);
  return dvis_check_launch("dvis_conv7x7s2");
}
