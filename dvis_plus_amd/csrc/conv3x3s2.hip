// 3x3 / stride 2 / pad 1 convolution, NCHW fp32, direct on the fp32 matrix cores (round 3).
//
// What it replaces: conv2 of the FIRST bottleneck of res3 / res4 / res5 (detectron2 BottleneckBlock with STRIDE_IN_1X1 = False,
// SURVEY.md App. B) — inside the pipeline MIOpen runs the res3 one (128 -> 128, 184 x 320 -> 92 x 160) as a VALU Winograd at
// 55 TFLOP/s (2.4 ms per 30 frames) and the other two in its implicit-GEMM kernel at 124 TFLOP/s, each followed by the
// bias + ReLU pass.  Winograd buys nothing at stride 2 (one output per 2x2 tile), so this is the plain contraction
//     out[k][pixel] = sum_{tap, c} W[k][tap, c] * X[tap, c][pixel],   X[tap = (i, jj), c][pixel (oy, ox)] = x[c][2 oy - 1 + i][2 ox - 1 + jj]
// built like csrc/winograd_conv.hip (same roles, layouts and tricks; what is different is named here):
//   * workgroup = 64 consecutive output pixels (in (n, oy, ox) order) x 64 output channels, 8 waves; a stage = 8 input channels
//     = 72 contraction rows k = 8 tap + c, double-buffered in LDS as 72 rows of 64 pixels (18 KB per stage; two workgroups per CU).
//   * wave w = (16 output channels kb16 = w & 3, half = w >> 2): the halves split the 18 k-steps of a stage 9 : 9 (no padding,
//     balanced), 4 accumulator tiles per wave (pixel 4 j + tb of lane column j: one ds_read_b128 per k-step serves 4 MFMAs).
//   * wave w loads the 3 x 3 neighbourhoods of channel w of the stage for the 64 pixels — three 16-byte loads per lane (row
//     2 oy - 1 + i from column 2 ox - 1; the 4th float is not used) — and writes the 9 taps to LDS: there is NO transform, so the
//     VALU work per stage is one AND (the column -1 of the leftmost pixel) and address arithmetic.
//   * weights: packed once by dvis_conv3x3s2_pack into the MFMA A layout, 9 floats per lane and stage as three 16-byte loads.
//   * the halves' partial sums meet through LDS (each wave stores 2 of the 4 accumulator tiles: 2 adjacent output pixels per lane).
// Fixed accumulation order: bit-reproducible.
#include "dvis_common.h"

namespace {

constexpr int kPix = 64, kKw = 64, kCc = 8;
constexpr int kRows = 9 * kCc;             // contraction rows per stage
constexpr int kStage = kRows * kPix;       // floats per stage (18 KB)
constexpr unsigned kOOB = 0x80000000u;

struct S2Args {
  const float *x, *uf, *bias;
  float *y;
  int N, C, K, H, W, OH, OW, relu, nsp;
  long long pixels;
};

// (register budget of 2 waves per SIMD although the kernel needs ~88 VGPRs = 5 waves: with a 4-wave budget hipcc reuses load
// destinations so tightly that its s_waitcnt insertion falls back to vmcnt(0) at the loop head)
__global__ __launch_bounds__(512, 2) void conv3x3s2_kernel(const S2Args a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int kb16 = wv & 3, half = wv >> 2;
  const int KB = a.K / kKw;
  const int grp = blockIdx.x / (8 * KB), rem = blockIdx.x - grp * 8 * KB;   // channel blocks of one pixel group: same XCD
  const int kb = rem >> 3, sp = grp * 8 + (rem & 7);
  if (sp >= a.nsp) return;
  const long long p0 = (long long)sp * kPix;
  const int per_img = a.OH * a.OW;
  const int n0 = (int)(p0 / per_img);
  const long long plane = (long long)a.H * a.W, img = plane * a.C;
  const int nch = a.C / kCc;

  // ---- load role: lane = output pixel, wave = channel of the stage.  Offsets against a base 4 bytes in front of the window
  // (+4 bias: a negative buffer offset zeroes the whole load); stage 0 of the prologue uses in-row loads (see winograd_conv.hip).
  unsigned rowq[3], rowq0[3];
  bool left;
  {
    const long long p = p0 + lane;
    const bool pv = p < a.pixels;
    const int pi = pv ? (int)(p - (long long)n0 * per_img) : 0;
    const int nn = pi / per_img, r = pi - nn * per_img;
    const int oy = r / a.OW, ox = r - oy * a.OW;
    left = ox == 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int yy = 2 * oy - 1 + i;
      const bool ok = pv && yy >= 0 && yy < a.H;
      const long long row = (long long)nn * img + (long long)yy * a.W;
      rowq0[i] = ok ? (unsigned)((row + (left ? 0 : 2 * ox - 1)) * 4) : kOOB;
      rowq[i] = ok ? (unsigned)((row + 2 * ox - 1) * 4 + 4) : kOOB;
    }
  }
  const unsigned lm = left ? ~0u : 0u;
  const int n_here = min(2, a.N - n0);
  const __amdgpu_buffer_rsrc_t rx0 = dvis_make_rsrc_uniform(a.x + (long long)n0 * img, (unsigned)(n_here * img * 4));
  const __amdgpu_buffer_rsrc_t rx = dvis_make_rsrc_uniform(reinterpret_cast<const char *>(a.x + (long long)n0 * img) - 4,
                                                           (unsigned)(n_here * img * 4 + 4));
  const __amdgpu_buffer_rsrc_t ru = dvis_make_rsrc_uniform(a.uf, (unsigned)(12ll * a.K * a.C * 4));
  const unsigned plane_bytes = (unsigned)(plane * 4);
  const unsigned u_lane = (unsigned)lane * 16u;
  const unsigned u_blk = (unsigned)((kb * 4 + kb16) * nch);

  auto load_d = [&](int ch, dvis_f4 (&d)[3]) {
    const unsigned so = (unsigned)(ch * kCc + wv) * plane_bytes;
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, rowq[i], so, 0));
  };
  // u[q] = 4 of the 12 floats of the lane: U[k-step s = 4 q + e][...] for s < 9 (the last three are padding)
  auto load_u = [&](int ch, dvis_f4 (&u)[3]) {
    const unsigned so = ((u_blk + (unsigned)ch) * 2u + (unsigned)half) * 3072u;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      u[q] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + 1024u * q, so, 0));
  };
  // row k = 8 tap + c of the stage, tap = 3 i + jj; 64 lanes write 64 consecutive dwords (conflict-free)
  auto store_taps = [&](const dvis_f4 (&d)[3], float *stage, bool in_row) {
    float *vw = stage + wv * kPix + lane;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float e0, e1, e2;
      if (in_row) {   // (prologue, stage 0: the leftmost pixel loaded columns 0..3 instead of -1..2)
        e0 = left ? 0.f : d[i][0], e1 = left ? d[i][0] : d[i][1], e2 = left ? d[i][1] : d[i][2];
      } else {
        e0 = __uint_as_float(__float_as_uint(d[i][0]) & ~lm), e1 = d[i][1], e2 = d[i][2];
      }
      vw[(3 * i) * kCc * kPix] = e0;
      vw[(3 * i + 1) * kCc * kPix] = e1;
      vw[(3 * i + 2) * kCc * kPix] = e2;
    }
  };

  dvis_f4 acc[4];
#pragma unroll
  for (int tb = 0; tb < 4; ++tb) acc[tb] = dvis_f4{0.f, 0.f, 0.f, 0.f};
  auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
  // one stage: the wave's 9 k-steps on `cur`; the next stage's taps go to `nxt` after the 3rd k-step, the patch of the stage
  // after next is requested right behind them
  auto stage = [&](const float *cur, const dvis_f4 (&u)[3], float *nxt, dvis_f4 (&d)[3], int ch_load) {
    const dvis_f4 *vr = reinterpret_cast<const dvis_f4 *>(cur + (half * 36 + g) * kPix + 4 * j);
    dvis_f4 b[2];
    b[0] = vr[0];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      if (s + 1 < 9) b[(s + 1) & 1] = vr[((s + 1) * 4 * kPix) / 4];
      const float av = u[s >> 2][s & 3];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) acc[tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[s & 1][tb], acc[tb], 0, 0, 0);
      if (s == 2) {
        store_taps(d, nxt, false);
        load_d(ch_load, d);
      }
      fence();
    }
  };

  float *s0 = lds, *s1 = lds + kStage;
  dvis_f4 d[3] = {}, d2[3] = {}, ua[3] = {}, ub[3] = {};
  load_u(0, ua);
  {
    const unsigned so = (unsigned)wv * plane_bytes;
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rx0, rowq0[i], so, 0));
  }
  store_taps(d, s0, true);
  load_d(1, d);
  fence();
  load_d(min(2, nch - 1), d2);
  fence();
  load_u(1, ub);
  fence();
#pragma unroll 1
  for (int ch = 0; ch < nch; ch += 2) {   // nch even; straight-line pairs with clamped stage indices (see winograd_conv.hip)
    const int c2 = min(ch + 2, nch - 1), c3 = min(ch + 3, nch - 1), c4 = min(ch + 4, nch - 1);
    __syncthreads();
    stage(s0, ua, s1, d, c3);
    load_u(c2, ua);
    fence();
    __syncthreads();
    stage(s1, ub, s0, d2, c4);
    load_u(c3, ub);
    fence();
  }

  // ---- the halves' partial sums: half h stores accumulator tiles tb = 2 h, 2 h + 1, the other two go to the partner through LDS
  __syncthreads();
  {
    float *ex = lds + (((1 - half) * 4 + kb16) * 2) * 4 * 64 + lane;   // [dst half][kb16][t2][r][lane]
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 4; ++r) ex[(t2 * 4 + r) * 64] = half ? acc[t2][r] : acc[2 + t2][r];
  }
  __syncthreads();
  const float *ex = lds + ((half * 4 + kb16) * 2) * 4 * 64 + lane;
  const int k0 = kb * kKw + kb16 * 16 + 4 * g;
  const long long pa = p0 + 4 * j + 2 * half;
  float o[2][4];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mine = half ? acc[2 + t2][r] : acc[t2][r], other = ex[(t2 * 4 + r) * 64];
      const float v = (half ? other + mine : mine + other) + (a.bias ? a.bias[k0 + r] : 0.f);
      o[t2][r] = a.relu ? fmaxf(v, 0.f) : v;
    }
  if ((a.OW & 1) == 0) {   // the two pixels lie in one output row, 8-byte aligned
    if (pa < a.pixels) {
      const int n = (int)(pa / per_img), rr = (int)(pa - (long long)n * per_img);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float2 *>(a.y + ((long long)n * a.K + k0 + r) * per_img + rr) = make_float2(o[0][r], o[1][r]);
    }
  } else {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const long long p = pa + t2;
      if (p >= a.pixels) continue;
      const int n = (int)(p / per_img), rr = (int)(p - (long long)n * per_img);
#pragma unroll
      for (int r = 0; r < 4; ++r) a.y[((long long)n * a.K + k0 + r) * per_img + rr] = o[t2][r];
    }
  }
}

// uf[kb16][stage][half][q (3)][lane = 16 g + i][e (4)] = W[k = 16 kb16 + i][c][tap] with contraction row 8 tap + c = 36 half + 4 s + g,
// s = 4 q + e (s < 9; the rest is zero padding that is never multiplied)
__global__ void conv3x3s2_pack_kernel(const float *__restrict__ w, float *__restrict__ uf, int K, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nch = C / kCc;
  const long long total = (long long)(K / 16) * nch * 2 * 3 * 64 * 4;
  if (idx >= total) return;
  int t = idx;
  const int e = t & 3; t >>= 2;
  const int ln = t & 63; t >>= 6;
  const int q = t % 3; t /= 3;
  const int half = t & 1; t >>= 1;
  const int st = t % nch, kb16 = t / nch;
  const int s = 4 * q + e, g = ln >> 4, i16 = ln & 15;
  float v = 0.f;
  if (s < 9) {
    const int row = 36 * half + 4 * s + g, tap = row >> 3, c = st * kCc + (row & 7);
    v = w[((long long)(kb16 * 16 + i16) * C + c) * 9 + tap];
  }
  uf[idx] = v;
}

}  // namespace

DVIS_EXPORT int dvis_conv3x3s2_supported(int C, int K, int H, int W) {
  if (C <= 0 || K <= 0 || H < 2 || W < 4 || (W & 1) || C % 16 != 0 || K % kKw != 0) return 0;
  const long long per_img = (long long)((H + 1) / 2) * (W / 2);
  if (per_img < kPix) return 0;
  if (2ll * C * H * W * 4 + 4 >= (1ll << 31) || 12ll * K * C * 4 >= (1ll << 31)) return 0;
  return 1;
}

DVIS_EXPORT int dvis_conv3x3s2_pack(const float *w, float *uf, int K, int C, void *stream) {
  DVIS_REQUIRE(w && uf && K > 0 && C > 0 && C % 16 == 0 && K % kKw == 0, "conv3x3s2_pack: K %% 64 == 0 and C %% 16 == 0");
  const long long n = 12ll * K * C;
  hipLaunchKernelGGL(conv3x3s2_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, uf, K, C);
  return dvis_check_launch("dvis_conv3x3s2_pack");
}

DVIS_EXPORT int dvis_conv3x3s2(const float *x, const float *uf, const float *bias, float *y, int N, int C, int K, int H, int W,
                               int relu, void *stream) {
  DVIS_REQUIRE(N >= 0, "conv3x3s2: bad batch");
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(x && uf && y, "conv3x3s2: null pointer");
  DVIS_REQUIRE(dvis_conv3x3s2_supported(C, K, H, W), "conv3x3s2: unsupported shape C=%d K=%d H=%d W=%d (dvis_conv3x3s2_supported)", C,
               K, H, W);
  DVIS_REQUIRE((((uintptr_t)x | (uintptr_t)uf | (uintptr_t)y) & 15) == 0, "conv3x3s2: 16-byte aligned tensors");
  S2Args a;
  a.x = x, a.uf = uf, a.bias = bias, a.y = y;
  a.N = N, a.C = C, a.K = K, a.H = H, a.W = W, a.relu = relu;
  a.OH = (H + 1) / 2, a.OW = W / 2;
  a.pixels = (long long)N * a.OH * a.OW;
  const long long nsp = (a.pixels + kPix - 1) / kPix;
  DVIS_REQUIRE(nsp * (K / kKw) + 8 * (K / kKw) < (1ll << 31), "conv3x3s2: grid too large");
  a.nsp = (int)nsp;
  const unsigned grid = (unsigned)(((nsp + 7) / 8) * 8 * (K / kKw));
  hipLaunchKernelGGL(conv3x3s2_kernel, dim3(grid), dim3(512), 2 * kStage * sizeof(float), (hipStream_t)stream, a);
  return dvis_check_launch("dvis_conv3x3s2");
}
