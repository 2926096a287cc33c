// 1x1 / stride 1 convolution + bias + residual + ReLU, NCHW fp32, for the COMPUTE-bound layers of the R50 (many input channels at
// the small maps: conv1 512 -> 128 / 1024 -> 256 / 2048 -> 512 and conv3 256 -> 1024 / 512 -> 2048 of the res3 - res5 bottlenecks),
// round 3.  csrc/conv1x1.hip keeps the whole weight matrix on chip and serves the memory-bound layers (few input channels, large
// maps); these ones went through the library's batched GEMM followed by the separate bias / shortcut / ReLU pass (`bias_act`).
// Same skeleton as csrc/conv3x3s2.hip with ONE tap:  out[k][pixel] = sum_c W[k][c] x[c][pixel].
//   * workgroup = 64 consecutive pixels (in (n, y, x) order) x 64 output channels, 8 waves; a stage = 64 input channels = 64
//     contraction rows of 64 pixels, double-buffered in LDS (16 KB per stage).
//   * wave w loads channels 8 w .. 8 w + 7 of the stage: 8 dword loads per lane (lane = pixel: 256 contiguous bytes per wave
//     instruction), no arithmetic; wave w = (16 output channels, half of the 16 k-steps of a stage); weights packed once into the
//     MFMA A layout (dvis_conv1x1_mfma_pack), 8 floats per lane and stage.
//   * epilogue: the halves meet through LDS, + bias[k] + residual, ReLU, two adjacent pixels per lane.
// Fixed accumulation order: bit-reproducible.
#include "dvis_common.h"

namespace {

constexpr int kPix = 64, kKw = 64, kCc = 64;
constexpr int kStage = kCc * kPix;   // floats per stage (16 KB)
constexpr unsigned kOOB = 0x80000000u;

struct C1Args {
  const float *x, *uf, *bias, *res;
  float *y;
  int N, C, K, relu, nsp;
  int stride, W_in, OW;       // stride 2 (the shortcut of the first res3 / res4 / res5 bottleneck): output pixel (oy, ox) reads (2 oy, 2 ox)
  long long HW, pixels;       // OUTPUT pixels per image / in total
  long long HW_in;            // input pixels per image
};

template <int KW>   // 16-channel blocks per wave: a workgroup covers 64 * KW output channels
__global__ __launch_bounds__(512, 2) void conv1x1_mfma_kernel(const C1Args a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int kb16 = wv & 3, half = wv >> 2;
  const int KB = a.K / (kKw * KW);
  const int grp = blockIdx.x / (8 * KB), rem = blockIdx.x - grp * 8 * KB;   // channel blocks of one pixel group: same XCD
  const int kb = rem >> 3, sp = grp * 8 + (rem & 7);
  if (sp >= a.nsp) return;
  const long long p0 = (long long)sp * kPix;
  const int n0 = (int)(p0 / a.HW);
  const long long img = a.HW_in * a.C;
  const int nch = a.C / kCc;

  // load role: lane = pixel; the workgroup's pixels lie in images n0 and n0 + 1 (HW >= 64)
  unsigned off;
  {
    const long long p = p0 + lane;
    const long long pi = p - (long long)n0 * a.HW;
    const int nn = (int)(pi / a.HW);
    long long in_pix = pi - (long long)nn * a.HW;
    if (a.stride == 2) {   // (wave-uniform)
      const int oy = (int)(in_pix / a.OW), ox = (int)(in_pix - (long long)oy * a.OW);
      in_pix = 2ll * oy * a.W_in + 2 * ox;
    }
    off = p < a.pixels ? (unsigned)(((long long)nn * img + in_pix) * 4) : kOOB;
  }
  const int n_here = min(2, a.N - n0);
  const __amdgpu_buffer_rsrc_t rx = dvis_make_rsrc_uniform(a.x + (long long)n0 * img, (unsigned)(n_here * img * 4));
  const __amdgpu_buffer_rsrc_t ru = dvis_make_rsrc_uniform(a.uf, (unsigned)((long long)a.K * a.C * 4));
  const unsigned plane_bytes = (unsigned)(a.HW_in * 4);
  const unsigned u_lane = (unsigned)lane * 16u;
  const unsigned u_blk = (unsigned)((kb * 4 * KW + kb16) * nch);   // (+ 4 nch per further block of the wave)

  auto load_d = [&](int ch, float (&d)[8]) {
    const unsigned so = (unsigned)(ch * kCc + wv * 8) * plane_bytes;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      d[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, so + (unsigned)i * plane_bytes, 0));
  };
  auto load_u = [&](int ch, dvis_f4 (&u)[KW][2]) {
#pragma unroll
    for (int w = 0; w < KW; ++w) {
      const unsigned so = ((u_blk + (unsigned)(4 * w * nch) + (unsigned)ch) * 2u + (unsigned)half) * 2048u;
#pragma unroll
      for (int q = 0; q < 2; ++q)
        u[w][q] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + 1024u * q, so, 0));
    }
  };
  auto store_rows = [&](const float (&d)[8], float *stage) {
    float *vw = stage + (wv * 8) * kPix + lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) vw[i * kPix] = d[i];
  };

  dvis_f4 acc[KW][4];
#pragma unroll
  for (int w = 0; w < KW; ++w)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[w][tb] = dvis_f4{0.f, 0.f, 0.f, 0.f};
  auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
  auto stage = [&](const float *cur, const dvis_f4 (&u)[KW][2], float *nxt, float (&d)[8], int ch_load) {
    const dvis_f4 *vr = reinterpret_cast<const dvis_f4 *>(cur + (half * 32 + g) * kPix + 4 * j);
    dvis_f4 b[2];
    b[0] = vr[0];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 1 < 8) b[(s + 1) & 1] = vr[(s + 1) * kPix];
#pragma unroll
      for (int w = 0; w < KW; ++w) {
        const float av = u[w][s >> 2][s & 3];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[w][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[s & 1][tb], acc[w][tb], 0, 0, 0);
      }
      if (s == 2) {
        store_rows(d, nxt);
        load_d(ch_load, d);
      }
      fence();
    }
  };

  float *s0 = lds, *s1 = lds + kStage;
  float d[8], d2[8];
  dvis_f4 ua[KW][2], ub[KW][2];
  load_u(0, ua);
  load_d(0, d);
  store_rows(d, s0);
  load_d(min(1, nch - 1), d);
  fence();
  load_d(min(2, nch - 1), d2);
  fence();
  load_u(min(1, nch - 1), ub);
  fence();
#pragma unroll 1
  for (int ch = 0; ch < nch; ch += 2) {   // nch even (C % 128 == 0); straight-line pairs with clamped stage indices (see winograd_conv.hip)
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

  // ---- the halves' partial sums through LDS; half h stores accumulator tiles 2 h, 2 h + 1 (pixels 4 j + tb)
  __syncthreads();
  {
    float *ex = lds + (((1 - half) * 4 * KW + kb16 * KW) * 2) * 4 * 64 + lane;   // [dst half][kb16][w][t2][r][lane]
#pragma unroll
    for (int w = 0; w < KW; ++w)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) ex[((w * 2 + t2) * 4 + r) * 64] = half ? acc[w][t2][r] : acc[w][2 + t2][r];
  }
  __syncthreads();
  const float *ex = lds + ((half * 4 * KW + kb16 * KW) * 2) * 4 * 64 + lane;
  const long long pa = p0 + 4 * j + 2 * half;
  if (pa >= a.pixels) return;   // (pixels even: HW even)
  const int n = (int)(pa / a.HW);
  const long long rr = pa - (long long)n * a.HW;
#pragma unroll
  for (int w = 0; w < KW; ++w) {
    const int k0 = kb * kKw * KW + (4 * w + kb16) * 16 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long o_idx = ((long long)n * a.K + k0 + r) * a.HW + rr;
      float2 rv = make_float2(0.f, 0.f);
      if (a.res) rv = *reinterpret_cast<const float2 *>(a.res + o_idx);
      const float bv = a.bias ? a.bias[k0 + r] : 0.f;
      float o[2];
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const float mine = half ? acc[w][2 + t2][r] : acc[w][t2][r], other = ex[((w * 2 + t2) * 4 + r) * 64];
        const float v = (half ? other + mine : mine + other) + bv + (t2 ? rv.y : rv.x);
        o[t2] = a.relu ? fmaxf(v, 0.f) : v;
      }
      *reinterpret_cast<float2 *>(a.y + o_idx) = make_float2(o[0], o[1]);
    }
  }
}

// uf[kb16][stage][half][q (2)][lane = 16 g + i][e (4)] = W[k = 16 kb16 + i][c = 64 stage + 32 half + 4 (4 q + e) + g]
__global__ void conv1x1_mfma_pack_kernel(const float *__restrict__ w, float *__restrict__ uf, int K, int C) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)K * C) return;
  long long t = idx;
  const int e = t & 3; t >>= 2;
  const int ln = t & 63; t >>= 6;
  const int q = t & 1; t >>= 1;
  const int half = t & 1; t >>= 1;
  const int nch = C / kCc;
  const int st = (int)(t % nch), kb16 = (int)(t / nch);
  const int s = 4 * q + e, g = ln >> 4, i16 = ln & 15;
  uf[idx] = w[(long long)(kb16 * 16 + i16) * C + st * kCc + 32 * half + 4 * s + g];
}

}  // namespace

DVIS_EXPORT int dvis_conv1x1_mfma_supported(int C, int K, int64_t HW) {
  if (C <= 0 || K <= 0 || HW < kPix || (HW & 1) || C % (2 * kCc) != 0 || K % kKw != 0) return 0;
  if (2ll * C * HW * 4 >= (1ll << 31) || (long long)K * C * 4 >= (1ll << 31)) return 0;
  return 1;
}

DVIS_EXPORT int dvis_conv1x1_mfma_pack(const float *w, float *uf, int K, int C, void *stream) {
  DVIS_REQUIRE(w && uf && K > 0 && C > 0 && C % (2 * kCc) == 0 && K % kKw == 0, "conv1x1_mfma_pack: K %% 64 == 0 and C %% 128 == 0");
  const long long n = (long long)K * C;
  hipLaunchKernelGGL(conv1x1_mfma_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, uf, K, C);
  return dvis_check_launch("dvis_conv1x1_mfma_pack");
}

namespace {
int launch_c1(C1Args a, void *stream);
}

DVIS_EXPORT int dvis_conv1x1_mfma(const float *x, const float *uf, const float *bias, const float *res, float *y, int N, int C, int K,
                                  int64_t HW, int relu, void *stream) {
  DVIS_REQUIRE(N >= 0, "conv1x1_mfma: bad batch");
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(x && uf && y, "conv1x1_mfma: null pointer");
  DVIS_REQUIRE(dvis_conv1x1_mfma_supported(C, K, HW), "conv1x1_mfma: unsupported shape C=%d K=%d HW=%lld (dvis_conv1x1_mfma_supported)", C, K,
               (long long)HW);
  DVIS_REQUIRE((((uintptr_t)x | (uintptr_t)uf | (uintptr_t)y | (uintptr_t)res) & 15) == 0, "conv1x1_mfma: 16-byte aligned tensors");
  C1Args a;
  a.x = x, a.uf = uf, a.bias = bias, a.res = res, a.y = y;
  a.N = N, a.C = C, a.K = K, a.relu = relu, a.HW = HW, a.HW_in = HW, a.stride = 1, a.W_in = 0, a.OW = 0;
  return launch_c1(a, stream);
}

// stride 2: y (N, K, ceil(H/2), ceil(W/2)) = relu?(w (K, C) x[:, :, ::2, ::2] + bias[k] + res)
DVIS_EXPORT int dvis_conv1x1s2_mfma(const float *x, const float *uf, const float *bias, const float *res, float *y, int N, int C, int K,
                                    int H, int W, int relu, void *stream) {
  DVIS_REQUIRE(N >= 0 && H > 0 && W > 0, "conv1x1s2_mfma: bad sizes");
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(x && uf && y, "conv1x1s2_mfma: null pointer");
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  DVIS_REQUIRE(dvis_conv1x1_mfma_supported(C, K, (int64_t)OH * OW) && 2ll * C * H * W * 4 < (1ll << 31),
               "conv1x1s2_mfma: unsupported shape C=%d K=%d H=%d W=%d (dvis_conv1x1_mfma_supported on the output size)", C, K, H, W);
  DVIS_REQUIRE((((uintptr_t)x | (uintptr_t)uf | (uintptr_t)y | (uintptr_t)res) & 15) == 0, "conv1x1s2_mfma: 16-byte aligned tensors");
  C1Args a;
  a.x = x, a.uf = uf, a.bias = bias, a.res = res, a.y = y;
  a.N = N, a.C = C, a.K = K, a.relu = relu, a.HW = (long long)OH * OW, a.HW_in = (long long)H * W, a.stride = 2, a.W_in = W, a.OW = OW;
  return launch_c1(a, stream);
}

namespace {
int launch_c1(C1Args a, void *stream) {
  const int K = a.K;
  const long long HW = a.HW;
  const int N = a.N;
  a.pixels = (long long)N * HW;
  const long long nsp = (a.pixels + kPix - 1) / kPix;
  DVIS_REQUIRE(nsp * (K / kKw) + 8 * (K / kKw) < (1ll << 31), "conv1x1_mfma: grid too large");
  a.nsp = (int)nsp;
  // 128 output channels per workgroup for the long contractions: an input stage is fetched and staged once per 128 instead of 64
  // channels (measured: 512 -> 2048 523 -> 491 us, 2048 -> 512 436 -> 425; at C = 256 the two-pair loop is too short for it: 591 -> 614)
  if (K % (2 * kKw) == 0 && a.C >= 512) {
    const unsigned grid = (unsigned)(((nsp + 7) / 8) * 8 * (K / (2 * kKw)));
    hipLaunchKernelGGL(conv1x1_mfma_kernel<2>, dim3(grid), dim3(512), 2 * kStage * sizeof(float), (hipStream_t)stream, a);
  } else {
    const unsigned grid = (unsigned)(((nsp + 7) / 8) * 8 * (K / kKw));
    hipLaunchKernelGGL(conv1x1_mfma_kernel<1>, dim3(grid), dim3(512), 2 * kStage * sizeof(float), (hipStream_t)stream, a);
  }
  return dvis_check_launch("dvis_conv1x1_mfma");
}
}  // namespace
