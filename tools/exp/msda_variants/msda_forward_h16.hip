// MSDeformAttn forward gathering from an fp16 COPY of `value` — gfx950.  EXPERIMENT, off by default.
//
// The tiled fp32 kernel (msda_forward.hip) is bound by the texture-address / L1 pipe: one 128-byte line = two 64-byte
// accesses per (sample, head, corner).  With `value` stored in fp16 a corner is ONE 64-byte access (4 lanes x 16 B per
// (query, head) pair, 16 pairs per wave-instruction): half the accesses, half the gathered bytes.  Everything else —
// offsets, logits, softmax, sampling locations, bilinear weights, accumulation — stays fp32; only the sampled feature
// values are rounded to fp16 (relative error <= 2^-11 per value, so |out error| <= 4.9e-4 * max|value|: inside
// BASELINE.json's 1e-3, outside the 1e-5 the op-level parity tests hold).  It therefore cannot replace the fp32 path
// without a decision on the op tolerance; it exists to measure what the byte halving buys.
// Measured on MI355X (30 frames, 720p, tools/msda_h16_probe.py): 30.1 us per frame-layer vs 37.9 us for the fp32 kernel on
// the same inputs, plus 5.0 us to make the fp16 copy — a wash: half the corner-load instructions buys 20 %, so the
// gather is not purely load-count bound (tap reads from LDS, conversions and FMAs stay).
// Fused interface only, D = 32, (L, P) = (3, 4).
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "dvis_common.h"
#include "msda_tap.h"

namespace {

using dvis_msda::kOOB;
using dvis_msda::make_tap;
using dvis_msda::Tap;

template <int L, int P, int B>
__global__ __launch_bounds__(256) void msda_fwd_tile_h16(
    const __half *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ off, int64_t off_stride, const float *__restrict__ logit, int64_t logit_stride,
    const float *__restrict__ refp, int nref, int S, int M, int Lq, float *__restrict__ out) {
  constexpr int D = 32, LP = L * P, QB = 64, G = 4, LOCV = LP / 2, WV = LP / 4;
  __shared__ float4 s_loc[QB * LOCV];
  __shared__ float4 s_w[QB * WV];
  __shared__ uint4 s_tap_o[QB * LP];
  __shared__ float4 s_tap_c[QB * LP];

  const int tid = threadIdx.x;
  const int m = blockIdx.x, n = blockIdx.z;
  const int MD = M * D;
  const int q0 = blockIdx.y * QB;
  int Hs[L], Ws[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
  }
  auto slot_query = [&](int ql) -> int { return q0 + ql < Lq ? q0 + ql : -1; };
  {
    const size_t row0 = (size_t)n * Lq;
    const unsigned lrow = (unsigned)((size_t)off_stride * sizeof(float));
    const unsigned wrow = (unsigned)((size_t)logit_stride * sizeof(float));
    const __amdgpu_buffer_rsrc_t lrs =
        dvis_make_rsrc_uniform(off + row0 * off_stride + (size_t)m * (LP * 2), (unsigned)(Lq - 1) * lrow + LP * 2 * 4);
    const __amdgpu_buffer_rsrc_t wrs =
        dvis_make_rsrc_uniform(logit + row0 * logit_stride + (size_t)m * LP, (unsigned)(Lq - 1) * wrow + LP * 4);
    for (int i = tid; i < QB * LOCV; i += 256) {
      const int ql = i / LOCV, k = i - ql * LOCV;
      const int q = slot_query(ql);
      const dvis_v4u v = __builtin_amdgcn_raw_buffer_load_b128(lrs, q >= 0 ? (unsigned)q * lrow + (unsigned)k * 16u : kOOB, 0, 0);
      s_loc[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    for (int i = tid; i < QB * WV; i += 256) {
      const int ql = i / WV, k = i - ql * WV;
      const int q = slot_query(ql);
      const dvis_v4u v = __builtin_amdgcn_raw_buffer_load_b128(wrs, q >= 0 ? (unsigned)q * wrow + (unsigned)k * 16u : kOOB, 0, 0);
      s_w[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  }
  __syncthreads();
  float *lf = reinterpret_cast<float *>(s_loc);
  float *wf = reinterpret_cast<float *>(s_w);
  const unsigned pix_bytes = (unsigned)MD * 2u;                    // fp16 pixels
  for (int i = tid; i < QB * LP; i += 256) {
    const int ql = i / LP, s = i - ql * LP;
    const int l = s / P;
    const int q = slot_query(ql);
    int Hl = Hs[0], Wl = Ws[0];
#pragma unroll
    for (int ll = 1; ll < L; ++ll)
      if (l == ll) { Hl = Hs[ll]; Wl = Ws[ll]; }
    float x = lf[ql * LP * 2 + 2 * s], y = lf[ql * LP * 2 + 2 * s + 1];
    if (q >= 0) {
      const size_t rrow = ((size_t)(nref == 1 ? 0 : n) * Lq + q) * L + l;
      const float2 r = *reinterpret_cast<const float2 *>(refp + rrow * 2);
      x = r.x + x / (float)Wl;
      y = r.y + y / (float)Hl;
    }
    const Tap t = make_tap(x, y, Hl, Wl, q >= 0, pix_bytes, 0u);
    s_tap_o[i] = make_uint4(t.o[0], t.o[1], t.o[2], t.o[3]);
    s_tap_c[i] = make_float4(t.c[0], t.c[1], t.c[2], t.c[3]);
  }
  if (tid < QB) {
    float *row = wf + tid * LP;
    float mx = row[0];
#pragma unroll
    for (int s = 1; s < LP; ++s) mx = fmaxf(mx, row[s]);
    float e[LP], sum = 0.f;
#pragma unroll
    for (int s = 0; s < LP; ++s) { e[s] = expf(row[s] - mx); sum += e[s]; }
#pragma unroll
    for (int s = 0; s < LP; ++s) row[s] = e[s] / sum;
  }
  __syncthreads();

  __amdgpu_buffer_rsrc_t rs[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const __half *base = value + (((size_t)n * S + (size_t)level_start[l]) * M + m) * D;
    rs[l] = dvis_make_rsrc_uniform(base, (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * MD + D) * sizeof(__half)));
  }
  const int lane = tid & 63, wv = tid >> 6;
  const int g = lane / G, j = lane - g * G;
  const unsigned lane_bytes = (unsigned)j * 16u;                    // 8 halves per lane
  const int ql = wv * 16 + g;
  const int q = slot_query(ql);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) {
#pragma unroll 1
    for (int pb = 0; pb < P / B; ++pb) {
      const int s0 = ql * LP + l * P + pb * B;
      dvis_v4u r[4 * B];
      float4 c[B];
      float aw[B];
#pragma unroll
      for (int i = 0; i < B; ++i) {
        const uint4 o = s_tap_o[s0 + i];
        r[4 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o.x + lane_bytes, 0, 0);
        r[4 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o.y + lane_bytes, 0, 0);
        r[4 * i + 2] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o.z + lane_bytes, 0, 0);
        r[4 * i + 3] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o.w + lane_bytes, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < B; ++i) {
        c[i] = s_tap_c[s0 + i];
        aw[i] = wf[s0 + i];
      }
#pragma unroll
      for (int i = 0; i < B; ++i) {
        const float cw[4] = {c[i].x, c[i].y, c[i].z, c[i].w};
        float sum[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) sum[k] = 0.f;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const unsigned u[4] = {r[4 * i + cc].x, r[4 * i + cc].y, r[4 * i + cc].z, r[4 * i + cc].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float2 f = __half22float2(*reinterpret_cast<const __half2 *>(&u[k]));
            sum[2 * k] += cw[cc] * f.x;       // corner order w1 v1 + w2 v2 + w3 v3 + w4 v4, as the fp32 kernel
            sum[2 * k + 1] += cw[cc] * f.y;
          }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += sum[k] * aw[i];
      }
    }
  }
  if (q >= 0) {
    float *dst = out + (((size_t)n * Lq + q) * M + m) * D + 8 * j;
    *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

}  // namespace

DVIS_EXPORT int dvis_msda_fused_forward_h16(const void *value_f16, const int64_t *shapes, const int64_t *level_start,
                                            const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                            const float *logits, int64_t logit_stride, int N, int S, int M, int D, int L,
                                            int Lq, int P, float *out, void *stream) {
  DVIS_REQUIRE(N >= 0 && S > 0 && M > 0 && Lq >= 0, "msda_fused_forward_h16: bad sizes");
  if (N == 0 || Lq == 0) return DVIS_OK;
  DVIS_REQUIRE(value_f16 && shapes && level_start && ref && offsets && logits && out, "msda_fused_forward_h16: null pointer");
  DVIS_REQUIRE(D == 32 && L == 3 && P == 4, "msda_fused_forward_h16: only D=32, L=3, P=4 (got %d, %d, %d)", D, L, P);
  DVIS_REQUIRE(Nref == 1 || Nref == N, "msda_fused_forward_h16: Nref must be 1 or N");
  DVIS_REQUIRE(off_stride % 4 == 0 && logit_stride % 4 == 0 && off_stride >= (int64_t)M * L * P * 2 &&
                   logit_stride >= (int64_t)M * L * P &&
                   ((((uintptr_t)offsets) | ((uintptr_t)logits) | ((uintptr_t)value_f16) | ((uintptr_t)out)) & 15) == 0,
               "msda_fused_forward_h16: 16-byte alignment / row strides");
  DVIS_REQUIRE((size_t)S * M * D * 2 < 0x7fffffffu && (size_t)Lq * (size_t)off_stride * 4 < 0x7fffffffu, "msda_fused_forward_h16: slice >= 2 GiB");
  const int nchunks = (Lq + 63) / 64;
  DVIS_REQUIRE(nchunks <= 65535 && N <= 65535, "msda_fused_forward_h16: grid too large");
  hipLaunchKernelGGL((msda_fwd_tile_h16<3, 4, 2>), dim3(M, nchunks, N), dim3(256), 0, (hipStream_t)stream,
                     (const __half *)value_f16, shapes, level_start, offsets, off_stride, logits, logit_stride, ref, Nref, S,
                     M, Lq, out);
  return dvis_check_launch("msda_fwd_tile_h16");
}
