// MSDA tile kernel with timing-experiment knobs (NOT product code; built into its own .so by build.sh):
//   EXP bit 0: skip the corner gathers of all levels but the last (what an LDS-served level would cost at best)
//   EXP bit 1: skip every gather (set-up + stores only)
//   EXP bit 2: value laid out head-major (N, M, S, D) instead of (N, S, M, D)
//   EXP bit 6: the corner lines of the COARSEST level (l = 0) go TA -> LDS (buffer_load ... lds, no VGPR return) and are read
//                back with ds_read_b128: is the wall the VGPR return path or the TCP tag / address path?  (round 4)
//   EXP bit 3 / 4 / 5: left corners of every second query / of every query / three corners of four get an out-of-range
//                offset (timing only: what register re-use of corners between x-neighbouring queries would save, and
//                whether an out-of-range lane costs a request at all)
#include <stdlib.h>
#include "../../../dvis_plus_amd/csrc/dvis_common.h"
#include "../../../dvis_plus_amd/csrc/msda_tap.h"
namespace {
using dvis_msda::kOOB;
using dvis_msda::make_tap;
using dvis_msda::Tap;
// QB queries per workgroup; WPS = register budget in waves/SIMD; B = samples per batch of corner loads.
template <int D, int L, int P, bool FUSED, int WPS, int B, int QB, int EXP>
__global__ __launch_bounds__(256, WPS) void msda_fwd_tile_f32(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc_or_off, int64_t off_stride, const float *__restrict__ w_or_logit,
    int64_t logit_stride, const float *__restrict__ refp, int nref, int S, int M, int Lq, float *__restrict__ out,
    const float *__restrict__ pos_off, const float *__restrict__ pos_logit, int64_t pos_stride) {
  constexpr int LP = L * P;
  constexpr int G = D / 4;          // lanes per (query, head) pair
  constexpr int GPW = 64 / G;       // pairs per wave-instruction
  constexpr int LOCV = LP / 2;      // float4s of (x, y) per pair
  constexpr int WV = LP / 4;        // float4s of weights per pair
  constexpr int ITERS = QB / (4 * GPW);
  static_assert(LP % 4 == 0 && D % 4 == 0 && 64 % G == 0 && P % B == 0 && QB % (4 * GPW) == 0, "tile shape");

  // Bilinear set-up of every (query, sample) of the block, computed ONCE by one thread.  The D/4 lanes of a pair used
  // to redo the same ~50 VALU instructions per sample each: PMC showed 4.1e8 VALU instructions per 30-frame launch =
  // 60 % VALU utilisation, contending with the L1 path for issue slots.
  __shared__ uint4 s_tap_o[QB * LP];    // 4 corner byte offsets (kOOB = outside the map / sample not counted)
  __shared__ float4 s_tap_c[QB * LP];   // 4 corner weights
  __shared__ float s_aw[QB * LP];       // attention weights
  __shared__ dvis_v4u s_dma[(EXP & 64) ? 4 : 1][(EXP & 64) ? 4 * B : 1][64];   // EXP bit 6: one 1 KB landing slot per load

  // grid = (M, ceil(Lq/QB), N): x is the fastest dispatch dimension, so linear id % 8 == m % 8 -> head m on XCD m % 8.
  // blockIdx.* are SGPRs: everything derived from them (bases, descriptors) is wave-uniform.
  const int tid = threadIdx.x;
  const int m = blockIdx.x;
  const int n = blockIdx.z;
  const int MD = M * D;
  const int q0 = blockIdx.y * QB;

  int Hs[L], Ws[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
  }
  // local slot -> global query index, or -1 when the slot is past the end
  auto slot_query = [&](int ql) -> int { return q0 + ql < Lq ? q0 + ql : -1; };

  // ---- set-up: thread (query tid / P, point tid % P) reads ITS parameters of all L levels straight into registers —
  // raw offsets, reference points and the pair's L*P logits (fused) or locations and weights — in ONE global round trip,
  // then softmax, loc = ref + off / (W_l, H_l) and the taps, and ONE barrier.  (The first form staged the rows in LDS,
  // synchronised, loaded the reference points, computed, synchronised again: with the gather switched off that set-up
  // alone took 12.7-15.5 us per 720p frame-layer, and with the loads switched off the kernel still took 23.5 of 35 us.)
  const unsigned pix_bytes = (EXP & 4) ? (unsigned)D * 4u : (unsigned)MD * 4u;
  static_assert(QB * P <= 256, "one set-up thread per (query, point)");
  if (tid < QB * P) {
    const int ql = tid / P, p = tid - ql * P;
    const int q = slot_query(ql);
    const bool active = q >= 0;
    const size_t qq = active ? q : 0;
    float2 xy[L];
    float aw[L];
    if (FUSED) {
      const float *orow = loc_or_off + ((size_t)n * Lq + qq) * off_stride + (size_t)m * (LP * 2);
      const float *lrow = w_or_logit + ((size_t)n * Lq + qq) * logit_stride + (size_t)m * LP;
      float2 ro[L], rr[L];
      float4 rl[LP / 4];
#pragma unroll
      for (int l = 0; l < L; ++l) {
        ro[l] = *reinterpret_cast<const float2 *>(orow + 2 * (l * P + p));
        rr[l] = *reinterpret_cast<const float2 *>(refp + (((size_t)(nref == 1 ? 0 : n) * Lq + qq) * L + l) * 2);
      }
#pragma unroll
      for (int k = 0; k < LP / 4; ++k) rl[k] = *reinterpret_cast<const float4 *>(lrow + 4 * k);
      if (pos_off != nullptr) {     // + projection of the query's position embedding (same for all n)
        const float *prow = pos_off + qq * pos_stride + (size_t)m * (LP * 2);
        const float *plrow = pos_logit + qq * pos_stride + (size_t)m * LP;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float2 pv = *reinterpret_cast<const float2 *>(prow + 2 * (l * P + p));
          ro[l].x += pv.x; ro[l].y += pv.y;
        }
#pragma unroll
        for (int k = 0; k < LP / 4; ++k) {
          const float4 pv = *reinterpret_cast<const float4 *>(plrow + 4 * k);
          rl[k].x += pv.x; rl[k].y += pv.y; rl[k].z += pv.z; rl[k].w += pv.w;
        }
      }
      float lg[LP];
#pragma unroll
      for (int k = 0; k < LP / 4; ++k) { lg[4 * k] = rl[k].x; lg[4 * k + 1] = rl[k].y; lg[4 * k + 2] = rl[k].z; lg[4 * k + 3] = rl[k].w; }
      float mx = lg[0];
#pragma unroll
      for (int s = 1; s < LP; ++s) mx = fmaxf(mx, lg[s]);
      float e[LP], sum = 0.f;
#pragma unroll
      for (int s = 0; s < LP; ++s) { e[s] = expf(lg[s] - mx); sum += e[s]; }
#pragma unroll
      for (int l = 0; l < L; ++l) {
        xy[l].x = rr[l].x + ro[l].x / (float)Ws[l];
        xy[l].y = rr[l].y + ro[l].y / (float)Hs[l];
        // e[] is indexed with a compile-time l and a run-time p: select instead of indexing registers dynamically
        float ev = e[l * P];
#pragma unroll
        for (int pp = 1; pp < P; ++pp) ev = (p == pp) ? e[l * P + pp] : ev;
        aw[l] = ev / sum;
      }
    } else {
      const float *lrow = loc_or_off + (((size_t)n * Lq + qq) * M + m) * (size_t)(LP * 2);
      const float *wrow = w_or_logit + (((size_t)n * Lq + qq) * M + m) * (size_t)LP;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        xy[l] = *reinterpret_cast<const float2 *>(lrow + 2 * (l * P + p));
        aw[l] = wrow[l * P + p];
      }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const Tap t = make_tap(xy[l].x, xy[l].y, Hs[l], Ws[l], active, pix_bytes, 0u);
      const int si = ql * LP + l * P + p;
      s_tap_o[si] = make_uint4(t.o[0], t.o[1], t.o[2], t.o[3]);
      s_tap_c[si] = make_float4(t.c[0], t.c[1], t.c[2], t.c[3]);
      s_aw[si] = aw[l];
    }
  }
  __syncthreads();
  const float *wf = s_aw;

  // ---- per-level buffer descriptors over this (frame, head) slice of `value`
  __amdgpu_buffer_rsrc_t rs[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float *base = (EXP & 4) ? value + (((size_t)n * M + m) * S + (size_t)level_start[l]) * D
                                  : value + (((size_t)n * S + (size_t)level_start[l]) * M + m) * D;
    rs[l] = dvis_make_rsrc_uniform(base, (EXP & 4) ? (unsigned)((size_t)Hs[l] * Ws[l] * D * sizeof(float))
                                                   : (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * MD + D) * sizeof(float)));
  }

  const int lane = tid & 63, wv = tid >> 6;
  const int g = lane / G, j = lane - g * G;
  const unsigned lane_bytes = (unsigned)j * 16u;   // kOOB + lane_bytes is still out of range
  float *const out_frame = out + ((size_t)n * Lq * M + m) * D;   // uniform

  // Latency is hidden by WAVES, not by a deep per-wave pipeline: each wave keeps one batch of B samples
  // (4*B corner loads) in flight, reads that batch's taps from LDS just in time, and stays within the
  // register budget of WPS waves/SIMD.  (A fully unrolled 12-sample body makes hipcc hoist all 48 loads and
  // spill to scratch; measured 3-10x slower.)
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    const int ql = (it * 4 + wv) * GPW + g;
    const int q = slot_query(ql);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      if ((EXP & 2) || ((EXP & 1) && l < L - 1)) continue;
#pragma unroll 1
      for (int pb = 0; pb < P / B; ++pb) {
        const int s0 = ql * LP + l * P + pb * B;
        uint4 o[B];
        float4 c[B];
        float aw[B];
        dvis_v4u r[4 * B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
          o[i] = s_tap_o[s0 + i];
          if ((EXP & 8) && (ql & 1)) { o[i].x = kOOB; o[i].z = kOOB; }   // every second query re-uses its left corners
          if (EXP & 16) { o[i].x = kOOB; o[i].z = kOOB; }                // every query does (a long walk along x)
          if (EXP & 32) { o[i].x = kOOB; o[i].z = kOOB; o[i].y = kOOB; }  // one corner of four left
#if defined(__HIP_DEVICE_COMPILE__)
          if ((EXP & 64) && l == 0) {
            typedef __attribute__((address_space(3))) void *lds_ptr;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[l], (lds_ptr)&s_dma[wv][4 * i][0], 16, o[i].x + lane_bytes, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[l], (lds_ptr)&s_dma[wv][4 * i + 1][0], 16, o[i].y + lane_bytes, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[l], (lds_ptr)&s_dma[wv][4 * i + 2][0], 16, o[i].z + lane_bytes, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[l], (lds_ptr)&s_dma[wv][4 * i + 3][0], 16, o[i].w + lane_bytes, 0, 0, 0);
            continue;
          }
#endif
          r[4 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[i].x + lane_bytes, 0, 0);
          r[4 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[i].y + lane_bytes, 0, 0);
          r[4 * i + 2] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[i].z + lane_bytes, 0, 0);
          r[4 * i + 3] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[i].w + lane_bytes, 0, 0);
        }
        if ((EXP & 64) && l == 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA writes are this wave's own: no barrier needed
#pragma unroll
          for (int k = 0; k < 4 * B; ++k) r[k] = s_dma[wv][k][lane];
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
          c[i] = s_tap_c[s0 + i];
          aw[i] = wf[s0 + i];
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
          const dvis_v4u r1 = r[4 * i], r2 = r[4 * i + 1], r3 = r[4 * i + 2], r4 = r[4 * i + 3];
          const float c1 = c[i].x, c2 = c[i].y, c3 = c[i].z, c4 = c[i].w;
          // reference order: (w1 v1 + w2 v2 + w3 v3 + w4 v4) * weight, accumulated over samples
          a0 += (c1 * __uint_as_float(r1.x) + c2 * __uint_as_float(r2.x) + c3 * __uint_as_float(r3.x) +
                 c4 * __uint_as_float(r4.x)) * aw[i];
          a1 += (c1 * __uint_as_float(r1.y) + c2 * __uint_as_float(r2.y) + c3 * __uint_as_float(r3.y) +
                 c4 * __uint_as_float(r4.y)) * aw[i];
          a2 += (c1 * __uint_as_float(r1.z) + c2 * __uint_as_float(r2.z) + c3 * __uint_as_float(r3.z) +
                 c4 * __uint_as_float(r4.z)) * aw[i];
          a3 += (c1 * __uint_as_float(r1.w) + c2 * __uint_as_float(r2.w) + c3 * __uint_as_float(r3.w) +
                 c4 * __uint_as_float(r4.w)) * aw[i];
        }
      }
    }
    if (q >= 0) {
      float *dst = out_frame + (size_t)q * MD + 4 * j;
      *reinterpret_cast<float4 *>(dst) = make_float4(a0, a1, a2, a3);
    }
  }
}


}  // namespace

void dvis_set_error(const char *, ...) {}

extern "C" __attribute__((visibility("default"))) int msda_probe(int exp, const float *value, const int64_t *shapes, const int64_t *ls,
    const float *ref, int nref, const float *off, int64_t off_stride, const float *lg, int64_t lg_stride, int N, int S,
    int M, int Lq, float *out, void *stream) {
  constexpr int QB = 64;
  dim3 grid(M, (Lq + QB - 1) / QB, N);
#define RUN(E) hipLaunchKernelGGL((msda_fwd_tile_f32<32, 3, 4, true, 2, 2, QB, E>), grid, dim3(256), 0, (hipStream_t)stream, value, shapes, ls, off, off_stride, lg, lg_stride, ref, nref, S, M, Lq, out, nullptr, nullptr, 0)
  switch (exp) {
    case 0: RUN(0); break; case 1: RUN(1); break; case 2: RUN(2); break; case 4: RUN(4); break; case 5: RUN(5); break; case 8: RUN(8); break; case 16: RUN(16); break; case 32: RUN(32); break; case 64: RUN(64); break; case 65: RUN(65); break;
    case 100: hipLaunchKernelGGL((msda_fwd_tile_f32<32, 3, 4, true, 2, 4, QB, 0>), grid, dim3(256), 0, (hipStream_t)stream, value, shapes, ls, off, off_stride, lg, lg_stride, ref, nref, S, M, Lq, out, nullptr, nullptr, 0); break;
    case 101: hipLaunchKernelGGL((msda_fwd_tile_f32<32, 3, 4, true, 2, 1, QB, 0>), grid, dim3(256), 0, (hipStream_t)stream, value, shapes, ls, off, off_stride, lg, lg_stride, ref, nref, S, M, Lq, out, nullptr, nullptr, 0); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
