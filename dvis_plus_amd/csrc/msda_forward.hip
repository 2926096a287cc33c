// Multi-scale deformable attention, forward — gfx950 (MI355X).
//
// Replaces ms_deformable_im2col_gpu_kernel + host wrapper of the reference
// (mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304,
//  ms_deform_attn_cuda.cu:25-85).  Semantics (restated):
//   out[n,q,m,:] = sum_{l,p} w[n,q,m,l,p] * bilinear(value[n, level l, :, m, :], (x*W_l-0.5, y*H_l-0.5))
//   sample counted iff -1 < h_im < H and -1 < w_im < W; corners outside the map contribute 0.
//
// Design (not the reference's one-thread-per-output-channel decomposition):
//   * The op is a GATHER bound by the vector-memory path, not by flops.  A workgroup owns ONE head m
//     and 64 consecutive queries of one frame; blockIdx % M == m, so with M == 8 every XCD (block b
//     runs on XCD b % 8) touches only its own head's 1/8 slice of `value` and that slice
//     (2.5 MB/frame at 720p) stays resident in the XCD's private 4 MB L2.
//   * A (query, head) pair is served by D/4 lanes, each owning 4 channels: one corner of one sample
//     is ONE 16-byte load per lane = a whole 128-byte line per pair (D = 32), 8 pairs per
//     wave-instruction.  The reference issues 4-byte loads and re-reads (loc, w) per channel thread.
//   * (loc, w) of the block are staged once through LDS with coalesced 16-byte reads and then
//     broadcast-read by the D/4 lanes of a pair (identical LDS addresses broadcast, no conflict).
//   * Corner loads go through wave-uniform buffer descriptors, one per level, sized to the level's
//     slice: an out-of-map corner gets an out-of-range offset and the hardware returns 0 — no
//     per-corner branch, no clamped re-read, exact zero padding.
//   * FUSED variant: the LDS stage also applies softmax over (L*P) and loc = ref + off / (W_l, H_l)
//     (ops/modules/ms_deform_attn.py:101-109), so sampling_locations / attention_weights
//     (22 MB per frame-layer at 720p) never exist in HBM.
//   * Any dtype / D / L / P outside the tiled set falls to the generic kernel (one thread per output
//     element, fp32 or fp64 accumulation) — correctness path for fp64, fp16/bf16 and odd D.
#include <stdlib.h>

#include "dvis_common.h"
#include "msda_tap.h"

namespace {

using dvis_msda::kOOB;
using dvis_msda::make_tap;
using dvis_msda::Tap;

// QB queries per workgroup; WPS = register budget in waves/SIMD; B = samples per batch of corner loads.
template <int D, int L, int P, bool FUSED, int WPS, int B, int QB>
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
  const unsigned pix_bytes = (unsigned)MD * 4u;
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
    const float *base = value + (((size_t)n * S + (size_t)level_start[l]) * M + m) * D;
    rs[l] = dvis_make_rsrc_uniform(base, (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * MD + D) * sizeof(float)));
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
          r[4 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[i].x + lane_bytes, 0, 0);
          r[4 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[i].y + lane_bytes, 0, 0);
          r[4 * i + 2] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[i].z + lane_bytes, 0, 0);
          r[4 * i + 3] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[i].w + lane_bytes, 0, 0);
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

// One thread per output element; any dtype, any D / L / P.  fp64 accumulates in fp64, the rest in fp32.
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_generic(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const T *__restrict__ loc, const T *__restrict__ w, size_t total, int S, int M, int D, int L, int Lq, int P,
    T *__restrict__ out) {
  using A = typename dvis_acc<T>::type;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % D);
    size_t t = idx / D;
    const int m = (int)(t % M);
    t /= M;
    const int q = (int)(t % Lq);
    const size_t n = t / Lq;
    const size_t pair = (n * Lq + q) * M + m;
    const T *lp = loc + pair * (size_t)(L * P * 2);
    const T *wp = w + pair * (size_t)(L * P);
    const size_t pix = (size_t)M * D;
    A col = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const T *vb = value + ((n * S + (size_t)level_start[l]) * M + m) * D + c;
      for (int p = 0; p < P; ++p) {
        const A x = dvis_load<A>(lp + 2 * (l * P + p));
        const A y = dvis_load<A>(lp + 2 * (l * P + p) + 1);
        const A aw = dvis_load<A>(wp + l * P + p);
        const A h_im = y * (A)H - (A)0.5, w_im = x * (A)W - (A)0.5;
        if (h_im > (A)-1 && w_im > (A)-1 && h_im < (A)H && w_im < (A)W) {
          const A hf = floor(h_im), wf = floor(w_im);
          const int h0 = (int)hf, w0 = (int)wf;
          const A lh = h_im - hf, lw = w_im - wf, hh = (A)1 - lh, hw = (A)1 - lw;
          A v1 = 0, v2 = 0, v3 = 0, v4 = 0;
          if (h0 >= 0 && w0 >= 0) v1 = dvis_load<A>(vb + ((size_t)h0 * W + w0) * pix);
          if (h0 >= 0 && w0 + 1 <= W - 1) v2 = dvis_load<A>(vb + ((size_t)h0 * W + w0 + 1) * pix);
          if (h0 + 1 <= H - 1 && w0 >= 0) v3 = dvis_load<A>(vb + ((size_t)(h0 + 1) * W + w0) * pix);
          if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) v4 = dvis_load<A>(vb + ((size_t)(h0 + 1) * W + w0 + 1) * pix);
          col += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * aw;
        }
      }
    }
    dvis_store<T, A>(out + idx, col);
  }
}

template <typename T>
int launch_generic(const void *value, const int64_t *shapes, const int64_t *ls, const void *loc, const void *w, int N,
                   int S, int M, int D, int L, int Lq, int P, void *out, hipStream_t st) {
  const size_t total = (size_t)N * Lq * M * D;
  const size_t blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks > 65536 ? 65536 : blocks);
  hipLaunchKernelGGL(msda_fwd_generic<T>, dim3(grid), dim3(256), 0, st, (const T *)value, shapes, ls, (const T *)loc,
                     (const T *)w, total, S, M, D, L, Lq, P, (T *)out);
  return dvis_check_launch("msda_fwd_generic");
}

// Developer knob (tools/msda_sweep.py): DVIS_MSDA_VARIANT selects the tile-kernel build variant.
// Schedules that were measured on MI355X and removed (bit-identical results, history in git and DESIGN.md §3.1):
// 8x8 query tiles (neutral: 49.0 vs 49.2 us), band-interleaved chunk order (HBM fetch -19 %, time +4 %).
int tile_variant() {
  static const int v = [] {
    const char *e = getenv("DVIS_MSDA_VARIANT");
    return e ? atoi(e) : 0;
  }();
  return v;
}

template <int D, int L, int P, bool FUSED, int WPS, int B, int QB>
int launch_variant(const float *value, const int64_t *shapes, const int64_t *ls, const float *a, int64_t a_stride,
                   const float *b, int64_t b_stride, const float *refp, int nref, int N, int S, int M, int Lq, float *out,
                   hipStream_t st, const float *pos_off, const float *pos_logit, int64_t pos_stride) {
  const int nchunks = (Lq + QB - 1) / QB;
  if (nchunks > 65535 || N > 65535) {
    dvis_set_error("msda: grid too large (Lq/%d and N must be <= 65535)", QB);
    return DVIS_E_ARG;
  }
  hipLaunchKernelGGL((msda_fwd_tile_f32<D, L, P, FUSED, WPS, B, QB>), dim3(M, nchunks, N), dim3(256), 0, st, value, shapes,
                     ls, a, a_stride, b, b_stride, refp, nref, S, M, Lq, out, pos_off, pos_logit, pos_stride);
  return dvis_check_launch("msda_fwd_tile_f32");
}

template <int D, int L, int P, bool FUSED>
int launch_tile(const float *value, const int64_t *shapes, const int64_t *ls, const float *a, int64_t a_stride,
                const float *b, int64_t b_stride, const float *refp, int nref, int N, int S, int M, int Lq, float *out,
                hipStream_t st, const float *pos_off, const float *pos_logit, int64_t pos_stride) {
#define DVIS_LAUNCH_VARIANT(wps, bsz, qb) \
  return launch_variant<D, L, P, FUSED, wps, bsz, qb>(value, shapes, ls, a, a_stride, b, b_stride, refp, nref, N, S, M, Lq, \
                                                      out, st, pos_off, pos_logit, pos_stride)
  constexpr int QMIN = 4 * (64 / (D / 4));   // queries covered by one pass of the 4 waves
  // (min waves/SIMD the register allocator must allow) x (samples per load batch) x (queries per block).  Measured on
  // MI355X, 720p, 30 frames: 35.2 / 35.2 / 37.0 us per frame-layer — occupancy-insensitive (4 vs 8 waves/SIMD), the
  // per-CU L1 data path is the limit.  With 2*QMIN queries the taps take 33 KB of LDS -> 4 workgroups per CU.
  switch (tile_variant()) {
    case 1: DVIS_LAUNCH_VARIANT(6, 2, QMIN);
    case 2: DVIS_LAUNCH_VARIANT(4, 4, QMIN);
    default: DVIS_LAUNCH_VARIANT(2, 2, 2 * QMIN);
  }
#undef DVIS_LAUNCH_VARIANT
}

template <bool FUSED>
int dispatch_tile(int D, int L, int P, const float *value, const int64_t *shapes, const int64_t *ls, const float *a,
                  int64_t a_stride, const float *b, int64_t b_stride, const float *refp, int nref, int N, int S, int M,
                  int Lq, float *out, hipStream_t st, bool *handled, const float *pos_off = nullptr,
                  const float *pos_logit = nullptr, int64_t pos_stride = 0) {
  *handled = true;
#define DVIS_TILE_CASE(d, l, p)  \
  if (D == d && L == l && P == p) \
    return launch_tile<d, l, p, FUSED>(value, shapes, ls, a, a_stride, b, b_stride, refp, nref, N, S, M, Lq, out, st, \
                                       pos_off, pos_logit, pos_stride);
  DVIS_TILE_CASE(32, 3, 4)
  DVIS_TILE_CASE(32, 4, 4)
  DVIS_TILE_CASE(32, 1, 4)
  DVIS_TILE_CASE(64, 1, 4)
  DVIS_TILE_CASE(64, 3, 4)
  DVIS_TILE_CASE(64, 4, 4)
#undef DVIS_TILE_CASE
  *handled = false;
  return DVIS_OK;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

DVIS_EXPORT int dvis_msda_forward(int dtype, const void *value, const int64_t *shapes, const int64_t *level_start,
                                  const void *loc, const void *w, int N, int S, int M, int D, int L, int Lq, int P,
                                  void *out, void *stream) {
  DVIS_REQUIRE(N >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_forward: bad sizes");
  if (N == 0 || Lq == 0) return DVIS_OK;   // empty batch: nothing to write (pointers may be null)
  DVIS_REQUIRE(value && shapes && level_start && loc && w && out, "msda_forward: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DVIS_F32) {
    const bool fits = (size_t)S * M * D * sizeof(float) < 0x7fffffffu &&
                      (size_t)Lq * M * L * P * 2 * sizeof(float) < 0x7fffffffu;
    if (fits && aligned16(value) && aligned16(loc) && aligned16(w) && aligned16(out)) {
      bool handled = false;
      int rc = dispatch_tile<false>(D, L, P, (const float *)value, shapes, level_start, (const float *)loc, 0,
                                    (const float *)w, 0, nullptr, 0, N, S, M, Lq, (float *)out, st, &handled);
      if (handled) return rc;
    }
    return launch_generic<float>(value, shapes, level_start, loc, w, N, S, M, D, L, Lq, P, out, st);
  }
  if (dtype == DVIS_F64) return launch_generic<double>(value, shapes, level_start, loc, w, N, S, M, D, L, Lq, P, out, st);
  if (dtype == DVIS_F16) return launch_generic<__half>(value, shapes, level_start, loc, w, N, S, M, D, L, Lq, P, out, st);
  if (dtype == DVIS_BF16)
    return launch_generic<__hip_bfloat16>(value, shapes, level_start, loc, w, N, S, M, D, L, Lq, P, out, st);
  dvis_set_error("msda_forward: unsupported dtype %d", dtype);
  return DVIS_E_ARG;
}

DVIS_EXPORT int dvis_msda_fused_forward_pos(const float *value, const int64_t *shapes, const int64_t *level_start,
                                            const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                            const float *logits, int64_t logit_stride, const float *pos_offsets,
                                            const float *pos_logits, int64_t pos_stride, int N, int S, int M, int D,
                                            int L, int Lq, int P, float *out, const int64_t *shapes_host, void *stream) {
  DVIS_REQUIRE(N >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_fused_forward: bad sizes");
  if (N == 0 || Lq == 0) return DVIS_OK;
  DVIS_REQUIRE(value && shapes && level_start && ref && offsets && logits && out, "msda_fused_forward: null pointer");
  DVIS_REQUIRE(Nref == 1 || Nref == N, "msda_fused_forward: Nref must be 1 or N");
  DVIS_REQUIRE(off_stride >= (int64_t)M * L * P * 2 && logit_stride >= (int64_t)M * L * P,
               "msda_fused_forward: row strides too small");
  DVIS_REQUIRE(off_stride % 4 == 0 && logit_stride % 4 == 0 && aligned16(offsets) && aligned16(logits) &&
                   aligned16(value) && aligned16(out),
               "msda_fused_forward: 16-byte alignment required");
  DVIS_REQUIRE((size_t)S * M * D * sizeof(float) < 0x7fffffffu, "msda_fused_forward: frame slice >= 2 GiB");
  bool handled = false;
  DVIS_REQUIRE((size_t)Lq * (size_t)(off_stride > logit_stride ? off_stride : logit_stride) * sizeof(float) < 0x7fffffffu,
               "msda_fused_forward: one frame of offsets/logits must stay below 2 GiB");
  const bool has_pos = pos_offsets != nullptr || pos_logits != nullptr;
  if (has_pos) {
    DVIS_REQUIRE(pos_offsets && pos_logits && pos_stride >= (int64_t)M * L * P * 2 && pos_stride % 4 == 0 &&
                     aligned16(pos_offsets) && aligned16(pos_logits),
                 "msda_fused_forward: position rows need both pointers, 16-byte alignment and a row stride multiple of 4");
    int rc2 = dispatch_tile<true>(D, L, P, value, shapes, level_start, offsets, off_stride, logits, logit_stride, ref, Nref,
                                  N, S, M, Lq, out, (hipStream_t)stream, &handled, pos_offsets, pos_logits, pos_stride);
    if (handled) return rc2;
    dvis_set_error("msda_fused_forward: unsupported (D=%d, L=%d, P=%d)", D, L, P);
    return DVIS_E_UNSUPPORTED;
  }
  int rc;
  rc = dispatch_tile<true>(D, L, P, value, shapes, level_start, offsets, off_stride, logits, logit_stride, ref,
                           Nref, N, S, M, Lq, out, (hipStream_t)stream, &handled);
  if (handled) return rc;
  dvis_set_error("msda_fused_forward: unsupported (D=%d, L=%d, P=%d); supported D in {32,64}, (L,P) in {(1,4),(3,4),(4,4)}",
                 D, L, P);
  return DVIS_E_UNSUPPORTED;
}

DVIS_EXPORT int dvis_msda_fused_forward(const float *value, const int64_t *shapes, const int64_t *level_start,
                                        const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                        const float *logits, int64_t logit_stride, int N, int S, int M, int D, int L,
                                        int Lq, int P, float *out, const int64_t *shapes_host, void *stream) {
  return dvis_msda_fused_forward_pos(value, shapes, level_start, ref, Nref, offsets, off_stride, logits, logit_stride,
                                     nullptr, nullptr, 0, N, S, M, D, L, Lq, P, out, shapes_host, stream);
}
