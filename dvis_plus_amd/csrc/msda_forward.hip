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

#include <algorithm>

#include "dvis_common.h"
#include "msda_tap.h"

int dvis_msda_l0lds_launch(const float *value, const int64_t *shapes, const int64_t *level_start, const float *ref, int nref,
                           const float *offsets, int64_t off_stride, const float *logits, int64_t logit_stride, int N, int S,
                           int M, int D, int L, int Lq, int P, float *out, const int64_t *shapes_host, hipStream_t st,
                           bool *handled);

namespace {

using dvis_msda::kOOB;
using dvis_msda::make_tap;
using dvis_msda::Tap;

constexpr int kQB = 64;        // queries per workgroup (tiled kernel)

constexpr int kMaxOrder = 640;   // chunks of 64 queries that an order table can describe (kernel-argument space)

// How the blocks of a launch map to queries (passed by value, wave-uniform).
struct QueryTiling {
  int enabled;        // 0: blockIdx.y * 64 consecutive queries;  1: 8x8 pixel tiles per level
  int tiles_cum[5];   // first block index of each level's tiles (+ total)
  int tiles_x[4];     // tiles per row of each level
  int use_order;      // 1: linear 64-query chunks, but issued in the order given below
  unsigned short order[kMaxOrder];   // chunk processed by block y (band-interleaved over the levels)
};

template <int D, int L, int P, bool FUSED, int WPS, int B>
__global__ __launch_bounds__(256, WPS) void msda_fwd_tile_f32(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc_or_off, int64_t off_stride, const float *__restrict__ w_or_logit,
    int64_t logit_stride, const float *__restrict__ refp, int nref, int S, int M, int Lq, QueryTiling tiling,
    float *__restrict__ out) {
  constexpr int LP = L * P;
  constexpr int G = D / 4;          // lanes per (query, head) pair
  constexpr int GPW = 64 / G;       // pairs per wave-instruction
  constexpr int LOCV = LP / 2;      // float4s of (x, y) per pair
  constexpr int WV = LP / 4;        // float4s of weights per pair
  constexpr int ITERS = kQB / (4 * GPW);
  static_assert(LP % 4 == 0 && D % 4 == 0 && 64 % G == 0 && P % B == 0 && B % 2 == 0, "tile shape");

  __shared__ float4 s_loc[kQB * LOCV];
  __shared__ float4 s_w[kQB * WV];

  // grid = (M, ceil(Lq/64), N): x is the fastest dispatch dimension, so linear id % 8 == m % 8 -> head m on XCD m % 8.
  // blockIdx.* are SGPRs: everything derived from them (bases, descriptors) is wave-uniform.
  const int tid = threadIdx.x;
  const int m = blockIdx.x;
  const int n = blockIdx.z;
  const int MD = M * D;

  int Hs[L], Ws[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
  }

  // ---- which 64 queries does this block own?  (all scalar / wave-uniform)
  //  linear : 64 consecutive query indices.
  //  2-D    : (encoder self-attention: the queries ARE the pixels of the L maps) one 8x8 pixel tile of one level, so
  //           the block's sampling footprints overlap in BOTH directions: ~2.5x fewer distinct value lines per block
  //           than a 64-pixel row segment -> higher L1 hit rate, less L2->L1 traffic (the measured bound).
  int tl_base = 0, tl_w = 0, tl_h = 0, tl_y0 = 0, tl_x0 = 0;
  const int q0 = (tiling.use_order ? (int)tiling.order[blockIdx.y] : (int)blockIdx.y) * kQB;
  if (tiling.enabled) {
    int l = 0;
#pragma unroll
    for (int ll = 1; ll < L; ++ll)
      if ((int)blockIdx.y >= tiling.tiles_cum[ll]) l = ll;
    const int t = blockIdx.y - tiling.tiles_cum[l];
    tl_y0 = (t / tiling.tiles_x[l]) * 8;
    tl_x0 = (t % tiling.tiles_x[l]) * 8;
#pragma unroll
    for (int ll = 0; ll < L; ++ll)
      if (l == ll) { tl_h = Hs[ll]; tl_w = Ws[ll]; tl_base = (int)level_start[ll]; }
  }
  // local slot (0..63) -> global query index, or -1 when the slot is empty
  auto slot_query = [&](int ql) -> int {
    if (tiling.enabled) {
      const int y = tl_y0 + (ql >> 3), x = tl_x0 + (ql & 7);
      return (y < tl_h && x < tl_w) ? tl_base + y * tl_w + x : -1;
    }
    return q0 + ql < Lq ? q0 + ql : -1;
  };

  // ---- stage (loc, w) [or raw (offsets, logits)] of this block's 64 queries x 1 head into LDS.
  // Descriptors cover this frame's rows; empty slots get an out-of-range offset and read as 0 without a branch.
  {
    const size_t row0 = (size_t)n * Lq;
    const float *lbase = FUSED ? loc_or_off + row0 * off_stride + (size_t)m * (LP * 2)
                               : loc_or_off + (row0 * M + m) * (size_t)(LP * 2);
    const float *wbase = FUSED ? w_or_logit + row0 * logit_stride + (size_t)m * LP
                               : w_or_logit + (row0 * M + m) * (size_t)LP;
    const unsigned lrow = (unsigned)((FUSED ? (size_t)off_stride : (size_t)M * LP * 2) * sizeof(float));
    const unsigned wrow = (unsigned)((FUSED ? (size_t)logit_stride : (size_t)M * LP) * sizeof(float));
    const __amdgpu_buffer_rsrc_t lrs = dvis_make_rsrc_uniform(lbase, (unsigned)(Lq - 1) * lrow + LP * 2 * 4);
    const __amdgpu_buffer_rsrc_t wrs = dvis_make_rsrc_uniform(wbase, (unsigned)(Lq - 1) * wrow + LP * 4);
    for (int i = tid; i < kQB * LOCV; i += 256) {
      const int ql = i / LOCV, k = i - ql * LOCV;
      const int q = slot_query(ql);
      const dvis_v4u v = __builtin_amdgcn_raw_buffer_load_b128(
          lrs, q >= 0 ? (unsigned)q * lrow + (unsigned)k * 16u : kOOB, 0, 0);
      s_loc[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    for (int i = tid; i < kQB * WV; i += 256) {
      const int ql = i / WV, k = i - ql * WV;
      const int q = slot_query(ql);
      const dvis_v4u v = __builtin_amdgcn_raw_buffer_load_b128(
          wrs, q >= 0 ? (unsigned)q * wrow + (unsigned)k * 16u : kOOB, 0, 0);
      s_w[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  }
  if (FUSED) {
    __syncthreads();
    float *lf = reinterpret_cast<float *>(s_loc);
    float *wf = reinterpret_cast<float *>(s_w);
    // loc = ref + off / (W_l, H_l)
    for (int i = tid; i < kQB * LP; i += 256) {
      const int ql = i / LP, s = i - ql * LP;
      const int l = s / P;
      const int q = slot_query(ql);
      if (q >= 0) {
        int Hl = Hs[0], Wl = Ws[0];
#pragma unroll
        for (int ll = 1; ll < L; ++ll)
          if (l == ll) { Hl = Hs[ll]; Wl = Ws[ll]; }
        const size_t rrow = ((size_t)(nref == 1 ? 0 : n) * Lq + q) * L + l;
        const float2 r = *reinterpret_cast<const float2 *>(refp + rrow * 2);
        lf[ql * LP * 2 + 2 * s] = r.x + lf[ql * LP * 2 + 2 * s] / (float)Wl;
        lf[ql * LP * 2 + 2 * s + 1] = r.y + lf[ql * LP * 2 + 2 * s + 1] / (float)Hl;
      }
    }
    // softmax over the L*P logits of each (query, head)
    if (tid < kQB) {
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
  }
  __syncthreads();

  // ---- per-level buffer descriptors over this (frame, head) slice of `value`
  __amdgpu_buffer_rsrc_t rs[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float *base = value + (((size_t)n * S + (size_t)level_start[l]) * M + m) * D;
    rs[l] = dvis_make_rsrc_uniform(base, (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * MD + D) * sizeof(float)));
  }

  const int lane = tid & 63, wv = tid >> 6;
  const int g = lane / G, j = lane - g * G;
  const unsigned pix_bytes = (unsigned)MD * 4u;
  const unsigned lane_bytes = (unsigned)j * 16u;
  float *const out_frame = out + ((size_t)n * Lq * M + m) * D;   // uniform

  const float *lds_loc = reinterpret_cast<const float *>(s_loc);
  const float *lds_w = reinterpret_cast<const float *>(s_w);

  // Latency is hidden by WAVES, not by a deep per-wave pipeline: each wave keeps one batch of B samples
  // (4*B corner loads) in flight, reads that batch's (x, y, w) from LDS just in time, and stays within the
  // register budget of WPS waves/SIMD.  (A fully unrolled 12-sample body makes hipcc hoist all 48 loads and
  // spill to scratch; measured 3-10x slower.)
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    const int ql = (it * 4 + wv) * GPW + g;
    const int q = slot_query(ql);
    const bool active = q >= 0;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int H = Hs[l], W = Ws[l];
#pragma unroll 1
      for (int pb = 0; pb < P / B; ++pb) {
        const int s0 = l * P + pb * B;
        float xy[2 * B], aw[B];
#pragma unroll
        for (int i = 0; i < B / 2; ++i) {
          const float4 v = *reinterpret_cast<const float4 *>(lds_loc + ql * (LP * 2) + 2 * s0 + 4 * i);
          xy[4 * i] = v.x; xy[4 * i + 1] = v.y; xy[4 * i + 2] = v.z; xy[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < B / 2; ++i) {
          const float2 v = *reinterpret_cast<const float2 *>(lds_w + ql * LP + s0 + 2 * i);
          aw[2 * i] = v.x; aw[2 * i + 1] = v.y;
        }
        Tap t[B];
        dvis_v4u r[4 * B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
          t[i] = make_tap(xy[2 * i], xy[2 * i + 1], H, W, active, pix_bytes, lane_bytes);
#pragma unroll
          for (int c = 0; c < 4; ++c) r[4 * i + c] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], t[i].o[c], 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
          const dvis_v4u r1 = r[4 * i], r2 = r[4 * i + 1], r3 = r[4 * i + 2], r4 = r[4 * i + 3];
          const float c1 = t[i].c[0], c2 = t[i].c[1], c3 = t[i].c[2], c4 = t[i].c[3];
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
    if (active) {
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

// Developer knob (tools/msda_sweep.py): DVIS_MSDA_VARIANT=0..6 selects the tile-kernel build variant.
// 8x8 query tiling is OFF by default: measured neutral on MI355X (49.0 vs 49.2 us/frame-layer) — the kernel is bound by
// the 64 B/clk/CU L1 data path (1 KB per wave-instruction, ~21 clk each = 75 % of that limit), not by L1 misses.
// DVIS_MSDA_TILE2D=1 enables it for experiments.
bool tile2d_enabled() {
  static const bool v = [] {
    const char *e = getenv("DVIS_MSDA_TILE2D");
    return e != nullptr && atoi(e) != 0;
  }();
  return v;
}

// Band-interleaved chunk order: measured on MI355X (30 frames/launch): HBM fetch -19 % (2*FETCH_SIZE 2.85 -> 2.31 GB),
// L2 hit rate 70.6 -> 75.1 %, but +4 % time (39.3 vs 37.9 us/frame-layer) — the kernel is bound by the L1 data path,
// not by HBM.  OFF by default; DVIS_MSDA_BAND_ORDER=1 enables it.
bool band_order_enabled() {
  static const bool v = [] {
    const char *e = getenv("DVIS_MSDA_BAND_ORDER");
    return e != nullptr && atoi(e) != 0;
  }();
  return v;
}

int tile_variant() {
  static const int v = [] {
    const char *e = getenv("DVIS_MSDA_VARIANT");
    return e ? atoi(e) : 0;
  }();
  return v;
}

template <int D, int L, int P, bool FUSED>
int launch_tile(const float *value, const int64_t *shapes, const int64_t *ls, const float *a, int64_t a_stride,
                const float *b, int64_t b_stride, const float *refp, int nref, int N, int S, int M, int Lq, float *out,
                hipStream_t st, const int64_t *shapes_host) {
  QueryTiling tiling = {};
  int nchunks = (Lq + kQB - 1) / kQB;
  if (shapes_host != nullptr && L <= 4 && tile2d_enabled()) {
    long long total = 0;
    int cum = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes_host[2 * l], W = (int)shapes_host[2 * l + 1];
      total += (long long)H * W;
      tiling.tiles_cum[l] = cum;
      tiling.tiles_x[l] = (W + 7) / 8;
      cum += ((H + 7) / 8) * ((W + 7) / 8);
    }
    tiling.tiles_cum[L] = cum;
    if (total == Lq && total == S) {   // the queries are exactly the pixels of the maps (encoder self-attention)
      tiling.enabled = 1;
      nchunks = cum;
    }
  }
  if (!tiling.enabled && shapes_host != nullptr && nchunks <= kMaxOrder && band_order_enabled()) {
    // Self-attention over L maps: the chunks of level 0, then 1, then 2 each sweep the WHOLE per-head value slice, so
    // it is fetched from HBM once per level (measured 1.9x the algorithmic traffic, L2 hit rate 68 %).  Issue the
    // chunks sorted by the image row band they belong to instead: all levels' queries of one band run together and
    // each value line is fetched once.
    long long total = 0, starts[5] = {0, 0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
      starts[l] = total;
      total += shapes_host[2 * l] * shapes_host[2 * l + 1];
    }
    starts[L] = total;
    if (total == Lq && total == S) {
      float key[kMaxOrder];
      for (int c = 0; c < nchunks; ++c) {
        const long long q = (long long)c * kQB;
        int l = 0;
        while (l + 1 < L && q >= starts[l + 1]) ++l;
        const long long y = (q - starts[l]) / shapes_host[2 * l + 1];
        key[c] = ((float)y + 0.5f) / (float)shapes_host[2 * l];
        tiling.order[c] = (unsigned short)c;
      }
      std::stable_sort(tiling.order, tiling.order + nchunks,
                       [&](unsigned short a, unsigned short b) { return key[a] < key[b]; });
      tiling.use_order = 1;
    }
  }
  if (nchunks > 65535 || N > 65535) {
    dvis_set_error("msda: grid too large (Lq/64 and N must be <= 65535)");
    return DVIS_E_ARG;
  }
  const dim3 grid(M, nchunks, N), block(256);
#define DVIS_LAUNCH_VARIANT(wps, bsz)                                                                              \
  hipLaunchKernelGGL((msda_fwd_tile_f32<D, L, P, FUSED, wps, bsz>), grid, block, 0, st, value, shapes, ls, a, a_stride,  \
                     b, b_stride, refp, nref, S, M, Lq, tiling, out)
  switch (tile_variant()) {   // register budget (waves/SIMD) x samples per load batch; default picked by measurement
    case 1: DVIS_LAUNCH_VARIANT(6, 2); break;
    case 2: DVIS_LAUNCH_VARIANT(4, 2); break;
    case 3: DVIS_LAUNCH_VARIANT(4, 4); break;
    case 4: DVIS_LAUNCH_VARIANT(5, 4); break;
    case 5: DVIS_LAUNCH_VARIANT(3, 4); break;
    case 6: DVIS_LAUNCH_VARIANT(2, 4); break;
    default: DVIS_LAUNCH_VARIANT(8, 2); break;
  }
#undef DVIS_LAUNCH_VARIANT
  return dvis_check_launch("msda_fwd_tile_f32");
}

template <bool FUSED>
int dispatch_tile(int D, int L, int P, const float *value, const int64_t *shapes, const int64_t *ls, const float *a,
                  int64_t a_stride, const float *b, int64_t b_stride, const float *refp, int nref, int N, int S, int M,
                  int Lq, float *out, hipStream_t st, bool *handled, const int64_t *shapes_host = nullptr) {
  *handled = true;
#define DVIS_TILE_CASE(d, l, p)  \
  if (D == d && L == l && P == p) \
    return launch_tile<d, l, p, FUSED>(value, shapes, ls, a, a_stride, b, b_stride, refp, nref, N, S, M, Lq, out, st, \
                                       shapes_host);
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

DVIS_EXPORT int dvis_msda_fused_forward(const float *value, const int64_t *shapes, const int64_t *level_start,
                                        const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                        const float *logits, int64_t logit_stride, int N, int S, int M, int D, int L,
                                        int Lq, int P, float *out, const int64_t *shapes_host, void *stream) {
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
  // preferred: persistent kernel with the coarsest value map LDS-resident (msda_forward_lds.hip)
  int rc = dvis_msda_l0lds_launch(value, shapes, level_start, ref, Nref, offsets, off_stride, logits, logit_stride, N, S, M,
                                  D, L, Lq, P, out, shapes_host, (hipStream_t)stream, &handled);
  if (handled) return rc;
  rc = dispatch_tile<true>(D, L, P, value, shapes, level_start, offsets, off_stride, logits, logit_stride, ref,
                           Nref, N, S, M, Lq, out, (hipStream_t)stream, &handled, shapes_host);
  if (handled) return rc;
  dvis_set_error("msda_fused_forward: unsupported (D=%d, L=%d, P=%d); supported D in {32,64}, (L,P) in {(1,4),(3,4),(4,4)}",
                 D, L, P);
  return DVIS_E_UNSUPPORTED;
}
