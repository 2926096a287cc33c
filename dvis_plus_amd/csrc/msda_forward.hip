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

// Storage types of the tiled kernel: a lane always moves 16 bytes per corner — 4 fp32 channels or 8 fp16 / bf16 channels —
// and accumulates in fp32 (the reference dispatches float / double only, ms_deform_attn_cuda.cu:69; under autocast its
// ViT-Adapter extractors therefore fall into the grid_sample path, ms_deform_attn.py:116-121).
template <typename T> struct Chan;
template <> struct Chan<float> {
  static constexpr int kPerLane = 4;
  static __device__ __forceinline__ void unpack(const dvis_v4u &r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
  }
  static __device__ __forceinline__ void store(float *dst, const float (&a)[4]) {
    *reinterpret_cast<float4 *>(dst) = make_float4(a[0], a[1], a[2], a[3]);
  }
  static __device__ __forceinline__ float load(const float *p) { return *p; }
  static __device__ __forceinline__ float2 load2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
  static __device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
};
template <> struct Chan<__half> {
  static constexpr int kPerLane = 8;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void unpack(const dvis_v4u &r, float (&f)[8]) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const h2 v = __builtin_bit_cast(h2, w[i]);
      f[2 * i] = (float)v[0];
      f[2 * i + 1] = (float)v[1];
    }
  }
  static __device__ __forceinline__ void store(__half *dst, const float (&a)[8]) {
    dvis_v4u o;
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h2 v;
      v[0] = (_Float16)a[2 * i];
      v[1] = (_Float16)a[2 * i + 1];
      w[i] = __builtin_bit_cast(unsigned, v);
    }
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *reinterpret_cast<dvis_v4u *>(dst) = o;
  }
  static __device__ __forceinline__ float load(const __half *p) { return __half2float(*p); }
  static __device__ __forceinline__ float2 load2(const __half *p) {
    const h2 v = *reinterpret_cast<const h2 *>(p);
    return make_float2((float)v[0], (float)v[1]);
  }
  static __device__ __forceinline__ float4 load4(const __half *p) {    // 8 bytes
    const float2 a = load2(p), b = load2(p + 2);
    return make_float4(a.x, a.y, b.x, b.y);
  }
};
template <> struct Chan<__hip_bfloat16> {
  static constexpr int kPerLane = 8;
  static __device__ __forceinline__ void unpack(const dvis_v4u &r, float (&f)[8]) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);             // bf16 = the upper half of an fp32
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ unsigned short rne(float x) {   // round to nearest even, NaN kept quiet
    const unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
  static __device__ __forceinline__ void store(__hip_bfloat16 *dst, const float (&a)[8]) {
    dvis_v4u o;
    o.x = (unsigned)rne(a[0]) | ((unsigned)rne(a[1]) << 16);
    o.y = (unsigned)rne(a[2]) | ((unsigned)rne(a[3]) << 16);
    o.z = (unsigned)rne(a[4]) | ((unsigned)rne(a[5]) << 16);
    o.w = (unsigned)rne(a[6]) | ((unsigned)rne(a[7]) << 16);
    *reinterpret_cast<dvis_v4u *>(dst) = o;
  }
  static __device__ __forceinline__ float load(const __hip_bfloat16 *p) {
    return __uint_as_float((unsigned)(*reinterpret_cast<const unsigned short *>(p)) << 16);
  }
  static __device__ __forceinline__ float2 load2(const __hip_bfloat16 *p) {
    const unsigned w = *reinterpret_cast<const unsigned *>(p);
    return make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
  }
  static __device__ __forceinline__ float4 load4(const __hip_bfloat16 *p) {
    const float2 a = load2(p), b = load2(p + 2);
    return make_float4(a.x, a.y, b.x, b.y);
  }
};

// Query order of a workgroup.  Default: QB consecutive queries.  With a TileMap (the deformable encoder's self-attention:
// query q of level lq IS pixel (y, x) of that level's map, msdeformattn.py:61-89) a workgroup owns an 8 x 8 TILE of query
// pixels of one level: neighbouring queries sample neighbouring locations, so a gathered 128-byte corner line is reused
// by up to four queries of the block through the CU's L1 instead of two (measured on MI355X, 30 frames of 720p, init-rule
// offsets + learned part: 35.3 vs 37.0 us per frame-layer; bit-identical results — it is only a schedule).
constexpr int kMaxLevels = 4;
struct TileMap {
  int tile_start[kMaxLevels + 1];   // first block index of every level (prefix sums); [L] = number of blocks
  int tiles_x[kMaxLevels];          // 8-pixel tiles per row of every level
  int h[kMaxLevels], w[kMaxLevels], q0[kMaxLevels];   // map size and first query of every level
  int tiled;                        // the 8 x 8 geometry above is valid (else: QB consecutive queries per workgroup)
  // FUSED form: floats between two heads' parameters inside a projection row; 0 = the reference's layout (all heads'
  // offsets, then all heads' logits: L*P*2 and L*P).  "Slots" = [head m: 2LP offsets | LP logits | pad]: a (query, head)
  // pair then touches ONE contiguous run of its row instead of two (fewer 128-byte lines shared between the 8 XCDs).
  int off_hs, logit_hs;
  // `value` is HEAD-MAJOR (M, N, S, D) instead of the reference's (N, S, M, D): a pixel's D channels of one head are
  // still one 128-byte line, but neighbouring pixels of the head are ADJACENT lines (two x-neighbouring corners = 256
  // contiguous bytes) instead of M * D * 4 = 1 KB apart — what the value projection writes through dvis_gemm_nt_hm.
  int value_hm;
};

bool make_tile_map(const int64_t *shapes_host, int L, int Lq, TileMap *tm) {
  if (shapes_host == nullptr || L > kMaxLevels) return false;
  long total = 0;
  int tiles = 0;
  for (int l = 0; l < kMaxLevels; ++l) {
    const long H = l < L ? shapes_host[2 * l] : 0, W = l < L ? shapes_host[2 * l + 1] : 0;
    if (l < L && (H <= 0 || W <= 0)) return false;
    tm->tile_start[l] = tiles;
    tm->tiles_x[l] = (int)((W + 7) / 8);
    tm->h[l] = (int)H;
    tm->w[l] = (int)W;
    tm->q0[l] = (int)total;
    tiles += tm->tiles_x[l] * (int)((H + 7) / 8);
    total += H * W;
  }
  tm->tile_start[kMaxLevels] = tiles;
  tm->tiled = total == Lq && tiles <= 65535;     // the queries are exactly the pixels of the maps
  return tm->tiled != 0;
}

// QB queries per workgroup; WPS = register budget in waves/SIMD; B = samples per batch of corner loads.
// T: storage type of value / locations (or raw offsets) / weights (or raw logits) / output; the arithmetic is fp32.  (Round 4: the
// FUSED form takes half-precision projections too — the reference evaluates under autocast, train_net_video.py:259, where the
// ViT-Adapter's extractors hand the op fp16 tensors.)
template <typename T, int D, int L, int P, bool FUSED, int WPS, int B, int QB, bool TILE2D>
__global__ __launch_bounds__(256, WPS) void msda_fwd_tile(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const T *__restrict__ loc_or_off, int64_t off_stride, const T *__restrict__ w_or_logit,
    int64_t logit_stride, const float *__restrict__ refp, int nref, int S, int M, int Lq, T *__restrict__ out,
    const float *__restrict__ pos_off, const float *__restrict__ pos_logit, int64_t pos_stride, TileMap tm) {
  static_assert(!TILE2D || QB == 64, "an 8 x 8 tile per workgroup");
  constexpr int LP = L * P;
  constexpr int CPL = Chan<T>::kPerLane;   // channels per lane (16 bytes)
  constexpr int G = D / CPL;        // lanes per (query, head) pair
  constexpr int GPW = 64 / G;       // pairs per wave-instruction
  constexpr int ITERS = (QB + 4 * GPW - 1) / (4 * GPW);
  static_assert(LP % 4 == 0 && D % CPL == 0 && 64 % G == 0 && P % B == 0, "tile shape");

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
  // 2-D: which level and which 8 x 8 tile this block is (wave-uniform: block index and kernel arguments are SGPRs)
  int t_h = 0, t_w = 0, t_q0 = 0, t_y = 0, t_x = 0;
  if constexpr (TILE2D) {
    int lq = 0;
#pragma unroll
    for (int l = 1; l < L; ++l) lq = (int)blockIdx.y >= tm.tile_start[l] ? l : lq;
    int txs = tm.tiles_x[0], t0 = tm.tile_start[0];
    t_h = tm.h[0]; t_w = tm.w[0]; t_q0 = tm.q0[0];
#pragma unroll
    for (int l = 1; l < L; ++l)
      if (lq == l) { txs = tm.tiles_x[l]; t0 = tm.tile_start[l]; t_h = tm.h[l]; t_w = tm.w[l]; t_q0 = tm.q0[l]; }
    const int trel = (int)blockIdx.y - t0;
    t_y = (trel / txs) * 8;
    t_x = (trel - (trel / txs) * txs) * 8;
  }
  // local slot -> global query index, or -1 when the slot is past the end (2-D: outside the map)
  auto slot_query = [&](int ql) -> int {
    if constexpr (TILE2D) {
      const int y = t_y + (ql >> 3), x = t_x + (ql & 7);
      return (y < t_h && x < t_w) ? t_q0 + y * t_w + x : -1;
    } else {
      return q0 + ql < Lq ? q0 + ql : -1;
    }
  };

  // ---- set-up: thread (query tid / P, point tid % P) reads ITS parameters of all L levels straight into registers —
  // raw offsets, reference points and the pair's L*P logits (fused) or locations and weights — in ONE global round trip,
  // then softmax, loc = ref + off / (W_l, H_l) and the taps, and ONE barrier.  (The first form staged the rows in LDS,
  // synchronised, loaded the reference points, computed, synchronised again: with the gather switched off that set-up
  // alone took 12.7-15.5 us per 720p frame-layer, and with the loads switched off the kernel still took 23.5 of 35 us.)
  const unsigned pix_bytes = (unsigned)(tm.value_hm ? D : MD) * (unsigned)sizeof(T);
  static_assert(QB * P <= 256, "one set-up thread per (query, point)");
  if (tid < QB * P) {
    const int ql = tid / P, p = tid - ql * P;
    const int q = slot_query(ql);
    const bool active = q >= 0;
    const size_t qq = active ? q : 0;
    float2 xy[L];
    float aw[L];
    if constexpr (FUSED) {
      const int off_hs = tm.off_hs ? tm.off_hs : LP * 2, logit_hs = tm.logit_hs ? tm.logit_hs : LP;
      const T *orow = loc_or_off + ((size_t)n * Lq + qq) * off_stride + (size_t)m * off_hs;
      const T *lrow = w_or_logit + ((size_t)n * Lq + qq) * logit_stride + (size_t)m * logit_hs;
      float2 ro[L], rr[L];
      float4 rl[LP / 4];
#pragma unroll
      for (int l = 0; l < L; ++l) {
        ro[l] = Chan<T>::load2(orow + 2 * (l * P + p));
        rr[l] = *reinterpret_cast<const float2 *>(refp + (((size_t)(nref == 1 ? 0 : n) * Lq + qq) * L + l) * 2);
      }
#pragma unroll
      for (int k = 0; k < LP / 4; ++k) rl[k] = Chan<T>::load4(lrow + 4 * k);
      if (pos_off != nullptr) {     // + projection of the query's position embedding (same for all n)
        const float *prow = pos_off + qq * pos_stride + (size_t)m * off_hs;
        const float *plrow = pos_logit + qq * pos_stride + (size_t)m * logit_hs;
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
      const T *lrow = loc_or_off + (((size_t)n * Lq + qq) * M + m) * (size_t)(LP * 2);
      const T *wrow = w_or_logit + (((size_t)n * Lq + qq) * M + m) * (size_t)LP;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        xy[l] = Chan<T>::load2(lrow + 2 * (l * P + p));
        aw[l] = Chan<T>::load(wrow + l * P + p);
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
    const T *base = tm.value_hm
                        ? value + (((size_t)m * gridDim.z + n) * S + (size_t)level_start[l]) * D
                        : value + (((size_t)n * S + (size_t)level_start[l]) * M + m) * D;
    rs[l] = dvis_make_rsrc_uniform(
        base, (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * (tm.value_hm ? D : MD) + D) * sizeof(T)));
  }

  const int lane = tid & 63, wv = tid >> 6;
  const int g = lane / G, j = lane - g * G;
  const unsigned lane_bytes = (unsigned)j * 16u;   // kOOB + lane_bytes is still out of range
  T *const out_frame = out + ((size_t)n * Lq * M + m) * D;   // uniform

  // Latency is hidden by WAVES, not by a deep per-wave pipeline: each wave keeps one batch of B samples
  // (4*B corner loads) in flight, reads that batch's taps from LDS just in time, and stays within the
  // register budget of WPS waves/SIMD.  (A fully unrolled 12-sample body makes hipcc hoist all 48 loads and
  // spill to scratch; measured 3-10x slower.)
#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    const int ql = (it * 4 + wv) * GPW + g;
    if (QB % (4 * GPW) != 0 && ql >= QB) break;          // wave-uniform: a wave's GPW pairs are one slot range
    const int q = slot_query(ql);
    float acc[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] = 0.f;
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
          float v1[CPL], v2[CPL], v3[CPL], v4[CPL];
          Chan<T>::unpack(r[4 * i], v1);
          Chan<T>::unpack(r[4 * i + 1], v2);
          Chan<T>::unpack(r[4 * i + 2], v3);
          Chan<T>::unpack(r[4 * i + 3], v4);
          const float c1 = c[i].x, c2 = c[i].y, c3 = c[i].z, c4 = c[i].w;
          // reference order: (w1 v1 + w2 v2 + w3 v3 + w4 v4) * weight, accumulated over samples
#pragma unroll
          for (int k = 0; k < CPL; ++k)
            acc[k] = dvis_msda::accumulate_sample(acc[k], c1, c2, c3, c4, v1[k], v2[k], v3[k], v4[k], aw[i]);
        }
      }
    }
    if (q >= 0) Chan<T>::store(out_frame + (size_t)q * MD + CPL * j, acc);
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

template <typename T, int D, int L, int P, bool FUSED, int WPS, int B, int QB>
int launch_variant(const T *value, const int64_t *shapes, const int64_t *ls, const T *a, int64_t a_stride,
                   const T *b, int64_t b_stride, const float *refp, int nref, int N, int S, int M, int Lq, T *out,
                   hipStream_t st, const float *pos_off, const float *pos_logit, int64_t pos_stride, const TileMap *tm) {
  if constexpr (QB == 64 && L <= kMaxLevels) {
    if (tm != nullptr && tm->tiled) {
      if (N > 65535) {
        dvis_set_error("msda: grid too large (N must be <= 65535)");
        return DVIS_E_ARG;
      }
      hipLaunchKernelGGL((msda_fwd_tile<T, D, L, P, FUSED, WPS, B, QB, true>), dim3(M, tm->tile_start[kMaxLevels], N),
                         dim3(256), 0, st, value, shapes, ls, a, a_stride, b, b_stride, refp, nref, S, M, Lq, out, pos_off,
                         pos_logit, pos_stride, *tm);
      return dvis_check_launch("msda_fwd_tile<2-D>");
    }
  }
  const int nchunks = (Lq + QB - 1) / QB;
  if (nchunks > 65535 || N > 65535) {
    dvis_set_error("msda: grid too large (Lq/%d and N must be <= 65535)", QB);
    return DVIS_E_ARG;
  }
  hipLaunchKernelGGL((msda_fwd_tile<T, D, L, P, FUSED, WPS, B, QB, false>), dim3(M, nchunks, N), dim3(256), 0, st, value,
                     shapes, ls, a, a_stride, b, b_stride, refp, nref, S, M, Lq, out, pos_off, pos_logit, pos_stride,
                     tm ? *tm : TileMap{});
  return dvis_check_launch("msda_fwd_tile");
}

template <typename T, int D, int L, int P, bool FUSED>
int launch_tile(const T *value, const int64_t *shapes, const int64_t *ls, const T *a, int64_t a_stride,
                const T *b, int64_t b_stride, const float *refp, int nref, int N, int S, int M, int Lq, T *out,
                hipStream_t st, const float *pos_off, const float *pos_logit, int64_t pos_stride, const TileMap *tm) {
#define DVIS_LAUNCH_VARIANT(wps, bsz, qb) \
  return launch_variant<T, D, L, P, FUSED, wps, bsz, qb>(value, shapes, ls, a, a_stride, b, b_stride, refp, nref, N, S, M, \
                                                         Lq, out, st, pos_off, pos_logit, pos_stride, tm)
  // queries covered by one pass of the 4 waves (fp32: 16 B = 4 channels per lane), never more than one set-up thread
  // per (query, point) allows
  constexpr int QMIN = 4 * (64 / (D / 4)) < 256 / P / 2 ? 4 * (64 / (D / 4)) : 256 / P / 2;
  // (min waves/SIMD the register allocator must allow) x (samples per load batch) x (queries per block).  Measured on
  // MI355X, 720p, 30 frames (round 1): (2, 2, 2 QMIN) 35.2 us per frame-layer = (6, 2, QMIN) 35.2 < (4, 4, QMIN) 37.0 —
  // occupancy-insensitive, the per-CU vector-memory path is the limit.  With 2 QMIN queries the taps take 27 KB of LDS.
  DVIS_LAUNCH_VARIANT(2, 2, 2 * QMIN);
#undef DVIS_LAUNCH_VARIANT
}

template <typename T, bool FUSED>
int dispatch_tile(int D, int L, int P, const T *value, const int64_t *shapes, const int64_t *ls, const T *a,
                  int64_t a_stride, const T *b, int64_t b_stride, const float *refp, int nref, int N, int S, int M,
                  int Lq, T *out, hipStream_t st, bool *handled, const float *pos_off = nullptr,
                  const float *pos_logit = nullptr, int64_t pos_stride = 0, const TileMap *tm = nullptr) {
  *handled = true;
#define DVIS_TILE_CASE(d, l, p)  \
  if (D == d && L == l && P == p) \
    return launch_tile<T, d, l, p, FUSED>(value, shapes, ls, a, a_stride, b, b_stride, refp, nref, N, S, M, Lq, out, st, \
                                          pos_off, pos_logit, pos_stride, tm);
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
      int rc = dispatch_tile<float, false>(D, L, P, (const float *)value, shapes, level_start, (const float *)loc, 0,
                                           (const float *)w, 0, nullptr, 0, N, S, M, Lq, (float *)out, st, &handled);
      if (handled) return rc;
    }
    return launch_generic<float>(value, shapes, level_start, loc, w, N, S, M, D, L, Lq, P, out, st);
  }
  if (dtype == DVIS_F64) return launch_generic<double>(value, shapes, level_start, loc, w, N, S, M, D, L, Lq, P, out, st);
  if (dtype == DVIS_F16 || dtype == DVIS_BF16) {
    // tiled path: 16 bytes = 8 channels per lane, fp32 accumulation (what the ViT-Adapter extractors need under autocast)
    const bool fits = (size_t)S * M * D * 2 < 0x7fffffffu && (size_t)Lq * M * L * P * 2 * 2 < 0x7fffffffu;
    if (fits && aligned16(value) && aligned16(out) && ((uintptr_t)loc & 3u) == 0 && (M * D) % 8 == 0) {
      bool handled = false;
      int rc = dtype == DVIS_F16
                   ? dispatch_tile<__half, false>(D, L, P, (const __half *)value, shapes, level_start, (const __half *)loc, 0,
                                                  (const __half *)w, 0, nullptr, 0, N, S, M, Lq, (__half *)out, st, &handled)
                   : dispatch_tile<__hip_bfloat16, false>(D, L, P, (const __hip_bfloat16 *)value, shapes, level_start,
                                                          (const __hip_bfloat16 *)loc, 0, (const __hip_bfloat16 *)w, 0,
                                                          nullptr, 0, N, S, M, Lq, (__hip_bfloat16 *)out, st, &handled);
      if (handled) return rc;
    }
    return dtype == DVIS_F16
               ? launch_generic<__half>(value, shapes, level_start, loc, w, N, S, M, D, L, Lq, P, out, st)
               : launch_generic<__hip_bfloat16>(value, shapes, level_start, loc, w, N, S, M, D, L, Lq, P, out, st);
  }
  dvis_set_error("msda_forward: unsupported dtype %d", dtype);
  return DVIS_E_ARG;
}

DVIS_EXPORT int dvis_msda_fused_forward_slots(const float *value, const int64_t *shapes, const int64_t *level_start,
                                              const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                              const float *logits, int64_t logit_stride, int off_head_stride,
                                              int logit_head_stride, int value_head_major, const float *pos_offsets,
                                              const float *pos_logits, int64_t pos_stride, int N, int S, int M, int D,
                                              int L, int Lq, int P, float *out, const int64_t *shapes_host,
                                              void *stream) {
  DVIS_REQUIRE(N >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_fused_forward: bad sizes");
  if (N == 0 || Lq == 0) return DVIS_OK;
  DVIS_REQUIRE(value && shapes && level_start && ref && offsets && logits && out, "msda_fused_forward: null pointer");
  DVIS_REQUIRE(Nref == 1 || Nref == N, "msda_fused_forward: Nref must be 1 or N");
  const int off_hs = off_head_stride ? off_head_stride : L * P * 2, logit_hs = logit_head_stride ? logit_head_stride : L * P;
  DVIS_REQUIRE(off_hs >= L * P * 2 && logit_hs >= L * P && off_hs % 4 == 0 && logit_hs % 4 == 0,
               "msda_fused_forward: head strides must cover a head's parameters and be multiples of 4 floats");
  DVIS_REQUIRE(off_stride >= (int64_t)(M - 1) * off_hs + L * P * 2 && logit_stride >= (int64_t)(M - 1) * logit_hs + L * P,
               "msda_fused_forward: row strides too small");
  DVIS_REQUIRE(off_stride % 4 == 0 && logit_stride % 4 == 0 && aligned16(offsets) && aligned16(logits) &&
                   aligned16(value) && aligned16(out),
               "msda_fused_forward: 16-byte alignment required");
  DVIS_REQUIRE((size_t)S * M * D * sizeof(float) < 0x7fffffffu, "msda_fused_forward: frame slice >= 2 GiB");
  bool handled = false;
  DVIS_REQUIRE((size_t)Lq * (size_t)(off_stride > logit_stride ? off_stride : logit_stride) * sizeof(float) < 0x7fffffffu,
               "msda_fused_forward: one frame of offsets/logits must stay below 2 GiB");
  // Encoder self-attention geometry (queries = the pixels of the L maps, shapes known on the host): 8 x 8 query tiles
  TileMap tm{};
  const bool has_pos = pos_offsets != nullptr || pos_logits != nullptr;
  if (!has_pos) make_tile_map(shapes_host, L, Lq, &tm);
  tm.off_hs = off_head_stride, tm.logit_hs = logit_head_stride;
  tm.value_hm = value_head_major ? 1 : 0;
  if (has_pos) {
    DVIS_REQUIRE(pos_offsets && pos_logits && pos_stride >= (int64_t)(M - 1) * off_hs + L * P * 2 && pos_stride % 4 == 0 &&
                     aligned16(pos_offsets) && aligned16(pos_logits),
                 "msda_fused_forward: position rows need both pointers, 16-byte alignment and a row stride multiple of 4");
    tm.tiled = 0;
  }
  const int rc = dispatch_tile<float, true>(D, L, P, value, shapes, level_start, offsets, off_stride, logits, logit_stride, ref,
                                            Nref, N, S, M, Lq, out, (hipStream_t)stream, &handled, pos_offsets, pos_logits,
                                            pos_stride, &tm);
  if (handled) return rc;
  dvis_set_error("msda_fused_forward: unsupported (D=%d, L=%d, P=%d); supported D in {32,64}, (L,P) in {(1,4),(3,4),(4,4)}",
                 D, L, P);
  return DVIS_E_UNSUPPORTED;
}

// The fused form on fp16 / bf16 storage: value, the raw offset / logit rows and the output in `dtype` (what nn.Linear hands
// over under autocast), reference points fp32, all arithmetic fp32.  The reference's row layout only (no slots / head-major).
template <typename T>
static int fused_half(const void *value, const int64_t *shapes, const int64_t *level_start, const float *ref, int Nref,
                      const void *offsets, int64_t off_stride, const void *logits, int64_t logit_stride, int N, int S, int M, int D,
                      int L, int Lq, int P, void *out, const int64_t *shapes_host, hipStream_t st, bool *handled) {
  TileMap tm{};
  make_tile_map(shapes_host, L, Lq, &tm);
  return dispatch_tile<T, true>(D, L, P, (const T *)value, shapes, level_start, (const T *)offsets, off_stride, (const T *)logits,
                                logit_stride, ref, Nref, N, S, M, Lq, (T *)out, st, handled, nullptr, nullptr, 0, &tm);
}

DVIS_EXPORT int dvis_msda_fused_forward_h(int dtype, const void *value, const int64_t *shapes, const int64_t *level_start,
                                          const float *ref, int Nref, const void *offsets, int64_t off_stride,
                                          const void *logits, int64_t logit_stride, int N, int S, int M, int D, int L, int Lq,
                                          int P, void *out, const int64_t *shapes_host, void *stream) {
  DVIS_REQUIRE(dtype == DVIS_F16 || dtype == DVIS_BF16, "msda_fused_forward_h: dtype must be fp16 or bf16 (fp32: dvis_msda_fused_forward)");
  DVIS_REQUIRE(N >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_fused_forward_h: bad sizes");
  if (N == 0 || Lq == 0) return DVIS_OK;
  DVIS_REQUIRE(value && shapes && level_start && ref && offsets && logits && out, "msda_fused_forward_h: null pointer");
  DVIS_REQUIRE(Nref == 1 || Nref == N, "msda_fused_forward_h: Nref must be 1 or N");
  DVIS_REQUIRE(off_stride >= (int64_t)M * L * P * 2 && logit_stride >= (int64_t)M * L * P && off_stride % 4 == 0 &&
                   logit_stride % 4 == 0 && (((uintptr_t)offsets | (uintptr_t)logits) & 7u) == 0 && aligned16(value) &&
                   aligned16(out) && (M * D) % 8 == 0 && (L * P) % 4 == 0,
               "msda_fused_forward_h: row strides must cover the rows and be multiples of 4 elements; 8-byte aligned rows, "
               "16-byte aligned value / out");
  DVIS_REQUIRE((size_t)S * M * D * 2 < 0x7fffffffu, "msda_fused_forward_h: frame slice >= 2 GiB");
  bool handled = false;
  const int rc = dtype == DVIS_F16
                     ? fused_half<__half>(value, shapes, level_start, ref, Nref, offsets, off_stride, logits, logit_stride, N, S, M,
                                          D, L, Lq, P, out, shapes_host, (hipStream_t)stream, &handled)
                     : fused_half<__hip_bfloat16>(value, shapes, level_start, ref, Nref, offsets, off_stride, logits, logit_stride,
                                                  N, S, M, D, L, Lq, P, out, shapes_host, (hipStream_t)stream, &handled);
  if (handled) return rc;
  dvis_set_error("msda_fused_forward_h: unsupported (D=%d, L=%d, P=%d); supported D in {32,64}, (L,P) in {(1,4),(3,4),(4,4)}", D, L, P);
  return DVIS_E_UNSUPPORTED;
}

DVIS_EXPORT int dvis_msda_fused_forward_pos(const float *value, const int64_t *shapes, const int64_t *level_start,
                                            const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                            const float *logits, int64_t logit_stride, const float *pos_offsets,
                                            const float *pos_logits, int64_t pos_stride, int N, int S, int M, int D,
                                            int L, int Lq, int P, float *out, const int64_t *shapes_host, void *stream) {
  return dvis_msda_fused_forward_slots(value, shapes, level_start, ref, Nref, offsets, off_stride, logits, logit_stride, 0, 0,
                                       0, pos_offsets, pos_logits, pos_stride, N, S, M, D, L, Lq, P, out, shapes_host, stream);
}

DVIS_EXPORT int dvis_msda_fused_forward(const float *value, const int64_t *shapes, const int64_t *level_start,
                                        const float *ref, int Nref, const float *offsets, int64_t off_stride,
                                        const float *logits, int64_t logit_stride, int N, int S, int M, int D, int L,
                                        int Lq, int P, float *out, const int64_t *shapes_host, void *stream) {
  return dvis_msda_fused_forward_pos(value, shapes, level_start, ref, Nref, offsets, off_stride, logits, logit_stride,
                                     nullptr, nullptr, 0, N, S, M, D, L, Lq, P, out, shapes_host, stream);
}
