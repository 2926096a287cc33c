// Multi-scale deformable attention forward for the deformable ENCODER's self-attention geometry — gfx950.
//
// Same op, same bits as msda_forward.hip's fused tile kernel (ops/modules/ms_deform_attn.py:101-117 +
// ms_deform_im2col_cuda.cuh:242-304); a different schedule, used when the queries ARE the pixels of the L value maps
// (Lq == S, query q of level lq = pixel (y, x) of that map: MSDeformAttnTransformerEncoder, msdeformattn.py:61-89) and
// the host passes the map shapes:
//
//   * a workgroup owns ONE head and an 8 x 8 TILE of query pixels of one level (not 64 consecutive pixels of a row):
//     neighbouring queries sample neighbouring locations, so inside a tile every gathered 128-byte corner line is
//     reused by up to four queries through the CU's L1 instead of two;
//   * the samples that fall into a COARSER level than the query's (2 of 3 levels for the finest queries = 76 % of all
//     queries) are served from LDS: the tile's footprint in such a level is tiny (8 x 8 queries cover 4 x 4 resp. 2 x 2
//     pixels, plus a halo of the largest sampling offset), so a 14 x 14 resp. 12 x 12 pixel box of the head's value slice
//     is staged once per tile (zero-filled outside the map) and the 4 corners of those samples come from ds_read_b128 at
//     4 x the L1 rate.  Measured on MI355X (tools/exp/msda_probe): the corner gathers of the two coarse levels cost
//     16.6 of the tile kernel's 36.4 us per 720p frame-layer and already run at the L1's peak rate (38 TB/s), so only
//     another data path can make them cheaper;
//   * samples of the query's own (or a finer) level keep the global path, and so does a whole level of a tile as soon
//     as ONE of its samples leaves the box (a learned offset larger than the halo): correctness never depends on the box.
// Arithmetic, order of accumulation and zero padding are those of the tile kernel: results are bit-identical.
#include "dvis_common.h"
#include "msda_tap.h"

namespace {

using dvis_msda::kOOB;
using dvis_msda::make_tap;
using dvis_msda::Tap;

constexpr int kTile = 8;            // 8 x 8 query pixels per workgroup
constexpr int kQB = kTile * kTile;
constexpr int kHalo = 4;            // largest |offset| (in pixels of the sampled level) a box covers
constexpr int kBoxA = 14;           // box edge for the level one step coarser than the query's:  ceil(7 / 2) + 2 * 4 + 2
constexpr int kBoxB = 12;           // ... two steps coarser:                                       ceil(7 / 4) + 2 * 4 + 2
constexpr int kMaxL = 4;

struct TileMap {
  int tile_start[kMaxL + 1];        // first tile index of every level (prefix sums), [L] = total
  int tiles_x[kMaxL];               // tiles per row of every level
};

template <int L, int P, int D, bool BOXES>
__global__ __launch_bounds__(256, 2) void msda_fwd_tile2d_f32(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ off, int64_t off_stride, const float *__restrict__ logit, int64_t logit_stride,
    const float *__restrict__ refp, int nref, int S, int M, int Lq, float *__restrict__ out, TileMap tm) {
  constexpr int LP = L * P;
  constexpr int G = D / 4;          // lanes per (query, head) pair
  constexpr int GPW = 64 / G;       // pairs per wave-instruction
  constexpr int ITERS = kQB / (4 * GPW);
  constexpr int PIX = D * 4;        // bytes of one head's slice of a pixel
  static_assert(D == 32 && P == 4 && kQB * P == 256 && L <= kMaxL, "tile shape");

  __shared__ uint4 s_tap_o[kQB * LP];                  // 4 corner byte offsets in the level's global slice (kOOB = zero)
  __shared__ float4 s_tap_c[kQB * LP];                 // 4 corner weights
  __shared__ float s_aw[kQB * LP];                     // attention weights
  __shared__ unsigned s_lds_o[BOXES ? kQB * 2 * P : 1];   // box byte offset of corner (y0, x0) for the two staged levels
  __shared__ __attribute__((aligned(16))) float s_box[BOXES ? (kBoxA * kBoxA + kBoxB * kBoxB) * D : 4];
  __shared__ unsigned s_miss[4];                       // per wave: bit k = a sample of slot k's level left its box

  const int tid = threadIdx.x;
  const int m = blockIdx.x;         // fastest grid dimension: head m on XCD m % 8 (its value slice stays in that L2)
  const int n = blockIdx.z;
  const int MD = M * D;
  const unsigned pix_bytes = (unsigned)MD * 4u;

  int Hs[L], Ws[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
  }
  // ---- which tile: level lq, tile (ty, tx)  (wave-uniform: blockIdx.y and kernel arguments are SGPRs)
  const int tile = blockIdx.y;
  int lq = 0;
#pragma unroll
  for (int l = 1; l < L; ++l) lq = tile >= tm.tile_start[l] ? l : lq;
  int Hq = Hs[0], Wq = Ws[0], q_base = 0, txs = tm.tiles_x[0], t0 = tm.tile_start[0];
#pragma unroll
  for (int l = 1; l < L; ++l)
    if (lq == l) { Hq = Hs[l]; Wq = Ws[l]; q_base = (int)level_start[l]; txs = tm.tiles_x[l]; t0 = tm.tile_start[l]; }
  const int trel = tile - t0;
  const int ty = trel / txs, tx = trel - ty * txs;

  // ---- boxes: level lq - 1 -> slot A, level lq - 2 -> slot B.  Origin = floor(coordinate of the tile's first query
  // in that level - halo); the box must hold corner x0 + 1 of the last query + halo, else the level stays global.
  struct Box { int l, y0, x0, edge, lds; bool on; };
  Box box[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    Box &b = box[k];
    b.l = lq - 1 - k;
    b.edge = k == 0 ? kBoxA : kBoxB;
    b.lds = k == 0 ? 0 : kBoxA * kBoxA * PIX;
    b.on = BOXES && b.l >= 0;
    b.y0 = b.x0 = 0;
    if (b.on) {
      int Hl = Hs[0], Wl = Ws[0];
#pragma unroll
      for (int l = 1; l < L; ++l)
        if (b.l == l) { Hl = Hs[l]; Wl = Ws[l]; }
      // centre of query pixel x in level l: (x + 0.5) / Wq * Wl - 0.5
      const float xlo = ((float)(tx * kTile) + 0.5f) / (float)Wq * (float)Wl - 0.5f;
      const float xhi = ((float)(tx * kTile + kTile - 1) + 0.5f) / (float)Wq * (float)Wl - 0.5f;
      const float ylo = ((float)(ty * kTile) + 0.5f) / (float)Hq * (float)Hl - 0.5f;
      const float yhi = ((float)(ty * kTile + kTile - 1) + 0.5f) / (float)Hq * (float)Hl - 0.5f;
      b.x0 = (int)floorf(xlo - (float)kHalo);
      b.y0 = (int)floorf(ylo - (float)kHalo);
      const int x1 = (int)floorf(xhi + (float)kHalo) + 1, y1 = (int)floorf(yhi + (float)kHalo) + 1;
      b.on = x1 - b.x0 < b.edge && y1 - b.y0 < b.edge;
    }
    b.on = __builtin_amdgcn_readfirstlane((int)b.on) != 0;
    b.x0 = __builtin_amdgcn_readfirstlane(b.x0);
    b.y0 = __builtin_amdgcn_readfirstlane(b.y0);
  }

  // ---- stage the boxes: 16-byte chunks, 8 per pixel, through registers (a buffer load outside the map returns 0, so
  // the zero padding of the reference is IN the box).  Issued first: the addresses depend on the tile only.
  constexpr int kChunksA = kBoxA * kBoxA * G, kChunksB = kBoxB * kBoxB * G;
  constexpr int kStageA = (kChunksA + 255) / 256, kStageB = (kChunksB + 255) / 256;
  dvis_v4u stagedA[kStageA], stagedB[kStageB];
  {
    // one descriptor over the whole (frame, head) slice: the pixel index carries the level's start
    const __amdgpu_buffer_rsrc_t brs =
        dvis_make_rsrc_uniform(value + ((size_t)n * S * M + m) * D, (unsigned)(((size_t)(S - 1) * MD + D) * sizeof(float)));
    auto stage = [&](const Box &b, int c, int nchunks) -> dvis_v4u {
      int Hl = 0, Wl = 0, ls = 0;
#pragma unroll
      for (int l = 0; l < L; ++l)
        if (b.l == l) { Hl = Hs[l]; Wl = Ws[l]; ls = (int)level_start[l]; }
      const int pix = c / G, j = c - pix * G;
      const int by = pix / b.edge, bx = pix - by * b.edge;
      const int y = b.y0 + by, x = b.x0 + bx;
      const bool ok = c < nchunks && y >= 0 && y < Hl && x >= 0 && x < Wl;
      const unsigned o = ok ? (unsigned)(ls + y * Wl + x) * pix_bytes + (unsigned)j * 16u : kOOB;
      return __builtin_amdgcn_raw_buffer_load_b128(brs, o, 0, 0);
    };
    if (box[0].on) {                                   // wave-uniform
#pragma unroll
      for (int i = 0; i < kStageA; ++i) stagedA[i] = stage(box[0], tid + i * 256, kChunksA);
    }
    if (box[1].on) {
#pragma unroll
      for (int i = 0; i < kStageB; ++i) stagedB[i] = stage(box[1], tid + i * 256, kChunksB);
    }
  }

  // ---- set-up: thread (query slot tid / P, point tid % P) — as in the tile kernel, plus the box offsets
  const int ql_s = tid / P, p = tid - ql_s * P;
  const int sy = ty * kTile + (ql_s >> 3), sx = tx * kTile + (ql_s & 7);
  const bool active = sy < Hq && sx < Wq;
  const size_t qq = active ? (size_t)(q_base + sy * Wq + sx) : 0;
  unsigned miss = 0;                                   // bit k: a sample of slot k's level is outside its box
  {
    const float *orow = off + ((size_t)n * Lq + qq) * off_stride + (size_t)m * (LP * 2);
    const float *lrow = logit + ((size_t)n * Lq + qq) * logit_stride + (size_t)m * LP;
    float2 ro[L], rr[L];
    float4 rl[LP / 4];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      ro[l] = *reinterpret_cast<const float2 *>(orow + 2 * (l * P + p));
      rr[l] = *reinterpret_cast<const float2 *>(refp + (((size_t)(nref == 1 ? 0 : n) * Lq + qq) * L + l) * 2);
    }
#pragma unroll
    for (int k = 0; k < LP / 4; ++k) rl[k] = *reinterpret_cast<const float4 *>(lrow + 4 * k);
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
      const float x = rr[l].x + ro[l].x / (float)Ws[l];
      const float y = rr[l].y + ro[l].y / (float)Hs[l];
      float ev = e[l * P];
#pragma unroll
      for (int pp = 1; pp < P; ++pp) ev = (p == pp) ? e[l * P + pp] : ev;
      const Tap t = make_tap(x, y, Hs[l], Ws[l], active, pix_bytes, 0u);
      const int si = ql_s * LP + l * P + p;
      s_tap_o[si] = make_uint4(t.o[0], t.o[1], t.o[2], t.o[3]);
      s_tap_c[si] = make_float4(t.c[0], t.c[1], t.c[2], t.c[3]);
      s_aw[si] = ev / sum;
      // box offset of corner (y0, x0): the same floor as make_tap's; a sample that is not counted (all weights 0) may
      // read anywhere inside the box
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (box[k].on && box[k].l == l) {
          const float h_im = y * (float)Hs[l] - 0.5f, w_im = x * (float)Ws[l] - 0.5f;
          const bool counted = active && h_im > -1.f && w_im > -1.f && h_im < (float)Hs[l] && w_im < (float)Ws[l];
          int by = (int)floorf(h_im) - box[k].y0, bx = (int)floorf(w_im) - box[k].x0;
          const bool inside = by >= 0 && bx >= 0 && by + 1 < box[k].edge && bx + 1 < box[k].edge;
          if (counted && !inside) miss |= 1u << k;
          if (!counted || !inside) by = bx = 0;
          s_lds_o[(ql_s * 2 + k) * P + p] = (unsigned)(box[k].lds + (by * box[k].edge + bx) * PIX);
        }
      }
    }
  }
  // ---- boxes into LDS
  if (box[0].on) {
#pragma unroll
    for (int i = 0; i < kStageA; ++i) {
      const int c = tid + i * 256;
      if (c < kChunksA) *reinterpret_cast<dvis_v4u *>(reinterpret_cast<char *>(s_box) + (size_t)c * 16) = stagedA[i];
    }
  }
  if (box[1].on) {
#pragma unroll
    for (int i = 0; i < kStageB; ++i) {
      const int c = tid + i * 256;
      if (c < kChunksB)
        *reinterpret_cast<dvis_v4u *>(reinterpret_cast<char *>(s_box) + (size_t)(kChunksA + c) * 16) = stagedB[i];
    }
  }
  {
    const unsigned wm = (__ballot(miss & 1u) ? 1u : 0u) | (__ballot(miss & 2u) ? 2u : 0u);
    if ((tid & 63) == 0) s_miss[tid >> 6] = wm;
  }
  __syncthreads();                                     // the one barrier of the kernel
  const unsigned missed = s_miss[0] | s_miss[1] | s_miss[2] | s_miss[3];
  const bool use_box[2] = {__builtin_amdgcn_readfirstlane((int)(box[0].on && !(missed & 1u))) != 0,
                           __builtin_amdgcn_readfirstlane((int)(box[1].on && !(missed & 2u))) != 0};

  // ---- per-level buffer descriptors over this (frame, head) slice of `value`
  __amdgpu_buffer_rsrc_t rs[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float *base = value + (((size_t)n * S + (size_t)level_start[l]) * M + m) * D;
    rs[l] = dvis_make_rsrc_uniform(base, (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * MD + D) * sizeof(float)));
  }
  const int lane = tid & 63, wv = tid >> 6;
  const int g = lane / G, j = lane - g * G;
  const unsigned lane_bytes = (unsigned)j * 16u;
  float *const out_frame = out + ((size_t)n * Lq * M + m) * D;
  const char *const boxp = reinterpret_cast<const char *>(s_box);

#pragma unroll 1
  for (int it = 0; it < ITERS; ++it) {
    const int ql = (it * 4 + wv) * GPW + g;
    const int qy = ty * kTile + (ql >> 3), qx = tx * kTile + (ql & 7);
    const bool q_ok = qy < Hq && qx < Wq;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto accumulate = [&](const dvis_v4u &r1, const dvis_v4u &r2, const dvis_v4u &r3, const dvis_v4u &r4, const float4 &c,
                          float aw) {
      const float f1[4] = {__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w)};
      const float f2[4] = {__uint_as_float(r2.x), __uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w)};
      const float f3[4] = {__uint_as_float(r3.x), __uint_as_float(r3.y), __uint_as_float(r3.z), __uint_as_float(r3.w)};
      const float f4[4] = {__uint_as_float(r4.x), __uint_as_float(r4.y), __uint_as_float(r4.z), __uint_as_float(r4.w)};
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
        acc[ch] = dvis_msda::accumulate_sample(acc[ch], c.x, c.y, c.z, c.w, f1[ch], f2[ch], f3[ch], f4[ch], aw);
    };
#pragma unroll
    for (int l = 0; l < L; ++l) {
      // which box slot serves level l for this tile (wave-uniform): lq - 1 -> slot 0, lq - 2 -> slot 1
      const int k = lq - 1 - l;
      const bool from_lds = (k == 0 && use_box[0]) || (k == 1 && use_box[1]);
      if (from_lds) {
        const int edge = k == 0 ? kBoxA : kBoxB;
        const unsigned row = (unsigned)(edge * PIX);
#pragma unroll
        for (int pp = 0; pp < P; pp += 2) {                       // 2 samples = 8 ds_read_b128 in flight
          dvis_v4u r[8];
          float4 c[2];
          float aw[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const unsigned b0 = s_lds_o[(ql * 2 + k) * P + pp + i] + lane_bytes;
            r[4 * i] = *reinterpret_cast<const dvis_v4u *>(boxp + b0);
            r[4 * i + 1] = *reinterpret_cast<const dvis_v4u *>(boxp + b0 + PIX);
            r[4 * i + 2] = *reinterpret_cast<const dvis_v4u *>(boxp + b0 + row);
            r[4 * i + 3] = *reinterpret_cast<const dvis_v4u *>(boxp + b0 + row + PIX);
            c[i] = s_tap_c[ql * LP + l * P + pp + i];
            aw[i] = s_aw[ql * LP + l * P + pp + i];
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) accumulate(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], c[i], aw[i]);
        }
      } else {
#pragma unroll 1
        for (int pp = 0; pp < P; pp += 2) {                       // 2 samples = 8 corner loads in flight per wave
          dvis_v4u r[8];
          float4 c[2];
          float aw[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const uint4 o = s_tap_o[ql * LP + l * P + pp + i];
            r[4 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o.x + lane_bytes, 0, 0);
            r[4 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o.y + lane_bytes, 0, 0);
            r[4 * i + 2] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o.z + lane_bytes, 0, 0);
            r[4 * i + 3] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o.w + lane_bytes, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            c[i] = s_tap_c[ql * LP + l * P + pp + i];
            aw[i] = s_aw[ql * LP + l * P + pp + i];
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) accumulate(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], c[i], aw[i]);
        }
      }
    }
    if (q_ok) {
      float *dst = out_frame + (size_t)(q_base + qy * Wq + qx) * MD + 4 * j;
      *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

}  // namespace

// Host side: returns DVIS_OK with *handled = false when the geometry is not the encoder's (caller takes the tile kernel).
int dvis_msda_tile2d_launch(const float *value, const int64_t *shapes, const int64_t *level_start, const float *ref,
                            int nref, const float *offsets, int64_t off_stride, const float *logits, int64_t logit_stride,
                            int N, int S, int M, int D, int L, int Lq, int P, float *out, const int64_t *shapes_host,
                            hipStream_t st, bool *handled, int boxes) {
  *handled = false;
  if (shapes_host == nullptr || D != 32 || P != 4 || L != 3 || Lq != S || N > 65535) return DVIS_OK;
  TileMap tm;
  long total = 0;
  int tiles = 0;
  for (int l = 0; l < L; ++l) {
    const long H = shapes_host[2 * l], W = shapes_host[2 * l + 1];
    if (H <= 0 || W <= 0) return DVIS_OK;
    // the staged levels must be coarser: maps ordered coarse -> fine like the pixel decoder's (msdeformattn.py:319-322)
    if (l > 0 && (H < shapes_host[2 * l - 2] || W < shapes_host[2 * l - 1])) return DVIS_OK;
    total += H * W;
    tm.tile_start[l] = tiles;
    tm.tiles_x[l] = (int)((W + kTile - 1) / kTile);
    tiles += tm.tiles_x[l] * (int)((H + kTile - 1) / kTile);
  }
  for (int l = L; l <= kMaxL; ++l) tm.tile_start[l] = tiles;
  for (int l = L; l < kMaxL; ++l) tm.tiles_x[l] = 1;
  if (total != Lq || tiles > 65535) return DVIS_OK;
  *handled = true;
  if (boxes)
    hipLaunchKernelGGL((msda_fwd_tile2d_f32<3, 4, 32, true>), dim3(M, tiles, N), dim3(256), 0, st, value, shapes, level_start,
                       offsets, off_stride, logits, logit_stride, ref, nref, S, M, Lq, out, tm);
  else
    hipLaunchKernelGGL((msda_fwd_tile2d_f32<3, 4, 32, false>), dim3(M, tiles, N), dim3(256), 0, st, value, shapes, level_start,
                       offsets, off_stride, logits, logit_stride, ref, nref, S, M, Lq, out, tm);
  return dvis_check_launch("msda_fwd_tile2d_f32");
}
