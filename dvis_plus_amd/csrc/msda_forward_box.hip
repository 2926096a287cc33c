// MSDeformAttn forward for encoder self-attention, corners served from LDS — gfx950.
//
// Why: the tiled kernel (msda_forward.hip) pulls every corner of every sample through the per-CU vector L1 — one
// 128-byte line per (sample, head, corner), 949 MB per 720p frame-layer — and that 64 B/clk/CU path is its bound
// (24 us per frame-layer at best, 37 us measured).  In encoder self-attention the queries ARE the pixels of the L
// maps and the sampling offsets are a few pixels around the query's own position, so the samples of an 8x8 tile of
// queries fall into a small box of every level.  This kernel makes that explicit WITHOUT assuming it:
//   1. a workgroup owns one head and one 8x8 query tile of one level; it stages the tile's offsets / logits in LDS,
//      applies softmax and loc = ref + off / (W_l, H_l) there (ops/modules/ms_deform_attn.py:101-109);
//   2. it reduces, per level, the bounding box of all corners its samples touch (wave min/max + 4 LDS atomics);
//   3. level by level: if the box has at most CAP pixels, the box is copied once into LDS with coalesced 16-byte
//      buffer loads (one 128-byte line per pixel, each fetched ONCE instead of once per sample that touches it) and
//      the 4 corners of every sample are ds_read_b128 from LDS (128 B/clk/CU, no TA/L1 involvement); corners outside
//      the map read a zero row.  If the box does not fit (large learned offsets, coarse-level query tiles whose
//      footprint on the fine maps is big) that level falls back to the tiled kernel's global gather with hardware
//      zero padding — same arithmetic, so the result does not depend on which source served a corner.
// L1 traffic per 8x8 level-2 tile at the initial offsets (+-4 px): 17x17 + 13x13 + 11x11 = 579 lines staged instead
// of 64 x 12 x 4 = 3072 gathered.  Grid (M, tiles, N): head fastest => block b on XCD b % 8 == head.
//
// Fused interface only (raw offsets / logits + reference points), fp32, D = 32.
#include <limits.h>
#include <stdlib.h>

#include "dvis_common.h"
#include "msda_tap.h"

namespace {

using dvis_msda::kOOB;

constexpr int kTile = 64;     // 8x8 queries
constexpr int kCap = 320;     // pixels of one level box that fit the LDS stage (x 128 B = 40 KB)

struct BoxTiling {
  int tiles_cum[5];   // first tile index of each level (+ total)
  int tiles_x[4];     // tiles per row of each level
};

// Bilinear set-up shared by the LDS and the global source.  Identical arithmetic to dvis_msda::make_tap.
struct Corner {
  int h0, w0;
  bool ok, h0ok, h1ok, w0ok, w1ok;
  float c[4];
};

__device__ __forceinline__ Corner make_corner(float x, float y, int H, int W, bool active) {
  Corner t;
  const float h_im = y * (float)H - 0.5f;
  const float w_im = x * (float)W - 0.5f;
  t.ok = active && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  const float hf = floorf(h_im), wf = floorf(w_im);
  t.h0 = (int)hf;
  t.w0 = (int)wf;
  const float lh = h_im - hf, lw = w_im - wf;
  const float hh = 1.f - lh, hw = 1.f - lw;
  t.h0ok = t.ok && t.h0 >= 0;
  t.h1ok = t.ok && t.h0 + 1 <= H - 1;
  t.w0ok = t.w0 >= 0;
  t.w1ok = t.w0 + 1 <= W - 1;
  t.c[0] = t.ok ? hh * hw : 0.f;
  t.c[1] = t.ok ? hh * lw : 0.f;
  t.c[2] = t.ok ? lh * hw : 0.f;
  t.c[3] = t.ok ? lh * lw : 0.f;
  return t;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

template <int L, int P>
__global__ __launch_bounds__(256, 3) void msda_fwd_box_f32(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ off, int64_t off_stride, const float *__restrict__ logit, int64_t logit_stride,
    const float *__restrict__ refp, int nref, int S, int M, int Lq, BoxTiling tiling, float *__restrict__ out) {
  constexpr int D = 32, LP = L * P, G = 8, GPW = 8, LOCV = LP / 2, WV = LP / 4;
  constexpr int ITERS = kTile / (4 * GPW);   // query groups per wave
  constexpr int B = 2;                       // samples per batch of corner reads
  static_assert(LP % 4 == 0 && P == 4, "tile shape");

  __shared__ float4 s_loc[kTile * LOCV];
  __shared__ float4 s_w[kTile * WV];
  __shared__ float4 s_val[(kCap + 1) * G];   // staged box, pixel-major; pixel kCap is the zero row
  __shared__ int s_box[L * 4];               // per level: min x, max x, min y, max y of the touched corners

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

  // ---- this block's 8x8 query tile (all wave-uniform)
  int ql_lvl = 0;
#pragma unroll
  for (int ll = 1; ll < L; ++ll)
    if ((int)blockIdx.y >= tiling.tiles_cum[ll]) ql_lvl = ll;
  const int t_idx = blockIdx.y - tiling.tiles_cum[ql_lvl];
  const int tl_y0 = (t_idx / tiling.tiles_x[ql_lvl]) * 8;
  const int tl_x0 = (t_idx % tiling.tiles_x[ql_lvl]) * 8;
  int tl_h = Hs[0], tl_w = Ws[0], tl_base = (int)level_start[0];
#pragma unroll
  for (int ll = 1; ll < L; ++ll)
    if (ql_lvl == ll) { tl_h = Hs[ll]; tl_w = Ws[ll]; tl_base = (int)level_start[ll]; }
  auto slot_query = [&](int ql) -> int {
    const int y = tl_y0 + (ql >> 3), x = tl_x0 + (ql & 7);
    return (y < tl_h && x < tl_w) ? tl_base + y * tl_w + x : -1;
  };

  // ---- stage raw offsets / logits of the tile (one head) into LDS; empty slots read 0 through the descriptor
  {
    const size_t row0 = (size_t)n * Lq;
    const unsigned lrow = (unsigned)((size_t)off_stride * sizeof(float));
    const unsigned wrow = (unsigned)((size_t)logit_stride * sizeof(float));
    const __amdgpu_buffer_rsrc_t lrs =
        dvis_make_rsrc_uniform(off + row0 * off_stride + (size_t)m * (LP * 2), (unsigned)(Lq - 1) * lrow + LP * 2 * 4);
    const __amdgpu_buffer_rsrc_t wrs =
        dvis_make_rsrc_uniform(logit + row0 * logit_stride + (size_t)m * LP, (unsigned)(Lq - 1) * wrow + LP * 4);
    for (int i = tid; i < kTile * LOCV; i += 256) {
      const int ql = i / LOCV, k = i - ql * LOCV;
      const int q = slot_query(ql);
      const dvis_v4u v =
          __builtin_amdgcn_raw_buffer_load_b128(lrs, q >= 0 ? (unsigned)q * lrow + (unsigned)k * 16u : kOOB, 0, 0);
      s_loc[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    for (int i = tid; i < kTile * WV; i += 256) {
      const int ql = i / WV, k = i - ql * WV;
      const int q = slot_query(ql);
      const dvis_v4u v =
          __builtin_amdgcn_raw_buffer_load_b128(wrs, q >= 0 ? (unsigned)q * wrow + (unsigned)k * 16u : kOOB, 0, 0);
      s_w[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    if (tid < L * 4) s_box[tid] = (tid & 1) ? INT_MIN : INT_MAX;
    if (tid < G) s_val[kCap * G + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  float *lf = reinterpret_cast<float *>(s_loc);
  float *wf = reinterpret_cast<float *>(s_w);
  // loc = ref + off / (W_l, H_l)
  for (int i = tid; i < kTile * LP; i += 256) {
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
  if (tid < kTile) {
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

  // ---- per level: bounding box of the corners the tile's samples touch.  Thread = (query tid>>2, point tid&3).
  {
    const int ql = tid >> 2, p = tid & 3;
    const bool active = slot_query(ql) >= 0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float2 xy = *reinterpret_cast<const float2 *>(lf + ql * (LP * 2) + 2 * (l * P + p));
      const Corner t = make_corner(xy.x, xy.y, Hs[l], Ws[l], active);
      int x0 = INT_MAX, x1 = INT_MIN, y0 = INT_MAX, y1 = INT_MIN;
      if (t.ok) {
        x0 = max(t.w0, 0); x1 = min(t.w0 + 1, Ws[l] - 1);
        y0 = max(t.h0, 0); y1 = min(t.h0 + 1, Hs[l] - 1);
      }
      x0 = wave_min(x0); x1 = wave_max(x1); y0 = wave_min(y0); y1 = wave_max(y1);
      if ((tid & 63) == 0) {
        atomicMin(&s_box[4 * l], x0);
        atomicMax(&s_box[4 * l + 1], x1);
        atomicMin(&s_box[4 * l + 2], y0);
        atomicMax(&s_box[4 * l + 3], y1);
      }
    }
  }
  __syncthreads();

  // ---- per-level descriptors over this (frame, head) slice of `value`
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
  const char *val_bytes = reinterpret_cast<const char *>(s_val);

  float acc[ITERS][4];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) acc[it][0] = acc[it][1] = acc[it][2] = acc[it][3] = 0.f;

#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = Hs[l], W = Ws[l];
    const int bx0 = __builtin_amdgcn_readfirstlane(s_box[4 * l]);
    const int bx1 = __builtin_amdgcn_readfirstlane(s_box[4 * l + 1]);
    const int by0 = __builtin_amdgcn_readfirstlane(s_box[4 * l + 2]);
    const int by1 = __builtin_amdgcn_readfirstlane(s_box[4 * l + 3]);
    const bool any = bx1 >= bx0 && by1 >= by0;
    const int bw = any ? bx1 - bx0 + 1 : 1;
    const int npx = any ? bw * (by1 - by0 + 1) : 0;
    const bool in_lds = npx <= kCap;          // wave-uniform (and block-uniform)

    if (in_lds) {
      if (l > 0) __syncthreads();             // everyone is done reading the previous level's box
      // pi / bw by multiply-shift: exact for pi * bw < 2^20 (pi < kCap <= 320, bw <= 320)
      const unsigned magic = ((1u << 20) + (unsigned)bw - 1u) / (unsigned)bw;
      const unsigned org = (unsigned)(by0 * W + bx0) * pix_bytes;
      // every load of the box is issued before the first LDS store: ONE global round trip per level, not one per
      // 256 x 16 B (slots past the box get an out-of-range offset: the hardware returns 0 and nothing is stored)
      constexpr int NST = (kCap * G + 255) / 256;
      dvis_v4u st[NST];
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int i = tid + 256 * k;
        const unsigned pi = (unsigned)i >> 3, jj = (unsigned)i & 7u;
        const unsigned py = (pi * magic) >> 20;
        const unsigned px = pi - py * (unsigned)bw;
        st[k] = __builtin_amdgcn_raw_buffer_load_b128(
            rs[l], i < npx * G ? org + (py * (unsigned)W + px) * pix_bytes + jj * 16u : kOOB, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < NST; ++k) {
        const int i = tid + 256 * k;
        if (i < npx * G)
          s_val[i] = make_float4(__uint_as_float(st[k].x), __uint_as_float(st[k].y), __uint_as_float(st[k].z),
                                 __uint_as_float(st[k].w));
      }
      __syncthreads();
    }

#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int ql = (it * 4 + wv) * GPW + g;
      const bool active = slot_query(ql) >= 0;
      // batches of B samples = 4*B corner reads in flight; rolled so that hipcc does not hoist every load of the level
#pragma unroll 1
      for (int pb = 0; pb < P / B; ++pb) {
        const int s0 = l * P + pb * B;
        float xy[2 * B], aw[B];
#pragma unroll
        for (int i = 0; i < B / 2; ++i) {
          const float4 v = *reinterpret_cast<const float4 *>(lf + ql * (LP * 2) + 2 * s0 + 4 * i);
          xy[4 * i] = v.x; xy[4 * i + 1] = v.y; xy[4 * i + 2] = v.z; xy[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < B / 2; ++i) {
          const float2 v = *reinterpret_cast<const float2 *>(wf + ql * LP + s0 + 2 * i);
          aw[2 * i] = v.x; aw[2 * i + 1] = v.y;
        }
        float4 r[4 * B];
        float cw[4 * B];
#pragma unroll
        for (int p = 0; p < B; ++p) {
          const Corner t = make_corner(xy[2 * p], xy[2 * p + 1], H, W, active);
#pragma unroll
          for (int c = 0; c < 4; ++c) cw[4 * p + c] = t.c[c];
          const bool k00 = t.h0ok && t.w0ok, k01 = t.h0ok && t.w1ok, k10 = t.h1ok && t.w0ok, k11 = t.h1ok && t.w1ok;
          if (in_lds) {
            const unsigned zero = (unsigned)kCap * 128u + lane_bytes;
            const unsigned o00 = (unsigned)((t.h0 - by0) * bw + (t.w0 - bx0)) * 128u + lane_bytes;
            const unsigned row = (unsigned)bw * 128u;
            r[4 * p] = *reinterpret_cast<const float4 *>(val_bytes + (k00 ? o00 : zero));
            r[4 * p + 1] = *reinterpret_cast<const float4 *>(val_bytes + (k01 ? o00 + 128u : zero));
            r[4 * p + 2] = *reinterpret_cast<const float4 *>(val_bytes + (k10 ? o00 + row : zero));
            r[4 * p + 3] = *reinterpret_cast<const float4 *>(val_bytes + (k11 ? o00 + row + 128u : zero));
          } else {
            const unsigned o00 = (unsigned)(t.h0 * W + t.w0) * pix_bytes + lane_bytes;
            const unsigned o[4] = {k00 ? o00 : kOOB, k01 ? o00 + pix_bytes : kOOB,
                                   k10 ? o00 + (unsigned)W * pix_bytes : kOOB,
                                   k11 ? o00 + (unsigned)W * pix_bytes + pix_bytes : kOOB};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const dvis_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[c], 0, 0);
              r[4 * p + c] =
                  make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
          }
        }
#pragma unroll
        for (int p = 0; p < B; ++p) {
          const float4 r1 = r[4 * p], r2 = r[4 * p + 1], r3 = r[4 * p + 2], r4 = r[4 * p + 3];
          const float c1 = cw[4 * p], c2 = cw[4 * p + 1], c3 = cw[4 * p + 2], c4 = cw[4 * p + 3];
          // reference order: (w1 v1 + w2 v2 + w3 v3 + w4 v4) * weight, accumulated over samples
          acc[it][0] += (c1 * r1.x + c2 * r2.x + c3 * r3.x + c4 * r4.x) * aw[p];
          acc[it][1] += (c1 * r1.y + c2 * r2.y + c3 * r3.y + c4 * r4.y) * aw[p];
          acc[it][2] += (c1 * r1.z + c2 * r2.z + c3 * r3.z + c4 * r4.z) * aw[p];
          acc[it][3] += (c1 * r1.w + c2 * r2.w + c3 * r3.w + c4 * r4.w) * aw[p];
        }
      }
    }
  }

  float *const out_frame = out + ((size_t)n * Lq * M + m) * D;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int q = slot_query((it * 4 + wv) * GPW + g);
    if (q >= 0)
      *reinterpret_cast<float4 *>(out_frame + (size_t)q * MD + 4 * j) =
          make_float4(acc[it][0], acc[it][1], acc[it][2], acc[it][3]);
  }
}

// OFF by default: measured on MI355X (30 frames / launch, init-rule offsets, every level-2 tile's boxes fit):
// 55.3 us per frame-layer vs 35.2 us for the tiled kernel.  PMC: LDS only 11 % busy, but 6.3e8 VALU instructions per
// launch (61 % VALU utilisation: box reduction, staging addresses, per-lane taps) and — the real limit — five
// serialised global round trips per workgroup (offsets, reference points, three box stages) with only 3 workgroups
// per CU (50 KB of LDS each) to overlap them.  DVIS_MSDA_BOX=1 enables it for experiments.
bool box_enabled() {
  static const bool v = [] {
    const char *e = getenv("DVIS_MSDA_BOX");
    return e != nullptr && atoi(e) != 0;
  }();
  return v;
}

}  // namespace

// Internal (not exported): called by dvis_msda_fused_forward.  *handled = false when the shape is not this kernel's.
int dvis_msda_box_launch(const float *value, const int64_t *shapes, const int64_t *level_start, const float *ref, int nref,
                         const float *offsets, int64_t off_stride, const float *logits, int64_t logit_stride, int N, int S,
                         int M, int D, int L, int Lq, int P, float *out, const int64_t *shapes_host, hipStream_t st,
                         bool *handled) {
  *handled = false;
  if (!box_enabled() || shapes_host == nullptr || D != 32 || L != 3 || P != 4) return DVIS_OK;
  BoxTiling tiling = {};
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
  // only when the queries are exactly the pixels of the maps (encoder self-attention)
  if (total != Lq || total != S || cum > 65535 || N > 65535) return DVIS_OK;
  *handled = true;
  hipLaunchKernelGGL((msda_fwd_box_f32<3, 4>), dim3(M, cum, N), dim3(256), 0, st, value, shapes, level_start, offsets,
                     off_stride, logits, logit_stride, ref, nref, S, M, Lq, tiling, out);
  return dvis_check_launch("msda_fwd_box_f32");
}
