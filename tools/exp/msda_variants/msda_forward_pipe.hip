// MSDeformAttn forward for encoder self-attention: persistent, software-pipelined, corners served from LDS — gfx950.
//
// Why: the tiled kernel (msda_forward.hip) is bound by the texture-address / L1 pipe (TA busy 93 %): every corner of
// every sample is a 128-byte line = two 64-byte L1 accesses, 14.8 M accesses per 720p frame-layer.  The samples of an
// 8x8 tile of queries fall into a small box of each level, so staging that box ONCE into LDS and gathering the corners
// with ds_read_b128 cuts the L1 accesses ~5x.  A first version of that idea (msda_forward_box.hip) lost because each
// workgroup serialised five global round trips with only three workgroups per CU to overlap them.  Here ONE persistent
// workgroup per CU walks its share of (frame, tile) items and pipelines them in software:
//
//   step (tile i, level l):   A  issue the global loads of the NEXT step's box into registers
//                             A' (l == 0) issue the loads of tile i+1's offsets / logits into registers
//                             B  gather this step's corners from LDS buffer [s & 1] (global fallback if the box was too
//                                big to stage) and accumulate
//                             C  store the staged registers into LDS buffer [(s + 1) & 1]
//                             D  (l == 1) softmax + sampling locations + bilinear taps + per-level bounding boxes of
//                                tile i+1, from the registers loaded in A'
//                             one __syncthreads()
//
// so every global latency is covered by an LDS gather and there is one barrier per step.  Same arithmetic as the tiled
// kernel (same tap set-up, same accumulation order over levels / points / corners).
// Fused interface only (raw offsets / logits + reference points), fp32, D = 32, (L, P) = (3, 4), queries = pixels.
#include <limits.h>
#include <stdlib.h>

#include "dvis_common.h"
#include "msda_tap.h"

namespace {

using dvis_msda::kOOB;

constexpr int kTile = 64;      // 8x8 queries
constexpr int kCap = 296;      // pixels of one staged box (17x17 = 289 fits); x 128 B = 37.9 KB per buffer
constexpr int kThreads = 512;

struct PipeTiling {
  int tiles_cum[5];   // first tile index of each level (+ total)
  int tiles_x[4];     // tiles per row of each level
};

struct TapRec {       // 24 bytes per (query, sample)
  int hw;             // h0 | w0 as two int16 (h0 in the high half)
  unsigned flags;     // bit0..3: corner (0,0) (0,1) (1,0) (1,1) inside the map and sample counted
  float c[4];         // corner weights (0 when the sample is not counted)
};

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
__global__ __launch_bounds__(kThreads, 1) void msda_fwd_pipe_f32(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ off, int64_t off_stride, const float *__restrict__ logit, int64_t logit_stride,
    const float *__restrict__ refp, int nref, int S, int M, int Lq, int N, PipeTiling tiling, float *__restrict__ out) {
  constexpr int D = 32, LP = L * P, G = 8;
  constexpr int NST = (kCap * G + kThreads - 1) / kThreads;   // float4 per thread to stage one box
  static_assert(L == 3 && P == 4, "shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_val = reinterpret_cast<float4 *>(smem);                          // [2][(kCap + 1) * G]; pixel kCap = zeros
  TapRec *s_tap = reinterpret_cast<TapRec *>(s_val + 2 * (kCap + 1) * G);    // [2][kTile * LP]
  float *s_aw = reinterpret_cast<float *>(s_tap + 2 * kTile * LP);           // [2][kTile * LP]
  int *s_box = reinterpret_cast<int *>(s_aw + 2 * kTile * LP);               // [2][L][4]: min x, max x, min y, max y

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int g = lane >> 3, j = lane & 7;
  const int m = blockIdx.x;
  const int worker = blockIdx.y, workers = gridDim.y;
  const int MD = M * D;
  const unsigned pix_bytes = (unsigned)MD * 4u;
  const unsigned lane_bytes = (unsigned)j * 16u;
  const int ntiles = tiling.tiles_cum[L];
  const int nitems = N * ntiles;

  int Hs[L], Ws[L], LSi[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    LSi[l] = (int)level_start[l];
  }

  // ---- item -> (frame, tile geometry); all wave-uniform
  struct Item { int n, y0, x0, h, w, base; };
  auto decode = [&](int it) -> Item {
    Item r;
    r.n = it / ntiles;
    const int t = it - r.n * ntiles;
    int lv = 0;
#pragma unroll
    for (int ll = 1; ll < L; ++ll)
      if (t >= tiling.tiles_cum[ll]) lv = ll;
    const int ti = t - tiling.tiles_cum[lv];
    r.y0 = (ti / tiling.tiles_x[lv]) * 8;
    r.x0 = (ti % tiling.tiles_x[lv]) * 8;
    r.h = Hs[0]; r.w = Ws[0]; r.base = LSi[0];
#pragma unroll
    for (int ll = 1; ll < L; ++ll)
      if (lv == ll) { r.h = Hs[ll]; r.w = Ws[ll]; r.base = LSi[ll]; }
    return r;
  };
  auto slot_query = [&](const Item &it, int ql) -> int {
    const int y = it.y0 + (ql >> 3), x = it.x0 + (ql & 7);
    return (y < it.h && x < it.w) ? it.base + y * it.w + x : -1;
  };

  // ---- registers that carry tile i+1's raw offsets / logits from step (i,0) to step (i,1)   [threads < 256]
  float2 r_off[L];
  float4 r_lg[LP / 4];
  auto load_params = [&](const Item &it) {
    if (tid < 4 * kTile) {
      const int ql = tid >> 2, p = tid & 3;
      const int q = slot_query(it, ql);
      const size_t row = (size_t)it.n * Lq + (q >= 0 ? q : 0);
      const float *orow = off + row * off_stride + (size_t)m * (LP * 2);
      const float *lrow = logit + row * logit_stride + (size_t)m * LP;
#pragma unroll
      for (int l = 0; l < L; ++l) r_off[l] = *reinterpret_cast<const float2 *>(orow + 2 * (l * P + p));
#pragma unroll
      for (int k = 0; k < LP / 4; ++k) r_lg[k] = *reinterpret_cast<const float4 *>(lrow + 4 * k);
    }
  };
  // softmax + loc + taps + boxes of a tile into LDS slot `ts`   [threads < 256; uses r_off / r_lg]
  auto make_taps = [&](const Item &it, int ts) {
    if (tid < 4 * kTile) {
      const int ql = tid >> 2, p = tid & 3;
      const int q = slot_query(it, ql);
      const bool active = q >= 0;
      float lg[LP];
#pragma unroll
      for (int k = 0; k < LP / 4; ++k) { lg[4 * k] = r_lg[k].x; lg[4 * k + 1] = r_lg[k].y; lg[4 * k + 2] = r_lg[k].z; lg[4 * k + 3] = r_lg[k].w; }
      float mx = lg[0];
#pragma unroll
      for (int s = 1; s < LP; ++s) mx = fmaxf(mx, lg[s]);
      float e[LP], sum = 0.f;
#pragma unroll
      for (int s = 0; s < LP; ++s) { e[s] = expf(lg[s] - mx); sum += e[s]; }
#pragma unroll
      for (int l = 0; l < L; ++l) {
        const int H = Hs[l], W = Ws[l];
        float x = r_off[l].x, y = r_off[l].y;
        if (active) {
          const size_t rrow = ((size_t)(nref == 1 ? 0 : it.n) * Lq + q) * L + l;
          const float2 r = *reinterpret_cast<const float2 *>(refp + rrow * 2);
          x = r.x + x / (float)W;
          y = r.y + y / (float)H;
        }
        const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
        const bool ok = active && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h0 = ok ? (int)hf : 0, w0 = ok ? (int)wf : 0;
        const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
        const bool h0ok = ok && h0 >= 0, h1ok = ok && h0 + 1 <= H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= W - 1;
        TapRec t;
        t.hw = (int)(((unsigned)(h0 + 1) << 16) | ((unsigned)(w0 + 1) & 0xffffu));   // h0, w0 >= -1: stored + 1
        t.flags = (h0ok && w0ok ? 1u : 0u) | (h0ok && w1ok ? 2u : 0u) | (h1ok && w0ok ? 4u : 0u) | (h1ok && w1ok ? 8u : 0u);
        t.c[0] = ok ? hh * hw : 0.f; t.c[1] = ok ? hh * lw : 0.f; t.c[2] = ok ? lh * hw : 0.f; t.c[3] = ok ? lh * lw : 0.f;
        const int si = ql * LP + l * P + p;
        s_tap[ts * (kTile * LP) + si] = t;
        s_aw[ts * (kTile * LP) + si] = e[l * P + p] / sum;
        int bx0 = INT_MAX, bx1 = INT_MIN, by0 = INT_MAX, by1 = INT_MIN;
        if (ok) {
          bx0 = max(w0, 0); bx1 = min(w0 + 1, W - 1);
          by0 = max(h0, 0); by1 = min(h0 + 1, H - 1);
        }
        bx0 = wave_min(bx0); bx1 = wave_max(bx1); by0 = wave_min(by0); by1 = wave_max(by1);
        if (lane == 0) {
          int *b = s_box + (ts * L + l) * 4;
          atomicMin(b, bx0); atomicMax(b + 1, bx1); atomicMin(b + 2, by0); atomicMax(b + 3, by1);
        }
      }
    }
  };
  auto reset_box = [&](int ts) {
    if (tid < L * 4) s_box[ts * L * 4 + tid] = (tid & 1) ? INT_MIN : INT_MAX;
  };
  struct Box { int x0, y0, bw, npx; bool staged; };
  auto read_box = [&](int ts, int l) -> Box {
    const int *b = s_box + (ts * L + l) * 4;
    const int x0 = __builtin_amdgcn_readfirstlane(b[0]), x1 = __builtin_amdgcn_readfirstlane(b[1]);
    const int y0 = __builtin_amdgcn_readfirstlane(b[2]), y1 = __builtin_amdgcn_readfirstlane(b[3]);
    Box r;
    const bool any = x1 >= x0 && y1 >= y0;
    r.x0 = any ? x0 : 0; r.y0 = any ? y0 : 0;
    r.bw = any ? x1 - x0 + 1 : 1;
    r.npx = any ? r.bw * (y1 - y0 + 1) : 0;
    r.staged = r.npx <= kCap;
    return r;
  };
  auto make_rsrc = [&](int n, int l) {
    const float *base = value + (((size_t)n * S + (size_t)LSi[l]) * M + m) * D;
    return dvis_make_rsrc_uniform(base, (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * MD + D) * sizeof(float)));
  };

  // zero rows of both value buffers, boxes
  if (tid < G) {
    s_val[kCap * G + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_val[(kCap + 1) * G + kCap * G + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  reset_box(0);
  reset_box(1);
  __syncthreads();

  int it = worker;
  if (it >= nitems) return;
  Item cur = decode(it);
  load_params(cur);
  make_taps(cur, 0);
  __syncthreads();
  // stage (tile 0, level 0) synchronously
  unsigned step = 0;   // parity = value buffer of the CURRENT step
  {
    const Box b = read_box(0, 0);
    if (b.staged && b.npx > 0) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(cur.n, 0);
      const unsigned magic = ((1u << 20) + (unsigned)b.bw - 1u) / (unsigned)b.bw;
      const unsigned org = (unsigned)(b.y0 * Ws[0] + b.x0) * pix_bytes;
      for (int i = tid; i < b.npx * G; i += kThreads) {
        const unsigned pi = (unsigned)i >> 3, jj = (unsigned)i & 7u;
        const unsigned py = (pi * magic) >> 20, px = pi - py * (unsigned)b.bw;
        const dvis_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, org + (py * (unsigned)Ws[0] + px) * pix_bytes + jj * 16u, 0, 0);
        s_val[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
    }
  }
  __syncthreads();

  int ts = 0;   // tap slot of the current tile
  while (true) {
    const int nxt_it = it + workers;
    const bool has_next = nxt_it < nitems;
    Item nxt = cur;
    if (has_next) nxt = decode(nxt_it);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int ql = wv * 8 + g;
    const int q = slot_query(cur, ql);

#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int buf = (int)(step & 1u);
      // ---- A: issue the loads of the next step's box
      const bool to_next_tile = l == L - 1;
      const int nl = to_next_tile ? 0 : l + 1;
      const bool stage_valid = !to_next_tile || has_next;
      Box nb = {0, 0, 1, 0, false};
      if (stage_valid) nb = read_box(to_next_tile ? ts ^ 1 : ts, nl);
      const bool do_stage = stage_valid && nb.staged && nb.npx > 0;
      dvis_v4u st[NST];
      if (do_stage) {
        const int Wn = Ws[0] * (nl == 0) + (L > 1 ? Ws[1] * (nl == 1) : 0) + (L > 2 ? Ws[2] * (nl == 2) : 0);
        const __amdgpu_buffer_rsrc_t rsn = make_rsrc(to_next_tile ? nxt.n : cur.n, nl);
        const unsigned magic = ((1u << 20) + (unsigned)nb.bw - 1u) / (unsigned)nb.bw;
        const unsigned org = (unsigned)(nb.y0 * Wn + nb.x0) * pix_bytes;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
          const int i = tid + kThreads * k;
          const unsigned pi = (unsigned)i >> 3, jj = (unsigned)i & 7u;
          const unsigned py = (pi * magic) >> 20, px = pi - py * (unsigned)nb.bw;
          st[k] = __builtin_amdgcn_raw_buffer_load_b128(
              rsn, i < nb.npx * G ? org + (py * (unsigned)Wn + px) * pix_bytes + jj * 16u : kOOB, 0, 0);
        }
      }
      // ---- A': raw parameters of the next tile
      if (l == 0 && has_next) load_params(nxt);
      if (l == 0) reset_box(ts ^ 1);           // last read of that slot's boxes was step (i-1, 2); filled again in D

      // ---- B: gather this step
      {
        const int H = Hs[l], W = Ws[l];
        const Box cb = read_box(ts, l);
        const __amdgpu_buffer_rsrc_t rsc = make_rsrc(cur.n, l);
        const char *vb = reinterpret_cast<const char *>(s_val + buf * (kCap + 1) * G);
        const TapRec *taps = s_tap + ts * (kTile * LP) + ql * LP + l * P;
        const float *aws = s_aw + ts * (kTile * LP) + ql * LP + l * P;
        constexpr int B = 2;                     // points per batch: 4*B corner reads in flight (register budget)
#pragma unroll 1
        for (int pb = 0; pb < P / B; ++pb) {
          float4 r[4 * B];
          float cw[4 * B];
#pragma unroll
          for (int pp = 0; pp < B; ++pp) {
            const TapRec t = taps[pb * B + pp];
            const int h0 = (int)((unsigned)t.hw >> 16) - 1, w0 = (int)((unsigned)t.hw & 0xffffu) - 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) cw[4 * pp + c] = t.c[c];
            if (cb.staged) {
              const unsigned zero = (unsigned)kCap * 128u + lane_bytes;
              const unsigned o00 = (unsigned)((h0 - cb.y0) * cb.bw + (w0 - cb.x0)) * 128u + lane_bytes;
              const unsigned row = (unsigned)cb.bw * 128u;
              r[4 * pp] = *reinterpret_cast<const float4 *>(vb + ((t.flags & 1u) ? o00 : zero));
              r[4 * pp + 1] = *reinterpret_cast<const float4 *>(vb + ((t.flags & 2u) ? o00 + 128u : zero));
              r[4 * pp + 2] = *reinterpret_cast<const float4 *>(vb + ((t.flags & 4u) ? o00 + row : zero));
              r[4 * pp + 3] = *reinterpret_cast<const float4 *>(vb + ((t.flags & 8u) ? o00 + row + 128u : zero));
            } else {
              const unsigned o00 = (unsigned)(h0 * W + w0) * pix_bytes + lane_bytes;
              const unsigned o[4] = {(t.flags & 1u) ? o00 : kOOB, (t.flags & 2u) ? o00 + pix_bytes : kOOB,
                                     (t.flags & 4u) ? o00 + (unsigned)W * pix_bytes : kOOB,
                                     (t.flags & 8u) ? o00 + (unsigned)W * pix_bytes + pix_bytes : kOOB};
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const dvis_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsc, o[c], 0, 0);
                r[4 * pp + c] =
                    make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
              }
            }
          }
#pragma unroll
          for (int pp = 0; pp < B; ++pp) {
            const float4 r1 = r[4 * pp], r2 = r[4 * pp + 1], r3 = r[4 * pp + 2], r4 = r[4 * pp + 3];
            const float c1 = cw[4 * pp], c2 = cw[4 * pp + 1], c3 = cw[4 * pp + 2], c4 = cw[4 * pp + 3];
            const float aw = aws[pb * B + pp];
            // reference order: (w1 v1 + w2 v2 + w3 v3 + w4 v4) * weight, accumulated over samples
            a0 += (c1 * r1.x + c2 * r2.x + c3 * r3.x + c4 * r4.x) * aw;
            a1 += (c1 * r1.y + c2 * r2.y + c3 * r3.y + c4 * r4.y) * aw;
            a2 += (c1 * r1.z + c2 * r2.z + c3 * r3.z + c4 * r4.z) * aw;
            a3 += (c1 * r1.w + c2 * r2.w + c3 * r3.w + c4 * r4.w) * aw;
          }
        }
      }

      // ---- C: staged registers -> the other value buffer
      if (do_stage) {
        float4 *dst = s_val + (buf ^ 1) * (kCap + 1) * G;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
          const int i = tid + kThreads * k;
          if (i < nb.npx * G)
            dst[i] = make_float4(__uint_as_float(st[k].x), __uint_as_float(st[k].y), __uint_as_float(st[k].z),
                                 __uint_as_float(st[k].w));
        }
      }
      // ---- D: taps and boxes of the next tile
      if (l == 1 && has_next) make_taps(nxt, ts ^ 1);
      ++step;
      __syncthreads();
    }
    if (q >= 0)
      *reinterpret_cast<float4 *>(out + (((size_t)cur.n * Lq + q) * M + m) * D + 4 * j) = make_float4(a0, a1, a2, a3);
    if (!has_next) break;
    it = nxt_it;
    cur = nxt;
    ts ^= 1;
  }
}

// OFF by default.  Measured on MI355X (30 frames / launch, init-rule offsets, parity-tested against the tiled kernel):
// 94.7 us per frame-layer vs 35.2 us.  PMC: the L1 accesses do drop 3.5x (TCP_TOTAL_CACHE_ACCESSES 1.27e8 vs 4.5e8, TA
// busy 18 %), LDS is 7 % busy, VALU 39 % (6.8e8 instructions: per-lane corner selects and staging addresses), and
// SQ_WAIT_ANY is 54 % of the wave cycles: with ONE workgroup per CU (119 KB of LDS) all 8 waves move through the
// phases in lock-step behind the per-step barrier, so TA, LDS and VALU are used one after the other instead of
// concurrently, and 2 waves per SIMD cannot cover the dependent LDS-read -> address -> LDS-read chains.  The tiled
// kernel's 16-32 independent waves per CU overlap those naturally.  A faster form needs warp-specialised producer /
// consumer waves without workgroup-wide barriers, or two out-of-phase workgroups per CU (<= 80 KB each).
// DVIS_MSDA_PIPE=1 enables it for experiments.
bool pipe_enabled() {
  static const bool v = [] {
    const char *e = getenv("DVIS_MSDA_PIPE");
    return e != nullptr && atoi(e) != 0;
  }();
  return v;
}

}  // namespace

// Internal (not exported): called by dvis_msda_fused_forward.  *handled = false when the shape is not this kernel's.
int dvis_msda_pipe_launch(const float *value, const int64_t *shapes, const int64_t *level_start, const float *ref, int nref,
                          const float *offsets, int64_t off_stride, const float *logits, int64_t logit_stride, int N, int S,
                          int M, int D, int L, int Lq, int P, float *out, const int64_t *shapes_host, hipStream_t st,
                          bool *handled) {
  *handled = false;
  if (!pipe_enabled() || shapes_host == nullptr || D != 32 || L != 3 || P != 4) return DVIS_OK;
  PipeTiling tiling = {};
  long long total = 0;
  int cum = 0;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes_host[2 * l], W = (int)shapes_host[2 * l + 1];
    if (H > 32000 || W > 32000) return DVIS_OK;   // h0 / w0 are packed as int16
    total += (long long)H * W;
    tiling.tiles_cum[l] = cum;
    tiling.tiles_x[l] = (W + 7) / 8;
    cum += ((H + 7) / 8) * ((W + 7) / 8);
  }
  tiling.tiles_cum[L] = cum;
  if (total != Lq || total != S || (long long)N * cum > 0x7fffffffll) return DVIS_OK;
  static int ncu = [] {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount;
  }();
  int workers = ncu / (M > 0 ? M : 1);
  if (workers < 1) workers = 1;
  const long long items = (long long)N * cum;
  if (workers > items) workers = (int)items;
  constexpr size_t lds = 2 * (kCap + 1) * 8 * sizeof(float4) + 2 * kTile * 12 * sizeof(TapRec) + 2 * kTile * 12 * sizeof(float) +
                         2 * 3 * 4 * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&msda_fwd_pipe_f32<3, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      dvis_set_error("msda pipe: cannot reserve %zu bytes of LDS", lds);
      return DVIS_E_LAUNCH;
    }
    attr_set = true;
  }
  *handled = true;
  hipLaunchKernelGGL((msda_fwd_pipe_f32<3, 4>), dim3(M, workers, 1), dim3(kThreads), lds, st, value, shapes, level_start,
                     offsets, off_stride, logits, logit_stride, ref, nref, S, M, Lq, N, tiling, out);
  return dvis_check_launch("msda_fwd_pipe_f32");
}
