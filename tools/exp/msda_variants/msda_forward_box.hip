// MSDeformAttn forward for encoder self-attention, corners served from LDS-staged boxes — gfx950.
//
// Why: the tiled kernel (msda_forward.hip) pulls every corner of every sample through the per-CU texture-address / L1
// pipe — one 128-byte line = two 64-byte accesses per (sample, head, corner), 14.8 M accesses per 720p frame-layer —
// and that pipe is its bound (TA busy 93 %).  In encoder self-attention the queries ARE the pixels of the L maps and
// the sampling offsets are a few pixels around the query's own position, so the samples of a small tile of queries
// fall into a small box of every level.  This kernel makes that explicit WITHOUT assuming it:
//   1. a workgroup owns one head and one 8x4 query tile of one level; 128 of its threads read the raw offsets / logits
//      of "their" (query, point) straight into registers, apply softmax and loc = ref + off / (W_l, H_l)
//      (ops/modules/ms_deform_attn.py:101-109) and set up the bilinear taps of the 3 levels ONCE per (query, sample);
//   2. the per-level bounding boxes of all touched corners are reduced (wave min/max + 4 LDS atomics per level);
//   3. the boxes are packed into one LDS region, level 0 first, as long as they fit (kCap pixels); they are copied
//      with coalesced 16-byte buffer loads — every line fetched ONCE instead of once per sample that touches it —
//      while the tap threads rewrite their taps into final form: 4 byte offsets into the LDS region (corners outside
//      the map point at a zero row) or, for a level whose box did not fit, 4 byte offsets into the level's value slice
//      (out-of-range offset = hardware zero), exactly as the tiled kernel;
//   4. every lane gathers its 12 samples x 4 corners with ds_read_b128 (or buffer loads for an unstaged level) and
//      accumulates in the reference's order.
// Two global round trips and two barriers per tile, ~45 KB of LDS: three workgroups per CU overlap each other.
// (First form of this kernel: per-lane tap set-up and one stage + barrier pair per level — 55.3 us per frame-layer, VALU
// 61 % busy, five serialised round trips; history in git.)
// Fused interface only (raw offsets / logits + reference points), fp32, D = 32, (L, P) = (3, 4), queries = pixels.
#include <limits.h>
#include <stdlib.h>

#include "dvis_common.h"
#include "msda_tap.h"

namespace {

using dvis_msda::kOOB;

constexpr int kTW = 8, kTH = 4, kTile = kTW * kTH;   // 8x4 queries
constexpr int kCap = 240;                            // staged pixels of all levels together (x 128 B = 30 KB)

struct BoxTiling {
  int tiles_cum[5];   // first tile index of each level (+ total)
  int tiles_x[4];     // tiles per row of each level
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
__global__ __launch_bounds__(256, 3) void msda_fwd_box_f32(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ off, int64_t off_stride, const float *__restrict__ logit, int64_t logit_stride,
    const float *__restrict__ refp, int nref, int S, int M, int Lq, int N, BoxTiling tiling, float *__restrict__ out) {
  constexpr int D = 32, LP = L * P, G = 8;
  constexpr int NST = (kCap * G + 255) / 256;        // float4 per thread to stage the whole region
  static_assert(L == 3 && P == 4 && kTile * G == 256, "shape");

  __shared__ float4 s_val[(kCap + 1) * G];           // staged boxes, pixel-major; pixel kCap is the zero row
  __shared__ uint4 s_tap_o[kTile * LP];              // 4 corner byte offsets (LDS or global, see `staged`)
  __shared__ float4 s_tap_c[kTile * LP];             // 4 corner weights
  __shared__ float s_aw[kTile * LP];                 // attention weights (softmax over the L*P logits)
  __shared__ int s_box[L * 4];                       // per level: min x, max x, min y, max y of the touched corners

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int m = blockIdx.x;
  const int MD = M * D;
  const unsigned pix_bytes = (unsigned)MD * 4u;
  const int ntiles = tiling.tiles_cum[L];
  const int nitems = N * ntiles;

  int Hs[L], Ws[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
  }

  if (threadIdx.x < G) s_val[kCap * G + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
  // Persistent: gridDim.y workers per head (3 per CU) walk the (frame, tile) items.  With one workgroup per tile the
  // launch is dispatch-bound: 151 k four-wave workgroups of ~9 us each left 2.9 waves resident per CU on average (PMC).
  struct Geo { int n, y0, x0, h, w, base; };
  auto decode = [&](int item) -> Geo {
    Geo r;
    r.n = item / ntiles;
    const int tile = item - r.n * ntiles;
    int lv = 0;
#pragma unroll
    for (int ll = 1; ll < L; ++ll)
      if (tile >= tiling.tiles_cum[ll]) lv = ll;
    const int ti = tile - tiling.tiles_cum[lv];
    r.y0 = (ti / tiling.tiles_x[lv]) * kTH;
    r.x0 = (ti % tiling.tiles_x[lv]) * kTW;
    r.h = Hs[0]; r.w = Ws[0]; r.base = (int)level_start[0];
#pragma unroll
    for (int ll = 1; ll < L; ++ll)
      if (lv == ll) { r.h = Hs[ll]; r.w = Ws[ll]; r.base = (int)level_start[ll]; }
    return r;
  };
  auto geo_query = [&](const Geo &gq, int ql) -> int {
    const int y = gq.y0 + ql / kTW, x = gq.x0 + ql % kTW;
    return (y < gq.h && x < gq.w) ? gq.base + y * gq.w + x : -1;
  };
  // raw parameters of an item for the tap threads ((query tid >> 2, point tid & 3)); loaded one item AHEAD, under the
  // previous item's gather, so that only the box staging is an exposed global round trip
  const bool tap_thread = threadIdx.x < kTile * P;
  float2 ro[L], rr[L];
  float4 rl[LP / 4];
  auto load_params = [&](const Geo &gq) {
    if (tap_thread) {
      const int q = geo_query(gq, threadIdx.x >> 2);
      const int p = threadIdx.x & 3;
      const size_t row = (size_t)gq.n * Lq + (q >= 0 ? q : 0);
      const float *orow = off + row * off_stride + (size_t)m * (LP * 2);
      const float *lrow = logit + row * logit_stride + (size_t)m * LP;
#pragma unroll
      for (int l = 0; l < L; ++l) {
        ro[l] = *reinterpret_cast<const float2 *>(orow + 2 * (l * P + p));
        rr[l] = *reinterpret_cast<const float2 *>(refp + (((size_t)(nref == 1 ? 0 : gq.n) * Lq + (q >= 0 ? q : 0)) * L + l) * 2);
      }
#pragma unroll
      for (int k = 0; k < LP / 4; ++k) rl[k] = *reinterpret_cast<const float4 *>(lrow + 4 * k);
    }
  };
  Geo cur = decode((int)blockIdx.y < nitems ? blockIdx.y : 0);
  if ((int)blockIdx.y < nitems) load_params(cur);
#pragma unroll 1
  for (int item = blockIdx.y; item < nitems; item += gridDim.y) {
  const int n = cur.n;
  auto slot_query = [&](int ql) -> int { return geo_query(cur, ql); };

  if (tid < L * 4) s_box[tid] = (tid & 1) ? INT_MIN : INT_MAX;
  __syncthreads();      // also: every wave is done gathering the previous item (taps / boxes / values are reused)

  // ---- phase 1 (tap threads): parameters (already in registers) -> taps of the 3 levels in registers
  int t_h0[L], t_w0[L];
  unsigned t_fl[L];
  float t_c[L][4];
  if (tap_thread) {
    const int ql = tid >> 2, p = tid & 3;
    const int q = slot_query(ql);
    const bool active = q >= 0;
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
      const int H = Hs[l], W = Ws[l];
      const float x = rr[l].x + ro[l].x / (float)W, y = rr[l].y + ro[l].y / (float)H;
      const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
      const bool ok = active && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
      const float hf = floorf(h_im), wf = floorf(w_im);
      const int h0 = ok ? (int)hf : 0, w0 = ok ? (int)wf : 0;
      const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
      const bool h0ok = ok && h0 >= 0, h1ok = ok && h0 + 1 <= H - 1, w0ok = w0 >= 0, w1ok = w0 + 1 <= W - 1;
      t_h0[l] = h0; t_w0[l] = w0;
      t_fl[l] = (h0ok && w0ok ? 1u : 0u) | (h0ok && w1ok ? 2u : 0u) | (h1ok && w0ok ? 4u : 0u) | (h1ok && w1ok ? 8u : 0u);
      t_c[l][0] = ok ? hh * hw : 0.f; t_c[l][1] = ok ? hh * lw : 0.f; t_c[l][2] = ok ? lh * hw : 0.f; t_c[l][3] = ok ? lh * lw : 0.f;
      s_aw[ql * LP + l * P + p] = e[l * P + p] / sum;
      int bx0 = INT_MAX, bx1 = INT_MIN, by0 = INT_MAX, by1 = INT_MIN;
      if (ok) {
        bx0 = max(w0, 0); bx1 = min(w0 + 1, W - 1);
        by0 = max(h0, 0); by1 = min(h0 + 1, H - 1);
      }
      bx0 = wave_min(bx0); bx1 = wave_max(bx1); by0 = wave_min(by0); by1 = wave_max(by1);
      if (lane == 0) {
        atomicMin(&s_box[4 * l], bx0); atomicMax(&s_box[4 * l + 1], bx1);
        atomicMin(&s_box[4 * l + 2], by0); atomicMax(&s_box[4 * l + 3], by1);
      }
    }
  }
  __syncthreads();

  // ---- boxes -> packing of the LDS region (wave-uniform): level l occupies pixels [pbase[l], pbase[l] + npx[l])
  int bx0[L], by0[L], bw[L], npx[L], pbase[L];
  bool staged[L];
  {
    int used = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int x0 = __builtin_amdgcn_readfirstlane(s_box[4 * l]), x1 = __builtin_amdgcn_readfirstlane(s_box[4 * l + 1]);
      const int y0 = __builtin_amdgcn_readfirstlane(s_box[4 * l + 2]), y1 = __builtin_amdgcn_readfirstlane(s_box[4 * l + 3]);
      const bool any = x1 >= x0 && y1 >= y0;
      bx0[l] = any ? x0 : 0; by0[l] = any ? y0 : 0;
      bw[l] = any ? x1 - x0 + 1 : 1;
      npx[l] = any ? bw[l] * (y1 - y0 + 1) : 0;
      staged[l] = used + npx[l] <= kCap;
      pbase[l] = used;
      if (staged[l]) used += npx[l];
    }
  }
  __amdgpu_buffer_rsrc_t rs[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float *base = value + (((size_t)n * S + (size_t)level_start[l]) * M + m) * D;
    rs[l] = dvis_make_rsrc_uniform(base, (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * MD + D) * sizeof(float)));
  }

  // ---- phase 2a: issue the loads of every staged box (all threads), ONE global round trip.  The region is walked as
  // one index space [0, used * G): thread-slot i belongs to the level whose pixel range contains i / G.
  const int used_px = (staged[0] ? npx[0] : 0) + (staged[1] ? npx[1] : 0) + (staged[2] ? npx[2] : 0);
  unsigned magic[L], org[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    magic[l] = ((1u << 20) + (unsigned)bw[l] - 1u) / (unsigned)bw[l];
    org[l] = (unsigned)(by0[l] * Ws[l] + bx0[l]) * pix_bytes;
  }
  dvis_v4u st[NST];
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + 256 * k;
    const int pi = i >> 3;
    const unsigned jj = (unsigned)i & 7u;
    st[k] = dvis_v4u{0u, 0u, 0u, 0u};
#pragma unroll
    for (int l = 0; l < L; ++l) {
      if (staged[l] && pi >= pbase[l] && pi < pbase[l] + npx[l]) {
        const unsigned rel = (unsigned)(pi - pbase[l]);
        const unsigned py = (rel * magic[l]) >> 20, px = rel - py * (unsigned)bw[l];
        st[k] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], org[l] + (py * (unsigned)Ws[l] + px) * pix_bytes + jj * 16u, 0, 0);
      }
    }
  }
  // ---- phase 2b (tap threads, under the loads): final tap offsets
  if (tap_thread) {
    const int ql = tid >> 2, p = tid & 3;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      uint4 o;
      if (staged[l]) {
        const unsigned zero = (unsigned)kCap * 128u;
        const unsigned o00 = (unsigned)(pbase[l] + (t_h0[l] - by0[l]) * bw[l] + (t_w0[l] - bx0[l])) * 128u;
        const unsigned rowb = (unsigned)bw[l] * 128u;
        o.x = (t_fl[l] & 1u) ? o00 : zero;
        o.y = (t_fl[l] & 2u) ? o00 + 128u : zero;
        o.z = (t_fl[l] & 4u) ? o00 + rowb : zero;
        o.w = (t_fl[l] & 8u) ? o00 + rowb + 128u : zero;
      } else {
        const unsigned o00 = (unsigned)(t_h0[l] * Ws[l] + t_w0[l]) * pix_bytes;
        const unsigned rowb = (unsigned)Ws[l] * pix_bytes;
        o.x = (t_fl[l] & 1u) ? o00 : kOOB;
        o.y = (t_fl[l] & 2u) ? o00 + pix_bytes : kOOB;
        o.z = (t_fl[l] & 4u) ? o00 + rowb : kOOB;
        o.w = (t_fl[l] & 8u) ? o00 + rowb + pix_bytes : kOOB;
      }
      const int si = ql * LP + l * P + p;
      s_tap_o[si] = o;
      s_tap_c[si] = make_float4(t_c[l][0], t_c[l][1], t_c[l][2], t_c[l][3]);
    }
  }
  // ---- phase 2c: staged registers -> LDS
#pragma unroll
  for (int k = 0; k < NST; ++k) {
    const int i = tid + 256 * k;
    if (i < used_px * G)
      s_val[i] = make_float4(__uint_as_float(st[k].x), __uint_as_float(st[k].y), __uint_as_float(st[k].z),
                             __uint_as_float(st[k].w));
  }
  __syncthreads();

  // ---- next item's parameters: in flight during the gather
  Geo nxt = cur;
  if (item + (int)gridDim.y < nitems) {
    nxt = decode(item + gridDim.y);
    load_params(nxt);
  }
  // ---- phase 3: gather.  Lane group g of wave wv = query wv*8 + g; lane j owns channels 4j..4j+3.
  const int wv = tid >> 6, g = lane >> 3, j = lane & 7;
  const int ql = wv * 8 + g;
  const int q = slot_query(ql);
  const unsigned lane_bytes = (unsigned)j * 16u;
  const char *vb = reinterpret_cast<const char *>(s_val);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    constexpr int B = 2;                       // points per batch: 4*B corner reads in flight (register budget)
#pragma unroll 1
    for (int pb = 0; pb < P / B; ++pb) {
      float4 r[4 * B];
      float4 cw[B];
      float aw[B];
#pragma unroll
      for (int pp = 0; pp < B; ++pp) {
        const int si = ql * LP + l * P + pb * B + pp;
        const uint4 o = s_tap_o[si];
        cw[pp] = s_tap_c[si];
        aw[pp] = s_aw[si];
        if (staged[l]) {
          r[4 * pp] = *reinterpret_cast<const float4 *>(vb + o.x + lane_bytes);
          r[4 * pp + 1] = *reinterpret_cast<const float4 *>(vb + o.y + lane_bytes);
          r[4 * pp + 2] = *reinterpret_cast<const float4 *>(vb + o.z + lane_bytes);
          r[4 * pp + 3] = *reinterpret_cast<const float4 *>(vb + o.w + lane_bytes);
        } else {
          const unsigned oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const dvis_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs[l], oo[c] + lane_bytes, 0, 0);
            r[4 * pp + c] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
          }
        }
      }
#pragma unroll
      for (int pp = 0; pp < B; ++pp) {
        const float4 r1 = r[4 * pp], r2 = r[4 * pp + 1], r3 = r[4 * pp + 2], r4 = r[4 * pp + 3];
        const float c1 = cw[pp].x, c2 = cw[pp].y, c3 = cw[pp].z, c4 = cw[pp].w;
        // reference order: (w1 v1 + w2 v2 + w3 v3 + w4 v4) * weight, accumulated over samples
        a0 += (c1 * r1.x + c2 * r2.x + c3 * r3.x + c4 * r4.x) * aw[pp];
        a1 += (c1 * r1.y + c2 * r2.y + c3 * r3.y + c4 * r4.y) * aw[pp];
        a2 += (c1 * r1.z + c2 * r2.z + c3 * r3.z + c4 * r4.z) * aw[pp];
        a3 += (c1 * r1.w + c2 * r2.w + c3 * r3.w + c4 * r4.w) * aw[pp];
      }
    }
  }
  if (q >= 0)
    *reinterpret_cast<float4 *>(out + (((size_t)n * Lq + q) * M + m) * D + 4 * j) = make_float4(a0, a1, a2, a3);
  cur = nxt;
  }   // items
}

// OFF by default (DVIS_MSDA_BOX=1 enables it; parity-tested against the tiled kernel).  Measured on MI355X, 30 frames
// per launch, init-rule offsets (every level-2 tile's boxes fit): 61.5 us per frame-layer vs 35.2 us for the tiled kernel,
// although the L1 accesses drop 3.4x (TCP_TOTAL_CACHE_ACCESSES 1.3e8 vs 4.5e8, TA busy 19 %) and LDS is 11 % busy.
// Switching phases off (results invalid, timing only): without the gather 49.5 us, without the staging loads 56.4 us,
// without both 41.9 us — parameter loads, tap set-up, box reduction, tap rewrite and three barriers per 32-query tile
// cost more than the whole tiled kernel.  One workgroup per tile vs a persistent grid: 61.5 vs 60.8 us (not dispatch-
// bound); prefetching the next tile's parameters under the gather: no change.  With 128 bytes per (pixel, head) the
// boxes leave 12 waves per CU, too few to cover two dependent global round trips per tile; the tiled kernel's 16-32
// independent waves per CU hide latency without any barrier after its prologue.
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
    tiling.tiles_x[l] = (W + kTW - 1) / kTW;
    cum += ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
  }
  tiling.tiles_cum[L] = cum;
  // only when the queries are exactly the pixels of the maps (encoder self-attention)
  if (total != Lq || total != S || (long long)N * cum > 0x7fffffffll) return DVIS_OK;
  static int ncu = [] {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount;
  }();
  long long workers = (3ll * ncu + M - 1) / M;              // 3 resident workgroups per CU (45 KB of LDS each)
  if (workers > (long long)N * cum) workers = (long long)N * cum;
  if (workers > 65535) workers = 65535;
  *handled = true;
  hipLaunchKernelGGL((msda_fwd_box_f32<3, 4>), dim3(M, (unsigned)workers, 1), dim3(256), 0, st, value, shapes, level_start,
                     offsets, off_stride, logits, logit_stride, ref, nref, S, M, Lq, N, tiling, out);
  return dvis_check_launch("msda_fwd_box_f32");
}
