// EXPERIMENT (round 2): MSDeformAttn forward by SPATIAL CELLS with the finest level's value tile staged in LDS.
//
// A workgroup owns one head and one cell = 8 x 8 pixels of the finest map = 4 x 4 of the middle = 2 x 2 of the coarsest
// (the three maps are exact 2x pyramids) — the 84 queries whose reference points fall into the cell.  All of them sample
// the finest level within a few pixels of the cell, so ONE (8 + 2*HALO)^2 pixel box of this head's value rows (41.5 KB
// at HALO = 5) is loaded once, coalesced (zero-filled outside the map), and the 84 x 4 x 4 corner reads of the finest level come
// from LDS (128 B/clk) instead of the vector L1's 64 B/clk return path, which is what bounds the tile kernel
// (949 MB delivered to registers per 720p frame-layer = 27.6 us at 2.1 GHz; measured 36.5).  The two coarser levels keep
// the global gather (their boxes would be mostly halo and they already run at the L1's peak).  A sample that leaves the
// box takes the global path (wave-uniform branch).
#include <stdlib.h>
#include "../../../dvis_plus_amd/csrc/dvis_common.h"
#include "../../../dvis_plus_amd/csrc/msda_tap.h"
namespace {
using dvis_msda::accumulate_sample;
using dvis_msda::kOOB;
using dvis_msda::make_tap;
using dvis_msda::Tap;

constexpr int kHalo = 5;
constexpr int kBox = 8 + 2 * kHalo;      // 18 pixels
constexpr int kQS = 84;                  // query slots of a cell: 64 + 16 + 4
constexpr int kSlots = 96;               // 3 iterations x 32 lane groups

struct alignas(16) BoxTap {          // one finest-level sample: 32 bytes
  unsigned short a[4];   // box pixel indices of the 4 corners (in-box samples)
  unsigned o00;          // global byte offset of corner (h0, w0) for the fall-back path (may be "negative": see flags)
  unsigned flags;        // bit 0: sample left the box -> global path; bits 1..4: corner k is inside the map
  float c[4];
};

template <int D, int P, int EXP>
__global__ __launch_bounds__(256, 2) void msda_fwd_cell(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ off, int64_t off_stride, const float *__restrict__ logit, int64_t logit_stride,
    const float *__restrict__ refp, int nref, int S, int M, int Lq, float *__restrict__ out, int cells_x) {
  constexpr int L = 3, LP = L * P, G = D / 4, GPW = 64 / G;
  static_assert(D == 32 && P == 4, "cell kernel shape");
  __shared__ __attribute__((aligned(16))) float s_box[kBox * kBox * D];     // 41 472 B
  __shared__ uint2 s_tap_o[kQS * 2 * P];     // levels 0, 1: (global offset of corner 00 or kOOB-ish, validity flags)
  __shared__ float4 s_tap_c[kQS * 2 * P];
  __shared__ BoxTap s_box_tap[kQS * P];
  __shared__ float s_aw[kQS * LP];

  const int tid = threadIdx.x;
  const int m = blockIdx.x, cell = blockIdx.y, n = blockIdx.z;
  const int MD = M * D;
  const int cy = cell / cells_x, cx = cell - cy * cells_x;
  int Hs[L], Ws[L], Q0[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    Hs[l] = (int)shapes[2 * l];
    Ws[l] = (int)shapes[2 * l + 1];
    Q0[l] = (int)level_start[l];
  }
  // query slot -> global query index (levels are stored coarse first: l = 0 coarsest ... 2 finest), -1 = outside the map
  auto slot_query = [&](int s) -> int {
    if (s >= kQS) return -1;
    int l, y, x;
    if (s < 64) { l = 2; y = cy * 8 + (s >> 3); x = cx * 8 + (s & 7); }
    else if (s < 80) { l = 1; y = cy * 4 + ((s - 64) >> 2); x = cx * 4 + ((s - 64) & 3); }
    else { l = 0; y = cy * 2 + ((s - 80) >> 1); x = cx * 2 + ((s - 80) & 1); }
    const int H = l == 2 ? Hs[2] : (l == 1 ? Hs[1] : Hs[0]), W = l == 2 ? Ws[2] : (l == 1 ? Ws[1] : Ws[0]);
    const int q0 = l == 2 ? Q0[2] : (l == 1 ? Q0[1] : Q0[0]);
    return (y < H && x < W) ? q0 + y * W + x : -1;
  };
  const unsigned pix_bytes = (unsigned)MD * 4u;
  const int by0 = cy * 8 - kHalo, bx0 = cx * 8 - kHalo;

  // ---- per-level buffer descriptors over this (frame, head) slice of `value`
  __amdgpu_buffer_rsrc_t rs[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float *base = value + (((size_t)n * S + (size_t)Q0[l]) * M + m) * D;
    rs[l] = dvis_make_rsrc_uniform(base, (unsigned)(((size_t)(Hs[l] * Ws[l] - 1) * MD + D) * sizeof(float)));
  }

  // ---- box fill: requested first, so that it is in flight under the set-up arithmetic
  constexpr int kFill = (kBox * kBox * G + 255) / 256;   // 16-byte pieces per thread
  dvis_v4u fill[kFill];
#pragma unroll
  for (int i = 0; i < kFill; ++i) {
    const int idx = tid + 256 * i;
    const int px = idx / G, piece = idx - px * G;
    const int by = px / kBox, bx = px - by * kBox;
    const int gy = by0 + by, gx = bx0 + bx;
    const bool ok = idx < kBox * kBox * G && gy >= 0 && gy < Hs[2] && gx >= 0 && gx < Ws[2];
    fill[i] = __builtin_amdgcn_raw_buffer_load_b128(rs[2], (ok && !(EXP & 1)) ? (unsigned)(gy * Ws[2] + gx) * pix_bytes + piece * 16u : kOOB,
                                                    0, 0);
  }

  // ---- set-up: item (query slot, point) computes its three samples.  84 x 4 = 336 items on 256 threads: the inputs of
  // a thread's second item are requested together with the first one's (one global round trip, not two)
  float2 ro[2][L], rr[2][L];
  float4 rl[2][LP / 4];
  int qs[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int item = tid + 256 * h;
    const int ql = item / P, p = item - ql * P;
    qs[h] = item < kQS * P ? slot_query(ql) : -1;
    const size_t qq = qs[h] >= 0 ? qs[h] : 0;
    const float *orow = off + ((size_t)n * Lq + qq) * off_stride + (size_t)m * (LP * 2);
    const float *lrow = logit + ((size_t)n * Lq + qq) * logit_stride + (size_t)m * LP;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      ro[h][l] = *reinterpret_cast<const float2 *>(orow + 2 * (l * P + p));
      rr[h][l] = *reinterpret_cast<const float2 *>(refp + (((size_t)(nref == 1 ? 0 : n) * Lq + qq) * L + l) * 2);
    }
#pragma unroll
    for (int k = 0; k < LP / 4; ++k) rl[h][k] = *reinterpret_cast<const float4 *>(lrow + 4 * k);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int item = tid + 256 * h;
    if (item >= kQS * P) continue;
    const int ql = item / P, p = item - ql * P;
    const bool active = qs[h] >= 0;
    float lg[LP];
#pragma unroll
    for (int k = 0; k < LP / 4; ++k) { lg[4 * k] = rl[h][k].x; lg[4 * k + 1] = rl[h][k].y; lg[4 * k + 2] = rl[h][k].z; lg[4 * k + 3] = rl[h][k].w; }
    float mx = lg[0];
#pragma unroll
    for (int s2 = 1; s2 < LP; ++s2) mx = fmaxf(mx, lg[s2]);
    float e[LP], sum = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < LP; ++s2) { e[s2] = expf(lg[s2] - mx); sum += e[s2]; }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float x = rr[h][l].x + ro[h][l].x / (float)Ws[l];
      const float y = rr[h][l].y + ro[h][l].y / (float)Hs[l];
      float ev = e[l * P];
#pragma unroll
      for (int pp = 1; pp < P; ++pp) ev = (p == pp) ? e[l * P + pp] : ev;
      s_aw[ql * LP + l * P + p] = ev / sum;
      const Tap t = make_tap(x, y, Hs[l], Ws[l], active, pix_bytes, 0u);
      // corner 00's offset without the validity test, and the four validity bits (make_tap folds them into kOOB)
      const float h_im = y * (float)Hs[l] - 0.5f, w_im = x * (float)Ws[l] - 0.5f;
      const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im);
      unsigned vbits = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) vbits |= (t.o[k] != kOOB) ? (2u << k) : 0u;
      const unsigned o00 = (unsigned)(h0 * Ws[l] + w0) * pix_bytes;   // wraps for h0 / w0 = -1; only used with its valid bit
      if (l < 2) {
        s_tap_o[(ql * 2 + l) * P + p] = make_uint2(o00, vbits);
        s_tap_c[(ql * 2 + l) * P + p] = make_float4(t.c[0], t.c[1], t.c[2], t.c[3]);
      } else {
        const int ry = h0 - by0, rx = w0 - bx0;
        const bool inbox = vbits == 0 || (ry >= 0 && ry + 1 < kBox && rx >= 0 && rx + 1 < kBox);   // (no valid corner: weights are 0)
        const int ryc = min(max(ry, 0), kBox - 2), rxc = min(max(rx, 0), kBox - 2);
        const unsigned a00 = (unsigned)(ryc * kBox + rxc);   // box pixel index
        uint4 w0_;
        w0_.x = a00 | ((a00 + 1) << 16);
        w0_.y = (a00 + kBox) | ((a00 + kBox + 1) << 16);
        w0_.z = o00;
        w0_.w = (inbox ? 0u : 1u) | vbits;
        uint4 *rec = reinterpret_cast<uint4 *>(&s_box_tap[ql * P + p]);
        rec[0] = w0_;
        rec[1] = make_uint4(__float_as_uint(t.c[0]), __float_as_uint(t.c[1]), __float_as_uint(t.c[2]), __float_as_uint(t.c[3]));
      }
    }
  }
  // ---- park the box
#pragma unroll
  for (int i = 0; i < kFill; ++i) {
    const int idx = tid + 256 * i;
    // 16-byte piece j of box pixel px sits at piece position j ^ (px & 7): the 8 lane groups of a wave read the SAME piece
    // index of 8 different pixels in one ds_read_b128, and unswizzled those all fall on the same 4 banks
    const int px = idx / G, piece = idx - px * G;
    if (idx < kBox * kBox * G) *reinterpret_cast<dvis_v4u *>(&s_box[(px * G + (piece ^ (px & 7))) * 4]) = fill[i];
  }
  __syncthreads();

  const int lane = tid & 63, wv = tid >> 6;
  const int g = lane / G, j = lane - g * G;
  const unsigned lane_bytes = (unsigned)j * 16u;
  float *const out_frame = out + ((size_t)n * Lq * M + m) * D;
  const unsigned row_bytes[L] = {(unsigned)Ws[0] * pix_bytes, (unsigned)Ws[1] * pix_bytes, (unsigned)Ws[2] * pix_bytes};

  auto corner_offsets = [&](unsigned o00, unsigned vbits, unsigned rowb, unsigned (&o)[4]) {
    o[0] = (vbits & 2u) ? o00 + lane_bytes : kOOB;
    o[1] = (vbits & 4u) ? o00 + pix_bytes + lane_bytes : kOOB;
    o[2] = (vbits & 8u) ? o00 + rowb + lane_bytes : kOOB;
    o[3] = (vbits & 16u) ? o00 + rowb + pix_bytes + lane_bytes : kOOB;
  };

#pragma unroll 1
  for (int it = 0; it < kSlots / (4 * GPW); ++it) {
    const int ql = (it * 4 + wv) * GPW + g;
    const int qlc = ql < kQS ? ql : kQS - 1;   // idle slots redo the last query (no store)
    const int q = slot_query(ql);
    // Two workgroups per CU (the box) = 8 waves: the latency has to be hidden INSIDE a wave.  All 32 global corner
    // loads of levels 0 and 1 and the 16 box reads of level 2 of a query are issued before anything is consumed
    // (192 VGPRs of landing space); weights are re-read from LDS at consume time.
    dvis_v4u rg[2][P][4];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (EXP & 4) continue;
        const uint2 t = s_tap_o[(qlc * 2 + l) * P + p];
        unsigned o[4];
        corner_offsets(t.x, t.y, row_bytes[l], o);
#pragma unroll
        for (int k = 0; k < 4; ++k) rg[l][p][k] = __builtin_amdgcn_raw_buffer_load_b128(rs[l], o[k], 0, 0);
      }
    dvis_v4u rb[P][4];
    unsigned any_out = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint4 ta = *reinterpret_cast<const uint4 *>(&s_box_tap[qlc * P + p]);   // a[0..3], o00, flags
      const unsigned a4[4] = {ta.x & 0xffffu, ta.x >> 16, ta.y & 0xffffu, ta.y >> 16};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        rb[p][k] = (EXP & 2) ? dvis_v4u{0u, 0u, 0u, a4[k]}
                             : *reinterpret_cast<const dvis_v4u *>(&s_box[(a4[k] * G + (j ^ (a4[k] & 7))) * 4]);
      any_out |= ta.w & 1u;
    }
    if (__any(any_out != 0)) {   // wave-uniform, rare: samples that left the box
#pragma unroll
      for (int p = 0; p < P; ++p) {   // (static indices: a run-time p would put rb[] into scratch)
        const uint4 ta = *reinterpret_cast<const uint4 *>(&s_box_tap[qlc * P + p]);
        unsigned o[4];
        corner_offsets(ta.z, ta.w, row_bytes[2], o);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const dvis_v4u gl = __builtin_amdgcn_raw_buffer_load_b128(rs[2], (ta.w & 1u) ? o[k] : kOOB, 0, 0);
          rb[p][k] = (ta.w & 1u) ? gl : rb[p][k];
        }
      }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto consume = [&](const dvis_v4u(&r)[4], const float4 c, const float aw) {
      acc[0] = accumulate_sample(acc[0], c.x, c.y, c.z, c.w, __uint_as_float(r[0].x), __uint_as_float(r[1].x), __uint_as_float(r[2].x), __uint_as_float(r[3].x), aw);
      acc[1] = accumulate_sample(acc[1], c.x, c.y, c.z, c.w, __uint_as_float(r[0].y), __uint_as_float(r[1].y), __uint_as_float(r[2].y), __uint_as_float(r[3].y), aw);
      acc[2] = accumulate_sample(acc[2], c.x, c.y, c.z, c.w, __uint_as_float(r[0].z), __uint_as_float(r[1].z), __uint_as_float(r[2].z), __uint_as_float(r[3].z), aw);
      acc[3] = accumulate_sample(acc[3], c.x, c.y, c.z, c.w, __uint_as_float(r[0].w), __uint_as_float(r[1].w), __uint_as_float(r[2].w), __uint_as_float(r[3].w), aw);
    };
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (EXP & 4) continue;
        consume(rg[l][p], s_tap_c[(qlc * 2 + l) * P + p], s_aw[qlc * LP + l * P + p]);
      }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float4 c = *reinterpret_cast<const float4 *>(&s_box_tap[qlc * P + p].c[0]);
      consume(rb[p], c, s_aw[qlc * LP + 2 * P + p]);
    }
    if (q >= 0) {
      float *dst = out_frame + (size_t)q * MD + 4 * j;
      *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int msda_cell(const float *value, const int64_t *shapes, const int64_t *ls,
    const float *ref, int nref, const float *off, int64_t off_stride, const float *lg, int64_t lg_stride, int N, int S,
    int M, int Lq, int H2, int W2, float *out, void *stream, int exp) {
  const int cells_x = (W2 + 7) / 8, cells_y = (H2 + 7) / 8;
  dim3 grid(M, cells_x * cells_y, N);
#define RUNC(E) hipLaunchKernelGGL((msda_fwd_cell<32, 4, E>), grid, dim3(256), 0, (hipStream_t)stream, value, shapes, ls, off, off_stride, lg, lg_stride, ref, nref, S, M, Lq, out, cells_x)
  switch (exp) { case 0: RUNC(0); break; case 1: RUNC(1); break; case 2: RUNC(2); break; case 3: RUNC(3); break; case 4: RUNC(4); break; case 6: RUNC(6); break; case 7: RUNC(7); break; default: return 1; }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
