// Mask-logit contraction of the masked-attention decoder — gfx950, exact-fp32 MFMA.
//
// Replaces, in forward_prediction_heads (dvis_Plus/video_mask2former_transformer_decoder.py:358-374):
//   MODE 0  outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)                       (:363)
//   MODE 1  attn_mask = (F.interpolate(outputs_mask, size, bilinear, align_corners=False).sigmoid() < 0.5)  (:367-371)
//           + the "row blocked everywhere -> allow everything" reset done with a host-syncing torch.where (:297)
//   MODE 2  the same mask from a POOLED feature map (round 5): the down-sizing is linear, so the four centre pixels of every
//           s x s block are averaged ONCE per clip into three small maps (dvis_center_pool3: one read of the 1.8 GB map for
//           all nine decoder layers) and a layer contracts its level's map — 1/4 of MODE 1's products, no re-read of the
//           stride-4 map: mask = (einsum(embed, pooled) < 0).  Same value up to the rounding ORDER (average-then-contract vs
//           contract-then-average): bits differ only where |logit| is of the order of its own rounding error, where the
//           reference's own fp32 result is as arbitrary (measured: 12 of 57 960 000 bits per clip, all |logit| < 5e-6).
// MODE 1 never writes the stride-4 logits to HBM (23.6 MB per call at 720p in the reference) and emits the mask
// once per frame, not replicated over the 8 heads.
//
// Why this is exact w.r.t. the reference's order of operations (contract -> interpolate -> threshold):
//   * bilinear down-sizing by an even integer factor s with align_corners=False samples exactly between the two
//     centre pixels of each s-block in both directions (src = (dst + 0.5) * s - 0.5), all four weights are 0.5*0.5,
//     so out = 0.25 * ((a + b) + (c + d)) with (a, b) / (c, d) the centre pairs of the upper / lower centre row.
//     Only those 4 of s*s logits are ever contracted (16x less work at s = 8, 4x at s = 4).
//   * sigmoid(x) < 0.5  <=>  x < 0 (up to |x| < 6e-8 where fp32 sigmoid rounds to 0.5; logits themselves differ
//     by ~1e-5 between any two fp32 summation orders, so such pixels are ambiguous in the reference too).
//   * v_mfma_f32_16x16x4_f32 is a k-ordered fp32 fma chain: no reduced precision.
//
// Partition (round 2): by PIXELS, not by queries.  A wave owns 64 source pixels (four 16-column MFMA tiles) and ALL
// query tiles of the pass (QT <= 7 tiles of 16 = 112 queries): 28 accumulators.  Its B operand (features) is needed by
// no other wave, so it goes from global memory straight into the MFMA operand layout — lane (j, g) loads the 4 pixels
// 4j..4j+3 of channel g*CQ + u as one 16-byte word, 16 lanes = 256 contiguous bytes — with no LDS staging and no
// barrier in the main loop; the A operand (mask_embed of the frame, 112 x 256 floats) sits in LDS for the whole
// workgroup, laid out [lane group][query][64 + 4] so that the 16-byte fragment reads of a 16-lane group cover all 64
// banks.  The first form partitioned by queries (8 waves x 16 queries, A in VGPRs, B staged through LDS for all of
// them): at 100 queries one wave in eight idled and 100 of 112 issued rows were useful, and every stage cost a barrier
// (DESIGN.md section 3.3 keeps its numbers).  The K index is permuted — lane group g of the MFMA sums channels
// [g*CQ, (g+1)*CQ) — so an A fragment is one contiguous run of mask_embed.
// For MODE 1 lane j owns output pixel o0 + j and its four MFMA tiles are the addends a, b, c, d of that output: the
// down-sizing is three in-lane adds in the reference's order, no lane exchange.
// Work is cut into steps of 8 wave groups (one per wave) of one frame; a workgroup takes an equal, contiguous share of
// all steps of all frames (A is reloaded when the frame changes), one workgroup per CU.
#include "dvis_common.h"

namespace {

constexpr int kARow = 68;        // floats per LDS row of A: 64 channels + 4 (16-lane groups of b128 reads hit all banks)
constexpr int kMaxQT = 7;        // query tiles per pass: 4 * 112 * 68 * 4 B = 119 KB of LDS
// Out-of-range buffer offset.  It is combined with SCALAR offsets (the channel of a k-step), and the hardware adds the
// two in 32 bits before its range check: 0xFFFFFF00 + a channel offset wraps back into the slab (harmless here — such
// a column is never stored and its A entries are zero — but not something to rely on).  2 GiB + an offset into a
// < 2 GiB slab neither wraps nor lands inside it (host checks).
constexpr unsigned kOOB = 0x80000000u;

template <int QT> constexpr size_t lds_bytes() { return (size_t)4 * QT * 16 * kARow * sizeof(float) + QT * 16 * sizeof(int); }

// MODE 0: VEC = rows of `feat` and of the output are 16-byte aligned (HW % 4 == 0 and aligned bases) -> b128 loads/stores.
// FULLC = C == 4 * CQ (C a multiple of 32, the production 256): no per-channel validity selects in the loop.
template <int MODE, int QT, bool VEC, bool FULLC>
__global__ __launch_bounds__(512) void mask_gemm_kernel(
    const float *__restrict__ embed, const float *__restrict__ feat, int Q, int qbeg, int C, int CQ, int H, int W, int h,
    int w, int sfac, int gpf, int spf, int total_steps, float *__restrict__ out_logits, uint8_t *__restrict__ out_mask,
    int *__restrict__ allowed_count) {
  extern __shared__ float lds[];   // A: [4 lane groups][QT * 16 queries][kARow]
  constexpr int ROWS = QT * 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const size_t HW = (size_t)H * W;
  const int OHW = h * w;
  const unsigned chan_bytes = (unsigned)(HW * sizeof(float));

  const int s_lo = (int)((long long)blockIdx.x * total_steps / gridDim.x);
  const int s_hi = (int)((long long)(blockIdx.x + 1) * total_steps / gridDim.x);

  int cur_b = -1;
  // MODE 1: allowed pixels per query row, summed in LDS over the workgroup's steps of one frame, then one global atomic
  int *counts = reinterpret_cast<int *>(lds + 4 * ROWS * kARow);
  if (MODE != 0 && tid < ROWS) counts[tid] = 0;
  auto flush_counts = [&]() {   // between two barriers
    if (MODE != 0 && cur_b >= 0 && tid < ROWS) {
      const int q = qbeg + tid, c = counts[tid];
      if (q < Q && c > 0) atomicAdd(&allowed_count[(size_t)cur_b * Q + q], c);
      counts[tid] = 0;
    }
  };

  // byte offsets of this lane's 4 B columns, lane group's first channel included; a column that does not exist gets
  // an out-of-range offset (reads 0).  The channel of a k-step goes into the load's SCALAR offset.
  unsigned off[4];
  bool ok[4];
  const unsigned gbase = (unsigned)(g * CQ) * chan_bytes;
  auto set_columns = [&](int group) {
    if (MODE != 1) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const size_t p = (size_t)group * 64 + 4 * j + n;
        ok[n] = p < HW;
        off[n] = ok[n] ? gbase + (unsigned)(p * 4) : kOOB;
      }
    } else {
      const int o = group * 16 + j;
      const int oi = o / w, oj = o - oi * w;
      const int y = oi * sfac + sfac / 2 - 1, x = oj * sfac + sfac / 2 - 1;
      ok[0] = ok[1] = ok[2] = ok[3] = o < OHW;
      const unsigned o0 = gbase + (unsigned)(y * W + x) * 4u;
      off[0] = ok[0] ? o0 : kOOB;
      off[1] = ok[0] ? o0 + 4u : kOOB;
      off[2] = ok[0] ? o0 + (unsigned)W * 4u : kOOB;
      off[3] = ok[0] ? o0 + (unsigned)W * 4u + 4u : kOOB;
    }
  };
  const int ulim = C - g * CQ;   // this lane group's channels are u < ulim (FULLC: all CQ of them)
  __amdgpu_buffer_rsrc_t rs = dvis_make_rsrc_uniform(feat, 0);
  // B fragments of 4 k-steps (channels g*CQ + u0 .. + 3): [k-step][pixel tile]
  auto load_b = [&](int u0, dvis_f4(&dst)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned so = (unsigned)(u0 + i) * chan_bytes;   // uniform
      const bool cv = FULLC ? true : (u0 + i < ulim);
      if (MODE != 1 && VEC) {
        const dvis_v4u t =
            __builtin_bit_cast(dvis_v4u, __builtin_amdgcn_raw_buffer_load_b128(rs, cv ? off[0] : kOOB, so, 0));
        dst[i] = __builtin_bit_cast(dvis_f4, t);
      } else {
#pragma unroll
        for (int n = 0; n < 4; ++n)
          dst[i][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, cv ? off[n] : kOOB, so, 0));
      }
    }
  };
  dvis_f4 b0[4], b1[4];
  bool have_b0 = false;   // wave-uniform: b0 already holds k-steps 0..3 of this step's group (fetched under the last
                          // MFMAs of the previous step — otherwise both waves of a SIMD sit out the same load latency)

  for (int step = s_lo; step < s_hi; ++step) {
    const int b = step / spf, sidx = step - b * spf;
    if (b != cur_b) {   // uniform over the workgroup
      __syncthreads();   // everyone is done with the previous frame's A (and its counts)
      flush_counts();
      const int n4 = CQ >> 2, total4 = 4 * ROWS * n4;
      for (int base = 0; base < total4; base += 4 * 512) {   // 4 granules in flight per thread
        dvis_f4 v[4];
        int dst[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = base + k * 512 + tid;
          const int u4 = idx % n4, rest = idx / n4;
          const int row = rest % ROWS, gg = rest / ROWS;
          const int q = qbeg + row, c = gg * CQ + u4 * 4;
          const float *erow = embed + ((size_t)b * Q + (q < Q ? q : 0)) * C;
          dst[k] = idx < total4 ? (gg * ROWS + row) * kARow + u4 * 4 : -1;
          const bool live = idx < total4 && q < Q;
          if (FULLC) {
            v[k] = live ? *reinterpret_cast<const dvis_f4 *>(erow + c) : dvis_f4{0.f, 0.f, 0.f, 0.f};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[k][i] = (live && c + i < C) ? erow[c + i] : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (dst[k] >= 0) *reinterpret_cast<dvis_f4 *>(&lds[dst[k]]) = v[k];
      }
      __syncthreads();
      cur_b = b;
      rs = dvis_make_rsrc_uniform(feat + (size_t)b * C * HW, (unsigned)((size_t)C * HW * sizeof(float)));
    }
    const int group = sidx * 8 + wv;
    if (group >= gpf) continue;   // wave-uniform; no barrier below (have_b0 is false: only existing groups are prefetched)

    if (!have_b0) {
      set_columns(group);
      load_b(0, b0);
    }
    bool okc[4];   // `ok` moves on to the next group during the last k-steps
#pragma unroll
    for (int n = 0; n < 4; ++n) okc[n] = ok[n];
    const float *arow = lds + (g * ROWS + j) * kARow;
    auto load_a = [&](int u0, dvis_f4(&dst)[QT]) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) dst[qt] = *reinterpret_cast<const dvis_f4 *>(arow + qt * 16 * kARow + u0);
    };

    dvis_f4 acc[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[qt][n] = dvis_f4{0.f, 0.f, 0.f, 0.f};

    auto contract = [&](const dvis_f4(&af)[QT], const dvis_f4(&bf)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[qt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[qt][i], bf[i][n], acc[qt][n], 0, 0, 0);
    };

    dvis_f4 af[QT];
    have_b0 = false;
#pragma unroll 1
    for (int u0 = 0; u0 < CQ; u0 += 8) {   // CQ % 8 == 0
      load_b(u0 + 4, b1);
      load_a(u0, af);
      contract(af, b0);
      if (u0 + 8 < CQ) {   // uniform
        load_b(u0 + 8, b0);
      } else if (step + 1 < s_hi) {
        const int nb = (step + 1) / spf, ng = (step + 1 - nb * spf) * 8 + wv;
        if (nb == b && ng < gpf) {   // same frame (same A, same descriptor), and the group exists
          set_columns(ng);
          load_b(0, b0);
          have_b0 = true;
        }
      }
      load_a(u0 + 4, af);
      contract(af, b1);
    }

    // ---- epilogue.  Accumulator layout: column (pixel) = lane & 15, row (query) = (lane >> 4) * 4 + reg.
    // Buffer stores: one 32-bit lane offset + a scalar row offset (28 64-bit row pointers would cost 56 VGPRs).
    if (MODE == 0) {
      const __amdgpu_buffer_rsrc_t ro =
          dvis_make_rsrc_uniform(out_logits + (size_t)b * Q * HW, (unsigned)((size_t)Q * HW * sizeof(float)));
      const unsigned vo = (unsigned)(((size_t)(qbeg + g * 4) * HW + (size_t)group * 64 + 4 * j) * 4);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned so = (unsigned)(qt * 16 + r) * chan_bytes;   // uniform
          if (qbeg + qt * 16 + g * 4 + r >= Q) continue;
          if (VEC) {
            const dvis_f4 v = dvis_f4{acc[qt][0][r], acc[qt][1][r], acc[qt][2][r], acc[qt][3][r]};
            // (row offset in the VECTOR offset: with an SGPR offset LLVM does not pad a VALU write to the data registers
            // of a > 8-byte store, and on MI355X the store then picks up the next row's value — see conv1x1.hip)
            if (okc[0]) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvis_v4u, v), ro, vo + so, 0, 0);
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const float v = acc[qt][n][r];   // (bit_cast straight from the vector-element lvalue reads element 0)
              if (okc[n]) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, vo + 4u * n, so, 0);
            }
          }
        }
    } else if (MODE == 2) {
      // lane j holds pixels group * 64 + 4 j + n (n = 0 .. 3) of query rows qt * 16 + 4 g + r: four mask bytes = one 32-bit store
      const __amdgpu_buffer_rsrc_t ro = dvis_make_rsrc_uniform(out_mask + (size_t)b * Q * HW, (unsigned)((size_t)Q * HW));
      const unsigned vo = (unsigned)((size_t)(qbeg + g * 4) * HW + (size_t)group * 64 + 4 * j);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          unsigned packed = 0;
          int n_allowed = 0;
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const bool blocked = acc[qt][n][r] < 0.f;
            packed |= (blocked ? 1u : 0u) << (8 * n);
            const unsigned long long bal = __ballot(okc[n] && !blocked);
            n_allowed += __popc((unsigned)((bal >> (16 * g)) & 0xffffull));
          }
          if (j == 0 && n_allowed > 0) atomicAdd(&counts[qt * 16 + g * 4 + r], n_allowed);   // LDS
          if (qbeg + qt * 16 + g * 4 + r >= Q) continue;
          const unsigned so = (unsigned)((size_t)(qt * 16 + r) * HW);   // uniform
          if (VEC) {
            if (okc[0]) __builtin_amdgcn_raw_buffer_store_b32(packed, ro, vo, so, 0);
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              if (okc[n]) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((packed >> (8 * n)) & 1u), ro, vo + n, so, 0);
          }
        }
    } else {
      const __amdgpu_buffer_rsrc_t ro = dvis_make_rsrc_uniform(out_mask + (size_t)b * Q * OHW, (unsigned)((size_t)Q * OHW));
      const unsigned vo = (unsigned)((qbeg + g * 4) * OHW + group * 16 + j);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s4 = (acc[qt][0][r] + acc[qt][1][r]) + (acc[qt][2][r] + acc[qt][3][r]);   // [x 0.25 > 0 dropped]
          const bool blocked = s4 < 0.f;
          const unsigned long long bal = __ballot(okc[0] && !blocked);
          const int n_allowed = __popc((unsigned)((bal >> (16 * g)) & 0xffffull));
          if (j == 0 && n_allowed > 0) atomicAdd(&counts[qt * 16 + g * 4 + r], n_allowed);   // LDS
          if (okc[0] && qbeg + qt * 16 + g * 4 + r < Q)
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(blocked ? 1 : 0), ro, vo, (unsigned)((qt * 16 + r) * OHW), 0);
        }
    }
  }
  __syncthreads();
  flush_counts();
}

template <int MODE, int QT, bool VEC, bool FULLC>
int launch_one(const float *embed, const float *feat, int B, int Q, int qbeg, int C, int H, int W, int h, int w,
               float *out_logits, uint8_t *out_mask, int *allowed, hipStream_t st) {
  static DvisLdsOptIn opted;   // per kernel instantiation, per device
  if (const int rc = dvis_lds_opt_in(reinterpret_cast<const void *>(&mask_gemm_kernel<MODE, QT, VEC, FULLC>),
                                     lds_bytes<QT>(), &opted, "mask_gemm"))
    return rc;
  const int CQ = ((C + 3) / 4 + 7) / 8 * 8;
  const int sfac = MODE == 1 ? H / h : 1;
  const long long npx = MODE == 1 ? (long long)h * w : (long long)H * W;
  const int per = MODE == 1 ? 16 : 64;
  const int gpf = (int)((npx + per - 1) / per);
  const int spf = (gpf + 7) / 8;
  const long long total = (long long)B * spf;
  // one workgroup per CU at QT = 7 (LDS, 2 waves per SIMD); the small forms fit two
  const int nwg = (int)(total < (QT == kMaxQT ? 256 : 512) ? total : (QT == kMaxQT ? 256 : 512));
  hipLaunchKernelGGL((mask_gemm_kernel<MODE, QT, VEC, FULLC>), dim3(nwg), dim3(512), lds_bytes<QT>(), st, embed, feat, Q, qbeg, C,
                     CQ, H, W, h, w, sfac, gpf, spf, (int)total, out_logits, out_mask, allowed);
  return DVIS_OK;
}

template <int MODE, bool VEC>
int launch(const float *embed, const float *feat, int B, int Q, int C, int H, int W, int h, int w, float *out_logits,
           uint8_t *out_mask, int *allowed, hipStream_t st) {
  const int tiles = (Q + 15) / 16;
  const int passes = (tiles + kMaxQT - 1) / kMaxQT;
  const int per_pass = (tiles + passes - 1) / passes;   // 200 queries: 13 tiles -> 7 + 6
  for (int pass = 0, t0 = 0; pass < passes; ++pass, t0 += per_pass) {
    const int nt = tiles - t0 < per_pass ? tiles - t0 : per_pass;
    const int qbeg = t0 * 16;
    int rc;
#define DVIS_MG(QT_)                                                                                              \
  (C % 32 == 0 ? launch_one<MODE, QT_, VEC, true>(embed, feat, B, Q, qbeg, C, H, W, h, w, out_logits, out_mask,  \
                                                   allowed, st)                                                   \
               : launch_one<MODE, QT_, false, false>(embed, feat, B, Q, qbeg, C, H, W, h, w, out_logits, out_mask, \
                                                     allowed, st))
    if (nt <= 2)
      rc = DVIS_MG(2);
    else if (nt <= 4)
      rc = DVIS_MG(4);
    else
      rc = DVIS_MG(7);
#undef DVIS_MG
    if (rc != DVIS_OK) return rc;
  }
  return dvis_check_launch("mask_gemm_kernel");
}

// The four centre pixels of every s x s block of a map, s = 2, 4, 8, averaged in the reference's order ((a + b) + (c + d)) * 0.25
// — what F.interpolate(bilinear, align_corners=False) by an even integer factor samples — for all three decoder levels in ONE
// read of the map.  One thread = one 8 x 8 block of one plane (lanes run along x: 32 contiguous bytes per lane and row).
__global__ __launch_bounds__(256) void center_pool3_kernel(const float *__restrict__ f, float *__restrict__ p2,
                                                           float *__restrict__ p4, float *__restrict__ p8, int H, int W,
                                                           long long blocks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= blocks) return;
  const int bw = W >> 3, bh = H >> 3;
  const int bx = (int)(i % bw);
  const long long r = i / bw;
  const int by = (int)(r % bh);
  const long long plane = r / bh;
  const float *src = f + (plane * H + by * 8) * (long long)W + bx * 8;
  float v[8][8];
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const dvis_f4 lo = *reinterpret_cast<const dvis_f4 *>(src + (long long)y * W);
    const dvis_f4 hi = *reinterpret_cast<const dvis_f4 *>(src + (long long)y * W + 4);
    v[y][0] = lo[0]; v[y][1] = lo[1]; v[y][2] = lo[2]; v[y][3] = lo[3];
    v[y][4] = hi[0]; v[y][5] = hi[1]; v[y][6] = hi[2]; v[y][7] = hi[3];
  }
  auto avg = [&](int y, int x) { return ((v[y][x] + v[y][x + 1]) + (v[y + 1][x] + v[y + 1][x + 1])) * 0.25f; };
  float *o2 = p2 + (plane * (H >> 1) + by * 4) * (long long)(W >> 1) + bx * 4;      // s = 2: 4 x 4 outputs
#pragma unroll
  for (int k = 0; k < 4; ++k)
    *reinterpret_cast<dvis_f4 *>(o2 + (long long)k * (W >> 1)) = dvis_f4{avg(2 * k, 0), avg(2 * k, 2), avg(2 * k, 4), avg(2 * k, 6)};
  float *o4 = p4 + (plane * (H >> 2) + by * 2) * (long long)(W >> 2) + bx * 2;      // s = 4: rows 4 i + 1, columns 4 j + 1
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    o4[(long long)k * (W >> 2)] = avg(4 * k + 1, 1);
    o4[(long long)k * (W >> 2) + 1] = avg(4 * k + 1, 5);
  }
  p8[(plane * bh + by) * (long long)bw + bx] = avg(3, 3);                           // s = 8: rows 3, 4; columns 3, 4
}

}  // namespace

DVIS_EXPORT int dvis_center_pool3(const float *feat, int64_t planes, int H, int W, float *p2, float *p4, float *p8, void *stream) {
  DVIS_REQUIRE(planes >= 0 && H > 0 && W > 0, "center_pool3: bad sizes");
  if (planes == 0) return DVIS_OK;
  DVIS_REQUIRE(feat && p2 && p4 && p8, "center_pool3: null pointer");
  DVIS_REQUIRE(H % 8 == 0 && W % 8 == 0 && (((uintptr_t)feat | (uintptr_t)p2) & 15) == 0,
               "center_pool3: H, W multiples of 8 and 16-byte aligned maps (H=%d W=%d)", H, W);
  const long long blocks = (long long)planes * (H / 8) * (W / 8);
  DVIS_REQUIRE((blocks + 255) / 256 < (1ll << 31), "center_pool3: grid too large");
  hipLaunchKernelGGL(center_pool3_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat, p2, p4, p8,
                     H, W, blocks);
  return dvis_check_launch("center_pool3_kernel");
}

DVIS_EXPORT int dvis_attn_mask_pooled(const float *embed, const float *pooled, int B, int Q, int C, int h, int w, uint8_t *mask,
                                      int32_t *allowed_count, void *stream) {
  DVIS_REQUIRE(B >= 0 && Q > 0 && C > 0 && h > 0 && w > 0, "attn_mask_pooled: bad sizes");
  if (B == 0) return DVIS_OK;
  DVIS_REQUIRE(embed && pooled && mask && allowed_count, "attn_mask_pooled: null pointer");
  DVIS_REQUIRE(C <= 256, "attn_mask_pooled: supports C <= 256 (got C=%d)", C);
  const int64_t HW = (int64_t)h * w;
  DVIS_REQUIRE((long long)((C + 31) / 32 * 32) * HW * 4 < (1ll << 31) && (long long)Q * HW < (1ll << 31),
               "attn_mask_pooled: one frame of the pooled map / of the mask must stay below 2 GiB");
  if (const int rc = dvis_zero_words(allowed_count, (size_t)B * Q, (hipStream_t)stream, "attn_mask_pooled: zero counts")) return rc;
  const bool vec = HW % 4 == 0 && (((uintptr_t)pooled | (uintptr_t)mask) & 15) == 0;
  return vec ? launch<2, true>(embed, pooled, B, Q, C, h, w, h, w, nullptr, mask, allowed_count, (hipStream_t)stream)
             : launch<2, false>(embed, pooled, B, Q, C, h, w, h, w, nullptr, mask, allowed_count, (hipStream_t)stream);
}

DVIS_EXPORT int dvis_mask_logits(const float *embed, const float *feat, int B, int Q, int C, int64_t HW, float *out,
                                 void *stream) {
  DVIS_REQUIRE(B >= 0 && Q > 0 && C > 0 && HW > 0, "mask_logits: bad sizes");
  if (B == 0) return DVIS_OK;
  DVIS_REQUIRE(embed && feat && out, "mask_logits: null pointer");
  DVIS_REQUIRE(C <= 256, "mask_logits: supports C <= 256 (got C=%d)", C);
  DVIS_REQUIRE(HW < (1ll << 31) && (long long)Q * HW * 4 < 0xFFFFFF00ll, "mask_logits: one frame of logits must stay below 4 GiB");
  DVIS_REQUIRE((long long)((C + 31) / 32 * 32) * HW * 4 < (1ll << 31),
               "mask_logits: one frame of mask_features must stay below 2 GiB");
  const bool vec = HW % 4 == 0 && ((uintptr_t)feat | (uintptr_t)out) % 16 == 0;
  return vec ? launch<0, true>(embed, feat, B, Q, C, 1, (int)HW, 0, 0, out, nullptr, nullptr, (hipStream_t)stream)
             : launch<0, false>(embed, feat, B, Q, C, 1, (int)HW, 0, 0, out, nullptr, nullptr, (hipStream_t)stream);
}

DVIS_EXPORT int dvis_attn_mask(const float *embed, const float *feat, int B, int Q, int C, int H, int W, int h, int w,
                               uint8_t *mask, int32_t *allowed_count, void *stream) {
  DVIS_REQUIRE(B >= 0 && Q > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0, "attn_mask: bad sizes");
  if (B == 0) return DVIS_OK;
  DVIS_REQUIRE(embed && feat && mask && allowed_count, "attn_mask: null pointer");
  DVIS_REQUIRE(C <= 256, "attn_mask: supports C <= 256 (got C=%d)", C);
  DVIS_REQUIRE(H % h == 0 && W % w == 0 && H / h == W / w && (H / h) % 2 == 0,
               "attn_mask: needs an even integer down-sizing factor (H=%d W=%d -> h=%d w=%d)", H, W, h, w);
  DVIS_REQUIRE((long long)((C + 31) / 32 * 32) * H * W * 4 < (1ll << 31),
               "attn_mask: one frame of mask_features must stay below 2 GiB");
  if (const int rc = dvis_zero_words(allowed_count, (size_t)B * Q, (hipStream_t)stream, "attn_mask: zero counts")) return rc;
  return launch<1, false>(embed, feat, B, Q, C, H, W, h, w, nullptr, mask, allowed_count, (hipStream_t)stream);
}
