// Mask-logit contraction of the masked-attention decoder — gfx950, exact-fp32 MFMA.
//
// Replaces, in forward_prediction_heads (dvis_Plus/video_mask2former_transformer_decoder.py:358-374):
//   MODE 0  outputs_mask = einsum("bqc,bchw->bqhw", mask_embed, mask_features)                       (:363)
//   MODE 1  attn_mask = (F.interpolate(outputs_mask, size, bilinear, align_corners=False).sigmoid() < 0.5)  (:367-371)
//           + the "row blocked everywhere -> allow everything" reset done with a host-syncing torch.where (:297)
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
// Tiling: a wave owns 16 queries (A operand = its 16 x C slice of mask_embed, resident in VGPRs for the whole
// block) and streams 16-pixel tiles; 8 waves = up to 128 queries per pass.  The B operand (features) is staged
// through LDS in 64-channel x 128-pixel stages, double-buffered (2 x 36 KB of dynamic LDS): the next stage (of this or of the next tile) is
// fetched into registers at the top of the current stage's MFMAs and written to the other buffer behind them, so a
// stage costs ONE barrier and the MFMA queue only drains there (the single-buffered form — write, barrier, MFMAs,
// barrier, at one workgroup per CU — ran at twice its MFMA time).  The K index is permuted — lane group g of the MFMA sums channels [g*CQ, (g+1)*CQ) — so each lane's
// A-slice is one contiguous run of mask_embed.
// For MODE 1 a 16-pixel tile is 2 centre rows x (4 outputs x 2 centre columns), so the 4 addends of an output
// sit in lanes j, j^1, j^8 of the accumulator layout and are combined with two wave shuffles.
#include "dvis_common.h"

namespace {

// channels per LDS stage (16 k-steps x 4 lane groups).  128-channel stages (half the barriers, 147 KB of LDS, 190 VGPRs)
// were measured neutral: attention masks of the three levels 2087 vs 2017 us, full logits 1295 vs 1286 us.
constexpr int kKC = 64;
constexpr int kU = kKC / 4;     // k-steps per stage
constexpr int kNPix = 128;      // source pixels per stage = 8 MFMA pixel tiles
constexpr int kLStride = 144;   // floats per LDS row: rows of lane groups 0/1 land in different bank halves
constexpr int kMaxStages = 256 / kKC;   // C <= 256
constexpr size_t kLdsBytes = 2 * kKC * kLStride * sizeof(float);   // two stage buffers

// Lane exchanges of the MODE 1 epilogue as DPP modifiers (VALU, no LDS round trip): __shfl_xor compiled to 64
// ds_bpermute_b32 per tile, each followed by a full lgkmcnt wait — ~6000 clk per tile and wave with the MFMA pipe idle.
__device__ __forceinline__ float lane_xor1(float v) {   // quad_perm [1,0,3,2]
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_xor8(float v) {   // row_ror:8 — within a row of 16 lanes (j + 8) % 16 == j ^ 8
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}

template <int MODE>
__global__ __launch_bounds__(512) void mask_gemm_kernel(
    const float *__restrict__ embed, const float *__restrict__ feat, int Q, int qbeg, int C, int CQ, int NS, int H, int W,
    int h, int w, int sfac, int ntiles, int tiles_per_block, float *__restrict__ out_logits,
    uint8_t *__restrict__ out_mask, int *__restrict__ allowed_count) {
  extern __shared__ float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const size_t HW = (size_t)H * W;
  const int OHW = h * w;
  const float *featb = feat + (size_t)b * C * HW;

  // ---- A operand: this wave's 16 x C slice(s) of mask_embed, lane (i=j, g) holds channels [g*CQ, g*CQ + NS*16)
  constexpr int QT = 1;   // one 16-query tile per wave; Q > 128 is covered by further launches (qbeg)
  float efrag[QT][kMaxStages * kU];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qbeg + (wv + 8 * qt) * 16 + j;
    const float *erow = embed + ((size_t)b * Q + (q < Q ? q : 0)) * C;
#pragma unroll
    for (int k = 0; k < kMaxStages * kU; ++k) {
      const int c = g * CQ + k;
      efrag[qt][k] = (q < Q && k < NS * kU && c < C) ? erow[c] : 0.f;
    }
  }

  // staging geometry: thread -> (pixel pair pp, LDS rows rbase + 8*i)
  const int pp = tid & 63, rbase = tid >> 6;

  // source pixel offset of this thread's pair (2 consecutive columns of the stage) for a tile; ok = the pair exists
  auto tile_src = [&](int tile, bool live, unsigned &src, bool &ok0, bool &ok1) {
    if (MODE == 0) {
      const size_t p = (size_t)tile * kNPix + 2 * pp;
      src = (unsigned)p;
      ok0 = live & (p < HW);
      ok1 = live & (p + 1 < HW);
    } else {
      const int pt = pp >> 3, j2 = pp & 7;
      const int o = tile * 32 + pt * 4 + (j2 & 3);
      const int oi = o / w, oj = o - oi * w;
      const int y = oi * sfac + sfac / 2 - 1 + (j2 >> 2);
      const int x = oj * sfac + sfac / 2 - 1;
      src = (unsigned)(y * W + x);
      ok0 = ok1 = live & (o < OHW);
    }
  };
  // Buffer loads: an out-of-range offset reads 0 in hardware, so no instruction after a load depends on a predicate.
  // (With flat loads + selects, and with the loads under uniform branches, the compiler put vmcnt(0) waits in front of
  // the stage's MFMAs — phi copies of loaded values — and the "prefetch" only overlapped across waves.)
  const __amdgpu_buffer_rsrc_t rs = dvis_make_rsrc_uniform(featb, (unsigned)((size_t)C * HW * sizeof(float)));
  constexpr unsigned kOOB = 0xFFFFFF00u;   // beyond any frame slab (host checks C * HW * 4 < 2^32 - 256)
  const unsigned chan_bytes = (unsigned)(HW * sizeof(float));
  float pre0[kKC / 8], pre1[kKC / 8];
  auto prefetch = [&](int t, unsigned src, bool ok0, bool ok1) {
#pragma unroll
    for (int i = 0; i < kKC / 8; ++i) {
      const int r = rbase + 8 * i;                 // LDS row: u = r >> 2, lane group = r & 3
      const int c = (r & 3) * CQ + kU * t + (r >> 2);
      const bool cv = (c < C) & ((r >> 2) + kU * t < CQ);
      const unsigned off = (unsigned)c * chan_bytes + src * 4u;
      pre0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (cv & ok0) ? off : kOOB, 0, 0));
      pre1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (cv & ok1) ? off + 4u : kOOB, 0, 0));
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kKC / 8; ++i)
      *reinterpret_cast<float2 *>(&lds[buf * (kKC * kLStride) + (rbase + 8 * i) * kLStride + 2 * pp]) =
          make_float2(pre0[i], pre1[i]);
  };

  const int tile0 = blockIdx.x * tiles_per_block;
  if (tile0 >= ntiles) return;   // uniform
  const bool mine = qbeg + (wv * 16) < Q;   // this wave's q-tile holds real queries
  unsigned src0;
  bool ok0, ok1;
  tile_src(tile0, true, src0, ok0, ok1);
  prefetch(0, src0, ok0, ok1);
  stage_store(0);
  __syncthreads();
  int buf = 0;

  for (int tt = 0; tt < tiles_per_block; ++tt) {
    const int tile = tile0 + tt;
    if (tile >= ntiles) break;   // uniform

    dvis_f4 acc[QT][8];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) acc[qt][pt] = dvis_f4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int t = 0; t < kMaxStages; ++t) {
      if (t < NS) {
        // fetch what the NEXT stage needs — the next channels of this tile or the first channels of the next tile
        // (nothing: all offsets out of range, zeros parked in a buffer nobody reads) — always the same instructions
        int nt = t + 1;
        if (nt == NS) {   // uniform; integer set-up only, no load under the branch
          nt = 0;
          tile_src(tile + 1, (tt + 1 < tiles_per_block) & (tile + 1 < ntiles), src0, ok0, ok1);
        }
        prefetch(nt, src0, ok0, ok1);
        const float *cur = lds + buf * (kKC * kLStride);
        if (mine) {
          // B fragments one k-step ahead: the compiler's own order was read, wait lgkmcnt(0), 2 MFMAs, read, ... — the
          // LDS latency exposed every 64 clk of MFMA work
          float bv[2][8];
#pragma unroll
          for (int pt = 0; pt < 8; ++pt) bv[0][pt] = cur[g * kLStride + pt * 16 + j];
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            if (u + 1 < kU) {
#pragma unroll
              for (int pt = 0; pt < 8; ++pt) bv[(u + 1) & 1][pt] = cur[((u + 1) * 4 + g) * kLStride + pt * 16 + j];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the reads above this k-step's MFMAs (the scheduler sinks them)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
              for (int pt = 0; pt < 8; ++pt)
                acc[qt][pt] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(efrag[qt][t * kU + u], bv[u & 1][pt], acc[qt][pt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // ... and park it in the other buffer (last read one stage ago, before the previous barrier)
        stage_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
      }
    }

    // ---- epilogue.  Accumulator layout: column (pixel) = lane & 15, row (query) = (lane >> 4) * 4 + reg.
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int qb = qbeg + (wv + 8 * qt) * 16 + g * 4;
      if (MODE == 0) {
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
          const size_t p = (size_t)tile * kNPix + pt * 16 + j;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (qb + r < Q && p < HW) out_logits[((size_t)b * Q + qb + r) * HW + p] = acc[qt][pt][r];
        }
      } else {
        int cnt[4] = {0, 0, 0, 0};
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
          const int o = tile * 32 + pt * 4 + (j >> 1);
          const bool writer = (j & 1) == 0 && j < 8 && o < OHW;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[qt][pt][r];
            const float hs = v + lane_xor1(v);           // (a + b) resp. (c + d)
            const float s4 = hs + lane_xor8(hs);         // (a + b) + (c + d)   [x 0.25 > 0 dropped]
            const bool blocked = s4 < 0.f;
            const unsigned long long bal = __ballot(writer && !blocked);
            cnt[r] += __popc((unsigned)((bal >> (16 * g)) & 0xffffull));
            if (writer && qb + r < Q) out_mask[((size_t)b * Q + qb + r) * OHW + o] = blocked ? 1 : 0;
          }
        }
        if (j == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (qb + r < Q && cnt[r] > 0) atomicAdd(&allowed_count[(size_t)b * Q + qb + r], cnt[r]);
        }
      }
    }
  }
}

int launch(int mode, const float *embed, const float *feat, int B, int Q, int C, int H, int W, int h, int w,
           float *out_logits, uint8_t *out_mask, int *allowed, hipStream_t st) {
  const int CQ = ((C + 3) / 4 + kU - 1) / kU * kU;
  const int NS = CQ / kU;
  const int sfac = mode == 1 ? H / h : 1;
  const long long npx = mode == 1 ? (long long)h * w : (long long)H * W;
  const int per = mode == 1 ? 32 : kNPix;
  const int ntiles = (int)((npx + per - 1) / per);
  int tpb = (int)(((long long)ntiles * B + 2047) / 2048);
  tpb = tpb < 1 ? 1 : (tpb > 16 ? 16 : tpb);
  const dim3 grid((ntiles + tpb - 1) / tpb, B), block(512);
  static bool lds_opt_in = false;   // > 64 KB of dynamic LDS needs the opt-in once per process
  if (!lds_opt_in) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mask_gemm_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mask_gemm_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) {
      dvis_set_error("mask_gemm: hipFuncSetAttribute(max dynamic LDS): %s", hipGetErrorString(e));
      return DVIS_E_LAUNCH;
    }
    lds_opt_in = true;
  }
  for (int qbeg = 0; qbeg < Q; qbeg += 128) {   // 8 waves x 16 queries per launch
    if (mode == 0)
      hipLaunchKernelGGL((mask_gemm_kernel<0>), grid, block, kLdsBytes, st, embed, feat, Q, qbeg, C, CQ, NS, H, W, h, w, sfac,
                         ntiles, tpb, out_logits, out_mask, allowed);
    else
      hipLaunchKernelGGL((mask_gemm_kernel<1>), grid, block, kLdsBytes, st, embed, feat, Q, qbeg, C, CQ, NS, H, W, h, w, sfac,
                         ntiles, tpb, out_logits, out_mask, allowed);
  }
  return dvis_check_launch("mask_gemm_kernel");
}

}  // namespace

DVIS_EXPORT int dvis_mask_logits(const float *embed, const float *feat, int B, int Q, int C, int64_t HW, float *out,
                                 void *stream) {
  DVIS_REQUIRE(B >= 0 && Q > 0 && C > 0 && HW > 0, "mask_logits: bad sizes");
  if (B == 0) return DVIS_OK;
  DVIS_REQUIRE(embed && feat && out, "mask_logits: null pointer");
  DVIS_REQUIRE(C <= 256, "mask_logits: supports C <= 256 (got C=%d)", C);
  DVIS_REQUIRE(HW < (1ll << 31) && B <= 65535, "mask_logits: HW / B too large");
  DVIS_REQUIRE((long long)C * HW * 4 < 0xFFFFFF00ll, "mask_logits: one frame of mask_features must stay below 4 GiB");
  return launch(0, embed, feat, B, Q, C, 1, (int)HW, 0, 0, out, nullptr, nullptr, (hipStream_t)stream);
}

DVIS_EXPORT int dvis_attn_mask(const float *embed, const float *feat, int B, int Q, int C, int H, int W, int h, int w,
                               uint8_t *mask, int32_t *allowed_count, void *stream) {
  DVIS_REQUIRE(B >= 0 && Q > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0, "attn_mask: bad sizes");
  if (B == 0) return DVIS_OK;
  DVIS_REQUIRE(embed && feat && mask && allowed_count, "attn_mask: null pointer");
  DVIS_REQUIRE(C <= 256, "attn_mask: supports C <= 256 (got C=%d)", C);
  DVIS_REQUIRE(H % h == 0 && W % w == 0 && H / h == W / w && (H / h) % 2 == 0,
               "attn_mask: needs an even integer down-sizing factor (H=%d W=%d -> h=%d w=%d)", H, W, h, w);
  DVIS_REQUIRE(B <= 65535, "attn_mask: B too large");
  DVIS_REQUIRE((long long)C * H * W * 4 < 0xFFFFFF00ll, "attn_mask: one frame of mask_features must stay below 4 GiB");
  hipError_t e = hipMemsetAsync(allowed_count, 0, (size_t)B * Q * sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) {
    dvis_set_error("attn_mask: hipMemsetAsync: %s", hipGetErrorString(e));
    return DVIS_E_LAUNCH;
  }
  return launch(1, embed, feat, B, Q, C, H, W, h, w, nullptr, mask, allowed_count, (hipStream_t)stream);
}
