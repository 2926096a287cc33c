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
// through LDS in 64-channel x 128-pixel stages (register prefetch of the next stage under the current stage's
// MFMAs).  The K index is permuted — lane group g of the MFMA sums channels [g*CQ, (g+1)*CQ) — so each lane's
// A-slice is one contiguous run of mask_embed.
// For MODE 1 a 16-pixel tile is 2 centre rows x (4 outputs x 2 centre columns), so the 4 addends of an output
// sit in lanes j, j^1, j^8 of the accumulator layout and are combined with two wave shuffles.
#include "dvis_common.h"

namespace {

constexpr int kKC = 64;         // channels per LDS stage (16 k-steps x 4 lane groups)
constexpr int kNPix = 128;      // source pixels per stage = 8 MFMA pixel tiles
constexpr int kLStride = 144;   // floats per LDS row: rows of lane groups 0/1 land in different bank halves
constexpr int kMaxStages = 4;   // C <= 256

template <int MODE>
__global__ __launch_bounds__(512) void mask_gemm_kernel(
    const float *__restrict__ embed, const float *__restrict__ feat, int Q, int qbeg, int C, int CQ, int NS, int H, int W,
    int h, int w, int sfac, int ntiles, int tiles_per_block, float *__restrict__ out_logits,
    uint8_t *__restrict__ out_mask, int *__restrict__ allowed_count) {
  __shared__ float lds[kKC * kLStride];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const size_t HW = (size_t)H * W;
  const int OHW = h * w;
  const float *featb = feat + (size_t)b * C * HW;

  // ---- A operand: this wave's 16 x C slice(s) of mask_embed, lane (i=j, g) holds channels [g*CQ, g*CQ + NS*16)
  constexpr int QT = 1;   // one 16-query tile per wave; Q > 128 is covered by further launches (qbeg)
  float efrag[QT][kMaxStages * 16];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qbeg + (wv + 8 * qt) * 16 + j;
    const float *erow = embed + ((size_t)b * Q + (q < Q ? q : 0)) * C;
#pragma unroll
    for (int k = 0; k < kMaxStages * 16; ++k) {
      const int c = g * CQ + k;
      efrag[qt][k] = (q < Q && k < NS * 16 && c < C) ? erow[c] : 0.f;
    }
  }

  // staging geometry: thread -> (pixel pair pp, LDS rows rbase + 8*i)
  const int pp = tid & 63, rbase = tid >> 6;

  for (int tt = 0; tt < tiles_per_block; ++tt) {
    const int tile = blockIdx.x * tiles_per_block + tt;
    if (tile >= ntiles) break;   // uniform

    // source pixel offsets of this thread's pair (2 consecutive columns of the stage)
    size_t src0;
    bool ok0, ok1, vec2 = false;
    if (MODE == 0) {
      const size_t p = (size_t)tile * kNPix + 2 * pp;
      src0 = p;
      ok0 = p < HW;
      ok1 = p + 1 < HW;
    } else {
      const int pt = pp >> 3, j2 = pp & 7;
      const int o = tile * 32 + pt * 4 + (j2 & 3);
      const int oi = o / w, oj = o - oi * w;
      const int y = oi * sfac + sfac / 2 - 1 + (j2 >> 2);
      const int x = oj * sfac + sfac / 2 - 1;
      src0 = (size_t)y * W + x;
      ok0 = ok1 = o < OHW;
    }
    (void)vec2;

    dvis_f4 acc[QT][8];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) acc[qt][pt] = dvis_f4{0.f, 0.f, 0.f, 0.f};

    float pre0[8], pre1[8];
    auto prefetch = [&](int t) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = rbase + 8 * i;                 // LDS row: u = r >> 2, lane group = r & 3
        const int c = (r & 3) * CQ + 16 * t + (r >> 2);
        const bool cv = c < C && (r >> 2) + 16 * t < CQ;
        const float *src = featb + (size_t)(cv ? c : 0) * HW + src0;
        pre0[i] = (cv && ok0) ? src[0] : 0.f;
        pre1[i] = (cv && ok1) ? src[1] : 0.f;
      }
    };
    prefetch(0);
#pragma unroll
    for (int t = 0; t < kMaxStages; ++t) {
      if (t < NS) {
        __syncthreads();   // everyone finished reading the previous stage
#pragma unroll
        for (int i = 0; i < 8; ++i)
          *reinterpret_cast<float2 *>(&lds[(rbase + 8 * i) * kLStride + 2 * pp]) = make_float2(pre0[i], pre1[i]);
        __syncthreads();
        if (t + 1 < NS) prefetch(t + 1);   // global loads fly under this stage's MFMAs
        const bool mine = qbeg + (wv * 16) < Q;   // at least the first q-tile of this wave is real
        if (mine) {
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            float bv[8];
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) bv[pt] = lds[(u * 4 + g) * kLStride + pt * 16 + j];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
              for (int pt = 0; pt < 8; ++pt)
                acc[qt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(efrag[qt][t * 16 + u], bv[pt], acc[qt][pt], 0, 0, 0);
          }
        }
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
            const float hs = v + __shfl_xor(v, 1);       // (a + b) resp. (c + d)
            const float s4 = hs + __shfl_xor(hs, 8);     // (a + b) + (c + d)   [x 0.25 > 0 dropped]
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
  const int CQ = ((C + 3) / 4 + 15) / 16 * 16;
  const int NS = CQ / 16;
  const int sfac = mode == 1 ? H / h : 1;
  const long long npx = mode == 1 ? (long long)h * w : (long long)H * W;
  const int per = mode == 1 ? 32 : kNPix;
  const int ntiles = (int)((npx + per - 1) / per);
  int tpb = (int)(((long long)ntiles * B + 2047) / 2048);
  tpb = tpb < 1 ? 1 : (tpb > 16 ? 16 : tpb);
  const dim3 grid((ntiles + tpb - 1) / tpb, B), block(512);
  for (int qbeg = 0; qbeg < Q; qbeg += 128) {   // 8 waves x 16 queries per launch
    if (mode == 0)
      hipLaunchKernelGGL((mask_gemm_kernel<0>), grid, block, 0, st, embed, feat, Q, qbeg, C, CQ, NS, H, W, h, w, sfac,
                         ntiles, tpb, out_logits, out_mask, allowed);
    else
      hipLaunchKernelGGL((mask_gemm_kernel<1>), grid, block, 0, st, embed, feat, Q, qbeg, C, CQ, NS, H, W, h, w, sfac,
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
  hipError_t e = hipMemsetAsync(allowed_count, 0, (size_t)B * Q * sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) {
    dvis_set_error("attn_mask: hipMemsetAsync: %s", hipGetErrorString(e));
    return DVIS_E_LAUNCH;
  }
  return launch(1, embed, feat, B, Q, C, H, W, h, w, nullptr, mask, allowed_count, (hipStream_t)stream);
}
