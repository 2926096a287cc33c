// gemm_x3.hip — the tall fp32 GEMMs of the deformable encoder (579 600 tokens x 256 per 30-frame clip and layer:
// value / offset / weight / output projections and the FFN pair, mask2former/modeling/pixel_decoder/msdeformattn.py:103-131,
// ops/modules/ms_deform_attn.py:96-117) on the F16 matrix cores with fp32-grade results.
//
// Arithmetic.  gfx950 multiplies fp32 operands on the matrix cores at the VECTOR rate (157 TF, v_mfma_f32_32x32x2_f32) and
// f16 operands 16 x faster (v_mfma_f32_32x32x16_f16, fp32 accumulate).  Every fp32 operand v is carried as TWO f16 terms
//     hi = rn16(v * 2^e),  lo = rn16(v * 2^e - hi)          (22 significand bits; e: a per-operand power of two)
// and a product a*w as THREE matrix-core products  a_lo*w_hi + a_hi*w_lo + a_hi*w_hi  accumulated in fp32 (the dropped
// a_lo*w_lo term is 2^-22 of the product).  Per-term error 2^-22 (2.4e-7) with random sign: a K-term dot product is off by
// 2.4e-7 / sqrt(K) of sum|a||w| — below the 1e-7 * sum|a||w| rounding of the fp32 accumulation chain itself
// (tests/test_gemm_x3_gpu.py measures both against fp64).  3 f16 products = 3/16 of the fp32 MFMA time.
//
// Decomposition (not the reference's, which calls cuBLAS through nn.Linear): everything is computed TRANSPOSED,
//     out^T (n, token) = W (n, k) x^T (k, token):   MFMA "A" operand = weight fragment, "B" operand = activations.
// A wave owns 32 tokens (the MFMA columns, lane & 31 = token) and ALL output features of a pass, so
//   * a token's features are the 16 accumulator registers x NB blocks of ONE lane pair (l, l ^ 32): LayerNorm statistics are
//     an in-lane sum + one exchange — no cross-wave reduction, no LDS;
//   * the activations are read straight from memory in fragment shape (32 B per lane and k-step) and split in registers: no
//     activation staging in LDS, no redundant split;
//   * the FFN's hidden block comes out of phase 1 already in the layout phase 2 wants as its "B" operand — the k-order of an
//     MFMA is free as long as both operands agree, and the packed W2 fragments are stored in ACCUMULATOR order — so
//     linear1 -> ReLU -> linear2 -> + residual -> LayerNorm is one kernel and the 2.4 GB hidden tensor never exists.
// Weights are split and packed once (dvis_x3_pack*) into the exact LDS image: fragments of 1 KB (64 lanes x 16 B), streamed
// with global_load_lds_dwordx4 (no VGPR round trip) through a ring of 32 - 36 KB items shared by the waves of a workgroup
// (x3_common.h); one raw s_barrier per item, counted vmcnt so that the next item stays in flight across it.
// Three kernels, all persistent (one workgroup per CU, dvis_x3_set_reserve CUs left out):
//   x3_linear_stream_kernel  projections (+ residual + LayerNorm (+ pos)): 8 waves = two per SIMD at <= 256 registers, the
//                            activations streamed in chunks of 64 k with the next chunk in flight; any K % 64 == 0;
//                            the 288-column offsets | logits projection with the position embedding added while the fragments
//                            are built (9 blocks: the item image carries a tenth, zero, block so that its pieces divide among
//                            the 8 waves; round 6: 453 -> 383 us per 30-frame layer against the resident-fragment kernel);
//   x3_linear_kernel         the other `x + embedding` projections (N = 128 / 192 / 256 at K = 256): 4 waves, the row's
//                            fragments resident (128 registers per lane);
//   x3_ffn_kernel            linear1 + ReLU + linear2 + residual + LayerNorm: 4 waves at 512 registers (fragments of the row
//                            128, accumulators 128 + 64, hidden fragments 64).
// The ring's LDS-DMA requests in the MUBUF encoding (x3_common.h: dma16): hipcc's own waits stay counted instead of full drains
// (linear 256 -> 256 0.364 -> 0.350 ms, FFN 1.895 -> 1.802 ms per 30-frame layer; the convolution kernels measured +- 1 % and
// keep the FLAT form).
#define DVIS_X3_DMA_MUBUF 1
#include "dvis_common.h"
#include "x3_common.h"

#include <stdlib.h>

namespace {

// Activation fragments of a wave's 32 tokens for K = 16 * KS, natural k order: lane (token j, half g) holds
// x[token][16 S + 8 g + 0..7] for k-step S.
// addp: NULL or the row of an embedding added in front of the split (x + pos is never written).
template <int KS>
__device__ __forceinline__ void load_x(const float *x, int64_t ldx, int64_t row, int g, float s, h8 *xh, h8 *xl,
                                       const float *addp = nullptr) {
  const float *p = x + row * ldx + 8 * g;
  f4 raw[2 * KS];
#pragma unroll
  for (int S = 0; S < KS; ++S) raw[2 * S] = *(const f4 *)(p + 16 * S), raw[2 * S + 1] = *(const f4 *)(p + 16 * S + 4);
  if (addp != nullptr) {
#pragma unroll
    for (int S = 0; S < KS; ++S) raw[2 * S] += *(const f4 *)(addp + 16 * S), raw[2 * S + 1] += *(const f4 *)(addp + 16 * S + 4);
  }
#pragma unroll
  for (int S = 0; S < KS; ++S) split8(raw[2 * S], raw[2 * S + 1], s, xh[S], xl[S]);
}

// Row-contiguous stores of one 32-feature block: the lane-per-token accumulator quads go through the wave's 4 KB of LDS
// (rows of 128 B, 16-byte slots XOR-swizzled with (row >> 1) & 7: conflict-free both ways) and leave as 8 lanes x 16 B per
// token row — 8 lines per store instruction instead of 32.
// `radd` (NULL or the same block of a residual tensor, row stride ldr) is added on the way out, in the row-contiguous shape.
__device__ __forceinline__ void store_block(char *scr, int lane, const f4 *v, float *dst, int64_t ld, int rows_valid,
                                            const float *radd = nullptr, int64_t ldr = 0) {
  const int j = lane & 31, g = lane >> 5, sw = (j >> 1) & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) *(f4 *)(scr + j * 128 + (((2 * q + g) ^ sw) << 4)) = v[q];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int jr = 8 * t + (lane >> 3), pp = lane & 7;
    f4 o = *(const f4 *)(scr + jr * 128 + ((pp ^ ((jr >> 1) & 7)) << 4));
    if (radd != nullptr && jr < rows_valid) o += *(const f4 *)(radd + jr * ldr + 4 * pp);
    if (jr < rows_valid) *(f4 *)(dst + jr * ld + 4 * pp) = o;
  }
  __builtin_amdgcn_wave_barrier();
}

struct EpiArgs {
  const float *bias;                 // N
  const float *res;                  // M x N (LayerNorm forms), row stride ldres
  int64_t ldres;
  const float *gamma, *beta;         // N
  float eps;
  const float *pos;                  // pos_rows x N or NULL: second output out2[t] = out[t] + pos[t mod pos_rows]
  int64_t pos_rows;
  float *out, *out2;
  int64_t ldo;
  float inv;                         // 2^-(xexp + wexp)
  int relu;                          // plain form: 0 none, 1 ReLU, 2 GELU (exact: 0.5 t (1 + erf(t / sqrt 2)), nn.GELU())
  const float *radd;                 // plain form: NULL or M x N residual (row stride ldres) added AFTER the activation
  int *flag;                         // range guard (x3_common.h): NULL or the device word that receives `tag` on a non-finite row
  int tag;
};

// acc (NB blocks x 16) -> out.  LN = false: act(acc * inv + bias).  LN = true: LayerNorm(acc * inv + bias + res) over the
// 32 * NB features (two-pass mean / centred variance as torch), optional out2 = out + pos.
// cb / cg / cbeta: bias (of this pass) / gamma / beta in LDS (staged once per workgroup): at 512 registers per lane hipcc
// serialises global loads in an epilogue (one in flight, 32 round trips in a row); the residual rows and the position rows
// are requested as ONE batch each, into the registers the activation fragments no longer need.
// EX: the plain form with the GELU / residual options (e.relu == 2, e.radd) — its own instantiation: the erf polynomial next to
// 128 live accumulator values costs the ReLU kernels registers (23 spilled at NB = 8).
template <int NB, bool LN, bool EX = false>
__device__ __forceinline__ void epilogue(f16v *acc, const EpiArgs &e, const float *cb, const float *cg, const float *cbeta, char *scr,
                                         int lane, int64_t tok0, int64_t M, int col0 = 0) {
  const int j = lane & 31, g = lane >> 5;
  const int64_t tok = tok0 + j < M ? tok0 + j : M - 1;
  const int rows_valid = M - tok0 < 32 ? (int)(M - tok0) : 32;
  constexpr int N = 32 * NB;
  float mean = 0.f, rstd = 1.f;
  // residual rows one block (4 x 16 bytes per lane) at a time, the next block in flight while one is consumed: the VALU sees
  // only 256 registers, so a whole row (128 values) next to the 128 accumulator values does not fit
  if constexpr (LN) {
    const float *rp = e.res + tok * e.ldres + 4 * g;
    f4 r[2][4];
    auto fetch = [&](int nb, f4 *dst) {
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q] = *(const f4 *)(rp + 32 * nb + 8 * q);
    };
    fetch(0, r[0]);
    float sum = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (nb + 1 < NB) fetch(nb + 1, r[(nb + 1) & 1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f4 b = *(const f4 *)(cb + 32 * nb + 8 * q + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = acc[nb][4 * q + i] * e.inv + b[i] + r[nb & 1][q][i];
          acc[nb][4 * q + i] = v;
          sum += v;
        }
      }
    }
    sum += __shfl_xor(sum, 32);
    // range guard: a split operand beyond the f16 range leaves a non-finite value in the row, hence in its sum
    if (e.flag != nullptr && !(fabsf(sum) <= 3.4028235e38f)) atomicCAS(e.flag, 0, e.tag);      // first offender wins: later layers only inherit its NaNs
    mean = sum * (1.f / N);
    float sq = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = acc[nb][i] - mean;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 32);
    rstd = rsqrtf(sq * (1.f / N) + e.eps);
  }
  float chk = 0.f;                   // range guard of the plain form: NaN as soon as one pre-activation value is inf / NaN
  const float *pr = nullptr;
  f4 pv[2][4];
  auto fetch_pos = [&](int nb, f4 *dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *(const f4 *)(pr + 32 * nb + 8 * q);
  };
  if constexpr (LN) {
    if (e.pos != nullptr) {
      pr = e.pos + (tok % e.pos_rows) * N + 4 * g;
      fetch_pos(0, pv[0]);
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    f4 v[4];
    if constexpr (LN) {
      if (e.pos != nullptr && nb + 1 < NB) fetch_pos(nb + 1, pv[(nb + 1) & 1]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 32 * nb + 8 * q + 4 * g;
      if constexpr (LN) {
        const f4 ga = *(const f4 *)(cg + n), be = *(const f4 *)(cbeta + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[q][i] = (acc[nb][4 * q + i] - mean) * rstd * ga[i] + be[i];
      } else {
        const f4 b = *(const f4 *)(cb + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float t = acc[nb][4 * q + i] * e.inv + b[i];
          chk = __builtin_fmaf(t, 0.f, chk);
          if constexpr (EX)
            v[q][i] = e.relu == 1 ? fmaxf(t, 0.f) : e.relu == 2 ? 0.5f * t * (1.f + erff(t * 0.70710678118654752440f)) : t;
          else
            v[q][i] = e.relu ? fmaxf(t, 0.f) : t;
        }
      }
    }
    if constexpr (!EX)
      store_block(scr, lane, v, e.out + tok0 * e.ldo + col0 + 32 * nb, e.ldo, rows_valid);
    else
      store_block(scr, lane, v, e.out + tok0 * e.ldo + col0 + 32 * nb, e.ldo, rows_valid,
                  e.radd ? e.radd + tok0 * e.ldres + col0 + 32 * nb : nullptr, e.ldres);
    if constexpr (LN) {
      if (e.pos != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += pv[nb & 1][q];
        store_block(scr, lane, v, e.out2 + tok0 * e.ldo + 32 * nb, e.ldo, rows_valid);
      }
    }
  }
  if constexpr (!LN) {
    if (e.flag != nullptr && chk != chk) atomicCAS(e.flag, 0, e.tag);      // first offender wins: later layers only inherit its NaNs
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// out (M x 32 NB npass) = epilogue( x (M x K) W^T ): one projection, its output features in `npass` passes of 32 NB over the
// same activation fragments.  Weight stream: [pass][K / 32 items of 2 k-steps x NB blocks].
template <int K, int NB, bool LN>
__global__ __launch_bounds__(kThreads) void x3_linear_kernel(const float *__restrict__ x, int64_t ldx, int64_t M,
                                                             const void *__restrict__ wp, float xscale, int npass,
                                                             const float *__restrict__ xadd, int64_t xadd_rows, EpiArgs e) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int KS = K / 16, NI = K / 32;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  const int64_t ntiles = (M + kTileTok - 1) / kTileTok;
  const int64_t my = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  Ring<NB> ring;
  ring.start(wp, lds, NI * npass, (int)(my * NI * npass), wave, lane);
  char *scr = lds + kStages * Ring<NB>::kItemBytes + wave * kScratch;
  // bias (all passes) | gamma | beta in LDS
  float *cst = (float *)(lds + kStages * Ring<NB>::kItemBytes + kWaves * kScratch);
  const int ntot = 32 * NB * npass;
  for (int i = threadIdx.x; i < ntot; i += kThreads) cst[i] = e.bias ? e.bias[i] : 0.f;
  if constexpr (LN)
    for (int i = threadIdx.x; i < 32 * NB; i += kThreads) cst[ntot + i] = e.gamma[i], cst[ntot + 32 * NB + i] = e.beta[i];
  __syncthreads();
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t tok0 = tile * kTileTok + wave * 32;
    const int64_t row = tok0 + j < M ? tok0 + j : M - 1;
    h8 xh[KS], xl[KS];
    load_x<KS>(x, ldx, row, g, xscale, xh, xl, xadd ? xadd + (row % xadd_rows) * K + 8 * g : nullptr);
    for (int pass = 0; pass < npass; ++pass) {
      f16v acc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
#pragma unroll
      for (int c = 0; c < NI; ++c) {
        const char *stage = ring.wait(c == 0);
        ring.begin_periodic();
        mma_item<2, NB, NB>(stage, lane, acc, xh + 2 * c, xl + 2 * c, [&](int i) { ring.piece(i); });
      }
      if (tok0 < M) epilogue<NB, LN>(acc, e, cst + pass * 32 * NB, cst + ntot, cst + ntot + 32 * NB, scr, lane, tok0, M, pass * 32 * NB);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same projection with the activations STREAMED: 8 waves (two per SIMD, <= 256 registers each: one wave's barrier,
// LDS-DMA issue and epilogue are covered by its partner's products) x 32 tokens; the 64 k-values of the chunk after the one
// being multiplied are in flight (8 x 16 bytes per lane; the ring's counted wait leaves them out), so neither operand waits
// behind the other and K is any multiple of 64.  Passes of 32 NB output features re-read the tile's rows from L2.
// NB = 9 (the 288-column offsets | logits projection): the item LAYOUT has ten blocks per k-step (40 pieces = 5 per wave; the tenth
// block is zero weights, never multiplied), the products run over nine.
// ADD: x + xadd[row mod xadd_rows] is formed while the fragments are built (`with_pos_embed(src, pos)` of the encoder layers); the
// rows and the embedding's rows are then fetched 32 k at a time (4 + 4 loads in flight: the same eight as the plain form's chunk).
constexpr int x3_stream_layout_blocks(int nb) { return 4 * nb % 8 == 0 ? nb : nb + 1; }

template <int NB, bool LN, bool EX = false, bool ADD = false>
__global__ __launch_bounds__(512) void x3_linear_stream_kernel(const float *__restrict__ x, int64_t ldx, int64_t M, int K,
                                                               const void *__restrict__ wp, float xscale, int npass,
                                                               const float *__restrict__ xadd, int64_t xadd_rows, EpiArgs e) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int NBL = x3_stream_layout_blocks(NB);
  constexpr int NW = 8, PW = 4 * NBL / NW, kTile = NW * 32;
  typedef Ring<PW, 8, NW> RingT;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  const int NI = K / 32, NC = K / 64;
  const int64_t ntiles = (M + kTile - 1) / kTile;
  const int64_t my = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  if (my == 0) return;
  RingT ring;
  ring.start(wp, lds, NI * npass, (int)(my * NI * npass), wave, lane);
  char *scr = lds + kStages * RingT::kItemBytes + wave * kScratch;
  float *cst = (float *)(lds + kStages * RingT::kItemBytes + NW * kScratch);
  const int ntot = 32 * NB * npass;
  for (int i = threadIdx.x; i < ntot; i += 512) cst[i] = e.bias ? e.bias[i] : 0.f;
  if constexpr (LN)
    for (int i = threadIdx.x; i < 32 * NB; i += 512) cst[ntot + i] = e.gamma[i], cst[ntot + 32 * NB + i] = e.beta[i];
  __syncthreads();
  // one item (2 k-steps) of products
  auto products = [&](const char *stage, f16v *acc, const h8 *xh, const h8 *xl) {
    if constexpr (NBL == NB) {
      mma_item<2, NB, PW>(stage, lane, acc, xh, xl, [&](int i) { ring.piece(i); });
    } else {
      mma_item<1, NB, 3>(stage, lane, acc, xh, xl, [&](int i) { ring.piece(i); });
      mma_item<1, NB, PW - 3>(stage + NBL * 2 * kPiece, lane, acc, xh + 1, xl + 1, [&](int i) { ring.piece(3 + i); });
    }
  };
  auto row_of = [&](int64_t tile) {
    const int64_t t = tile * kTile + wave * 32 + j;
    return t < M ? t : M - 1;
  };
  if constexpr (ADD) {
    f4 raw[4], rawp[4];
    auto load_raw = [&](const float *rowp, const float *posp, int it) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        raw[2 * s] = *(const f4 *)(rowp + 32 * it + 16 * s), raw[2 * s + 1] = *(const f4 *)(rowp + 32 * it + 16 * s + 4);
        rawp[2 * s] = *(const f4 *)(posp + 32 * it + 16 * s), rawp[2 * s + 1] = *(const f4 *)(posp + 32 * it + 16 * s + 4);
      }
    };
    int64_t r0 = row_of(blockIdx.x);
    const float *rowp = x + r0 * ldx + 8 * g, *posp = xadd + (r0 % xadd_rows) * K + 8 * g;
    load_raw(rowp, posp, 0);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t tok0 = tile * kTile + wave * 32;
      const int64_t rn = row_of(tile + gridDim.x < ntiles ? tile + gridDim.x : tile);
      const float *nrowp = x + rn * ldx + 8 * g, *nposp = xadd + (rn % xadd_rows) * K + 8 * g;
      for (int pass = 0; pass < npass; ++pass) {
        f16v acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
        for (int it = 0; it < NI; ++it) {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(raw[i]), "+v"(rawp[i]));
          h8 xh[2], xl[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) split8(raw[2 * s] + rawp[2 * s], raw[2 * s + 1] + rawp[2 * s + 1], xscale, xh[s], xl[s]);
          // ALWAYS 8 loads here: the next item's rows, the next pass's first, the next tile's first
          if (it + 1 < NI)
            load_raw(rowp, posp, it + 1);
          else if (pass + 1 < npass)
            load_raw(rowp, posp, 0);
          else
            load_raw(nrowp, nposp, 0);
          const char *stage = ring.wait(false);
          ring.begin_periodic();
          products(stage, acc, xh, xl);
        }
        if (tok0 < M) epilogue<NB, LN, EX>(acc, e, cst + pass * 32 * NB, cst + ntot, cst + ntot + 32 * NB, scr, lane, tok0, M, pass * 32 * NB);
      }
      rowp = nrowp, posp = nposp;
    }
    return;
  }
  f4 raw[8];
  auto load_raw = [&](const float *rowp, int kc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) raw[2 * s] = *(const f4 *)(rowp + 64 * kc + 16 * s), raw[2 * s + 1] = *(const f4 *)(rowp + 64 * kc + 16 * s + 4);
  };
  auto row_ptr = [&](int64_t tile) { return x + row_of(tile) * ldx + 8 * g; };
  const float *rowp = row_ptr(blockIdx.x);
  load_raw(rowp, 0);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t tok0 = tile * kTile + wave * 32;
    const float *nrowp = row_ptr(tile + gridDim.x < ntiles ? tile + gridDim.x : tile);
    for (int pass = 0; pass < npass; ++pass) {
      f16v acc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
      for (int kc = 0; kc < NC; ++kc) {
        // the chunk's values are taken HERE (see csrc/conv1x1_x3.hip)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(raw[i]));
        h8 xh[4], xl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) split8(raw[2 * s], raw[2 * s + 1], xscale, xh[s], xl[s]);
        // ALWAYS 8 loads here: the next chunk, the next pass's first, the next tile's first
        if (kc + 1 < NC)
          load_raw(rowp, kc + 1);
        else
          load_raw(pass + 1 < npass ? rowp : nrowp, 0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const char *stage = ring.wait(false);
          ring.begin_periodic();
          products(stage, acc, xh + 2 * half, xl + 2 * half);
        }
      }
      if (tok0 < M) epilogue<NB, LN, EX>(acc, e, cst + pass * 32 * NB, cst + ntot, cst + ntot + 32 * NB, scr, lane, tok0, M, pass * 32 * NB);
    }
    rowp = nrowp;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// out = LayerNorm( x + linear2( relu( linear1(x) ) ) ) for K = N = 256 model features and H = 128 HB hidden units.
// Weight stream per tile: HB hidden blocks x [4 items of linear1 (ONE block of 32 hidden units x all 16 k-steps), 4 items of
// linear2 (2 hidden k-steps x 8 blocks of 32 outputs, k in accumulator order)], every item 32 KB.  A block of hidden units is
// complete after its item, so its bias + ReLU + split (80 VALU instructions) is threaded through the NEXT item's products,
// five instructions behind the first product of each of its 16 blocks (mma_item's valu) — between the items it cost 0.3 ms
// of 1.8 per 30-frame layer with the matrix pipe idle.
struct FfnArgs {
  const float *b1;                   // H
  float inv1, hscale;                // 2^-(xexp + w1exp); 2^hexp (the hidden activations' split exponent)
  int HB;
};

__global__ __launch_bounds__(kThreads) void x3_ffn_kernel(const float *__restrict__ x, int64_t ldx, int64_t M,
                                                          const void *__restrict__ wp, float xscale, FfnArgs f, EpiArgs e) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int KS = 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  const int64_t ntiles = (M + kTileTok - 1) / kTileTok;
  const int64_t my = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int per_tile = 8 * f.HB;
  Ring<8> ring;
  ring.start(wp, lds, per_tile, (int)(my * per_tile), wave, lane);
  char *scr = lds + kStages * Ring<8>::kItemBytes + wave * kScratch;
  // linear1's bias lives in LDS: a global load between two items would make the compiler drain the ring behind it
  float *b1s = (float *)(lds + kStages * Ring<8>::kItemBytes + kWaves * kScratch);
  for (int i = threadIdx.x; i < 128 * f.HB; i += kThreads) b1s[i] = f.b1[i];
  float *cst = b1s + 128 * f.HB;             // b2 | gamma | beta
  for (int i = threadIdx.x; i < 256; i += kThreads) cst[i] = e.bias[i], cst[256 + i] = e.gamma[i], cst[512 + i] = e.beta[i];
  __syncthreads();
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t tok0 = tile * kTileTok + wave * 32;
    const int64_t row = tok0 + j < M ? tok0 + j : M - 1;
    h8 xh[KS], xl[KS];
    load_x<KS>(x, ldx, row, g, xscale, xh, xl);
    f16v acc2[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[nb][i] = 0.f;
    for (int hb = 0; hb < f.HB; ++hb) {
      f16v acc1[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[t][i] = 0.f;
      h8 hh[8], hl[8];
      // half-step t of the conversion of hidden tile `tc`: pair = t >> 1 (k-step u = pair >> 2 of the tile, elements 2 pp, 2 pp + 1),
      // first half: bias + ReLU + scale, second half: the split
      f2 cv = {0.f, 0.f};
      auto convert = [&](int tc, int t) {
        const int pair = t >> 1, u = pair >> 2, pp = pair & 3;
        if ((t & 1) == 0) {
          const f2 b = *(const f2 *)(b1s + 128 * hb + 32 * tc + 16 * u + 8 * (pp >> 1) + 4 * g + 2 * (pp & 1));
          cv[0] = fmaxf(acc1[tc][8 * u + 2 * pp] * f.inv1 + b[0], 0.f) * f.hscale;
          cv[1] = fmaxf(acc1[tc][8 * u + 2 * pp + 1] * f.inv1 + b[1], 0.f) * f.hscale;
        } else {
          const h2 h = __builtin_convertvector(cv, h2);
          const f2 r = cv - __builtin_convertvector(h, f2);
          const h2 l = __builtin_convertvector(r, h2);
          hh[2 * tc + u][2 * pp] = h.x, hh[2 * tc + u][2 * pp + 1] = h.y;
          hl[2 * tc + u][2 * pp] = l.x, hl[2 * tc + u][2 * pp + 1] = l.y;
        }
      };
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const char *stage = ring.wait(hb == 0 && c == 0);
        ring.begin_periodic();
        mma_item<16, 1, 8>(stage, lane, acc1 + c, xh, xl, [&](int i) { ring.piece(i); }, [&](int t) {
          if (c > 0) convert(c - 1, t);
        });
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const char *stage = ring.wait(false);
        ring.begin_periodic();
        if (c == 0) {
          // tile 3 is converted behind item 0's products, which read tile 0's fragments only (hh[0], hh[1])
          mma_item<2, 8, 8>(stage, lane, acc2, hh, hl, [&](int i) { ring.piece(i); }, [&](int t) { convert(3, t); });
        } else {
          mma_item<2, 8, 8>(stage, lane, acc2, hh + 2 * c, hl + 2 * c, [&](int i) { ring.piece(i); });
        }
      }
    }
    if (tok0 < M) epilogue<8, true>(acc2, e, cst, cst + 256, cst + 512, scr, lane, tok0, M);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// FFN stream: [hb][linear1: 4 items][linear2: 4 items]
__global__ void x3_ffn_pack_kernel(const float *w1, int64_t ldw1, const float *w2, int64_t ldw2, int H, float s1, float s2,
                                   _Float16 *out, int64_t fragments) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (hb, item, fragment pair, lane)
  if (idx >= fragments) return;
  const int lane = idx & 63;
  int64_t t = idx >> 6;
  const int fr = t & 15;             // fragment pair within the item
  t >>= 4;
  const int item = t & 7, hb = t >> 3;
  _Float16 *o = out + (((int64_t)hb * 8 + item) * 16 + fr) * 1024 + lane * 8;
  if (item < 4) {                    // W1 rows 128 hb + 32 item + i (one tile of 32 hidden units), all 16 k-steps
    x3_pack_fragment(w1, ldw1, H, 256, 128 * hb + 32 * item + (lane & 31), 0, fr, lane >> 5, s1, o, o + 512);
  } else {                           // W2 rows 32 nb + i, hidden k-steps 8 hb + 2 (item - 4) + s, accumulator order
    const int s = fr >> 3, nb = fr & 7;
    x3_pack_fragment(w2, ldw2, 256, H, 32 * nb + (lane & 31), 1, 8 * hb + 2 * (item - 4) + s, lane >> 5, s2, o, o + 512);
  }
}

int x3_grid(int64_t ntiles) {
  const int cus = dvis_x3_persistent_cus();
  return (int)(ntiles < cus ? ntiles : cus);
}

float x3_pow2(int e) { return ldexpf(1.f, e); }

template <typename Kern, typename... Args>
int x3_launch(Kern kern, DvisLdsOptIn *opted, size_t lds_bytes, int64_t M, hipStream_t st, const char *what, Args... args) {
  const int rc = dvis_lds_opt_in((const void *)kern, lds_bytes, opted, what);
  if (rc != DVIS_OK) return rc;
  const int64_t ntiles = (M + kTileTok - 1) / kTileTok;
  hipLaunchKernelGGL(kern, dim3(x3_grid(ntiles)), dim3(kThreads), lds_bytes, st, args...);
  return dvis_check_launch(what);
}

int x3_check_common(const float *x, int64_t ldx, int64_t M, const void *wp, const float *out, int64_t ldo) {
  DVIS_REQUIRE(x && wp && out, "dvis_x3: null operand");
  DVIS_REQUIRE(M >= 0 && M < (int64_t)1 << 31, "dvis_x3: M = %lld out of range", (long long)M);
  DVIS_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)wp % 16 == 0,
               "dvis_x3: operands must be 16-byte aligned with row strides %% 4 == 0");
  return DVIS_OK;
}

}  // namespace

// output features per pass (in blocks of 32) and the number of passes; 0 = not served
static int x3_passes(int N, int *nb) {
  if (N == 128 || N == 192 || N == 256 || N == 288) return *nb = N / 32, 1;
  if (N > 0 && N % 256 == 0 && N <= 8192) return *nb = 8, N / 256;
  return *nb = 0, 0;
}

// These kernels hold a CU's whole register file for the length of a launch (one persistent workgroup per CU), so a concurrent
// stream's small kernels — the previous clip's tracker chain under DVIS_Plus_offline.stream() — could otherwise start only
// between two launches.  `reserve` CUs (rounded down to a multiple of 8: one per XCD and step) are left out of the grids.
static int g_x3_reserve = [] {
  const char *e = getenv("DVIS_X3_RESERVE");
  const int r = e ? atoi(e) : 0;
  return r < 0 ? 0 : r / 8 * 8;
}();

int dvis_x3_persistent_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v >= 8) cus = v;
  }
  cus = cus / 8 * 8;
  const int r = __atomic_load_n(&g_x3_reserve, __ATOMIC_RELAXED);
  return cus - r >= 8 ? cus - r : cus;
}

DVIS_EXPORT int dvis_x3_set_reserve(int cus) {
  const int r = cus < 0 ? 0 : cus / 8 * 8;
  return __atomic_exchange_n(&g_x3_reserve, r, __ATOMIC_RELAXED);
}

// ---- range guard state: one device word per device (registered by the host layer), one tag per thread for the next launch
static int *g_x3_flag[64] = {};
static thread_local int t_x3_tag = 1;

X3Guard dvis_x3_guard() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return X3Guard{nullptr, 0};
  return X3Guard{__atomic_load_n(&g_x3_flag[dev], __ATOMIC_RELAXED), t_x3_tag};
}

DVIS_EXPORT int dvis_x3_set_range_flag(int32_t *flag) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    dvis_set_error("dvis_x3_set_range_flag: no current device");
    return DVIS_E_ARG;
  }
  __atomic_store_n(&g_x3_flag[dev], (int *)flag, __ATOMIC_RELAXED);
  return DVIS_OK;
}

DVIS_EXPORT int dvis_x3_set_tag(int tag) {
  const int prev = t_x3_tag;
  t_x3_tag = tag > 0 ? tag : 1;
  return prev;
}

DVIS_EXPORT int dvis_x3_linear_supported(int N, int K, int ln) {
  int nb;
  if (K < 64 || K % 64 != 0 || K > 8192) return 0;
  if (ln) return N == 256;
  if (N == 288) return K == 256;            // (the offsets | logits projection of the 3-level encoder)
  return x3_passes(N, &nb) > 0;
}

DVIS_EXPORT int64_t dvis_x3_packed_bytes(int N, int K) {
  if (!dvis_x3_linear_supported(N, K, 0)) return -1;
  int NB;
  const int npass = x3_passes(N, &NB);
  return (int64_t)(K / 16) * npass * x3_stream_layout_blocks(NB) * 2 * kPiece;
}

DVIS_EXPORT int dvis_x3_pack(const float *w, int64_t ldw, int N, int K, int wexp, void *packed, void *stream) {
  DVIS_REQUIRE(w && packed, "dvis_x3_pack: null operand");
  DVIS_REQUIRE(dvis_x3_linear_supported(N, K, 0), "dvis_x3_pack: (N, K) = (%d, %d) is not served (K %% 64 == 0; N in 128 / 192 / 256 or N %% 256 == 0; N = 288 at K = 256)", N, K);
  DVIS_REQUIRE(wexp >= -60 && wexp <= 60, "dvis_x3_pack: wexp = %d", wexp);
  int NB;
  const int npass = x3_passes(N, &NB);
  const int NBL = x3_stream_layout_blocks(NB);       // (N = 288: ten blocks per k-step in the image, the tenth zero)
  const int64_t fragments = (int64_t)(K / 16) * npass * NBL * 64;
  hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((fragments + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K,
                     NBL, 0, x3_pow2(wexp), (_Float16 *)packed, fragments);
  return dvis_check_launch("dvis_x3_pack");
}

static int x3_linear_impl(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *xadd,
                          int64_t xadd_rows, const float *bias, int relu, float *out, int64_t ldo, void *stream,
                          const float *radd = nullptr, int64_t ldr = 0);

DVIS_EXPORT int dvis_x3_linear(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp,
                               const float *bias, int relu, float *out, int64_t ldo, void *stream) {
  return x3_linear_impl(x, ldx, M, K, wp, N, xexp, wexp, nullptr, 0, bias, relu, out, ldo, stream);
}

DVIS_EXPORT int dvis_x3_linear_add(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp,
                                   const float *xadd, int64_t xadd_rows, const float *bias, int relu, float *out, int64_t ldo,
                                   void *stream) {
  DVIS_REQUIRE(xadd && xadd_rows > 0 && (uintptr_t)xadd % 16 == 0, "dvis_x3_linear_add: xadd (xadd_rows x K, 16-byte aligned) is required");
  DVIS_REQUIRE(K == 256 && dvis_x3_linear_supported(N, K, 0), "dvis_x3_linear_add: served for K = 256 (N %d, K %d)", N, K);
  return x3_linear_impl(x, ldx, M, K, wp, N, xexp, wexp, xadd, xadd_rows, bias, relu, out, ldo, stream);
}

DVIS_EXPORT int dvis_x3_linear_res(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp,
                                   const float *bias, int act, const float *res, int64_t ldres, float *out, int64_t ldo, void *stream) {
  DVIS_REQUIRE(act >= 0 && act <= 2, "dvis_x3_linear_res: act must be 0 (none), 1 (ReLU) or 2 (GELU)");
  DVIS_REQUIRE(res == nullptr || ((uintptr_t)res % 16 == 0 && ldres % 4 == 0 && ldres >= N),
               "dvis_x3_linear_res: res must be 16-byte aligned with a row stride that is a multiple of 4 floats and >= N");
  return x3_linear_impl(x, ldx, M, K, wp, N, xexp, wexp, nullptr, 0, bias, act, out, ldo, stream, res, ldres);
}

static int x3_linear_impl(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *xadd,
                          int64_t xadd_rows, const float *bias, int relu, float *out, int64_t ldo, void *stream, const float *radd,
                          int64_t ldr) {
  DVIS_REQUIRE(dvis_x3_linear_supported(N, K, 0), "dvis_x3_linear: (N, K) = (%d, %d) is not served (K %% 64 == 0; N in 128 / 192 / 256 or N %% 256 == 0; N = 288 at K = 256)", N, K);
  const int rc = x3_check_common(x, ldx, M, wp, out, ldo);
  if (rc != DVIS_OK) return rc;
  if (M == 0) return DVIS_OK;
  EpiArgs e = {};
  e.bias = bias, e.out = out, e.ldo = ldo, e.inv = x3_pow2(-(xexp + wexp)), e.relu = relu;
  e.radd = radd, e.ldres = ldr;
  const X3Guard gd = dvis_x3_guard();
  e.flag = gd.flag, e.tag = gd.tag;
  const float xs = x3_pow2(xexp);
  hipStream_t st = (hipStream_t)stream;
  int NB;
  const int npass = x3_passes(N, &NB);
#define DVIS_X3_STREAM_FORM(NBV, LNV, EXV, ADDV, XADD, XROWS, WHAT, LDS_EXTRA)                                        \
  {                                                                                                                  \
    static DvisLdsOptIn opted;                                                                                       \
    typedef Ring<4 * x3_stream_layout_blocks(NBV) / 8, 8, 8> R;                                                      \
    const size_t lds_bytes = kStages * R::kItemBytes + 8 * kScratch + (LDS_EXTRA);                                   \
    const int rc2 = dvis_lds_opt_in((const void *)x3_linear_stream_kernel<NBV, LNV, EXV, ADDV>, lds_bytes, &opted, WHAT);  \
    if (rc2 != DVIS_OK) return rc2;                                                                                  \
    const int64_t ntiles = (M + 255) / 256;                                                                          \
    hipLaunchKernelGGL((x3_linear_stream_kernel<NBV, LNV, EXV, ADDV>), dim3(x3_grid(ntiles)), dim3(512), lds_bytes, st, x, ldx, M, K, wp, \
                       xs, npass, XADD, XROWS, e);                                                                   \
    return dvis_check_launch(WHAT);                                                                                  \
  }
#define DVIS_X3_STREAM_EX(NBV, LNV, EXV, WHAT, LDS_EXTRA) DVIS_X3_STREAM_FORM(NBV, LNV, EXV, false, (const float *)nullptr, (int64_t)0, WHAT, LDS_EXTRA)
#define DVIS_X3_STREAM(NBV, LNV, WHAT, LDS_EXTRA) DVIS_X3_STREAM_EX(NBV, LNV, false, WHAT, LDS_EXTRA)
#define DVIS_X3_RESIDENT(NBV)                                                                                        \
  {                                                                                                                  \
    static DvisLdsOptIn opted;                                                                                       \
    return x3_launch(x3_linear_kernel<256, NBV, false>, &opted, kStages * Ring<NBV>::kItemBytes + kWaves * kScratch + (size_t)N * 4, \
                     M, st, "dvis_x3_linear", x, ldx, M, wp, xs, npass, xadd, xadd_rows, e);                         \
  }
  if (relu == 2 || radd) {      // GELU / residual epilogue: the ViT blocks' shapes (N a multiple of 256)
    DVIS_REQUIRE(NB == 8 && !xadd, "dvis_x3_linear_res: the GELU / residual epilogue is served for N %% 256 == 0 (N = %d)", N);
    DVIS_X3_STREAM_EX(8, false, true, "dvis_x3_linear_res", (size_t)N * 4)
  }
  if (xadd) {      // the embedding is added while the row's fragments are built (K = 256)
    switch (NB) {
      case 4: DVIS_X3_RESIDENT(4)
      case 6: DVIS_X3_RESIDENT(6)
      case 8: DVIS_X3_RESIDENT(8)
      default: DVIS_X3_STREAM_FORM(9, false, false, true, xadd, xadd_rows, "dvis_x3_linear_add", (size_t)N * 4)
    }
  }
  switch (NB) {
    case 4: DVIS_X3_STREAM(4, false, "dvis_x3_linear", (size_t)N * 4)
    case 6: DVIS_X3_STREAM(6, false, "dvis_x3_linear", (size_t)N * 4)
    case 8: DVIS_X3_STREAM(8, false, "dvis_x3_linear", (size_t)N * 4)
    default: DVIS_X3_STREAM(9, false, "dvis_x3_linear", (size_t)N * 4)
  }
#undef DVIS_X3_RESIDENT
}

DVIS_EXPORT int dvis_x3_linear_ln(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp,
                                  const float *bias, const float *res, int64_t ldres, const float *gamma, const float *beta,
                                  float eps, const float *pos, int64_t pos_rows, float *out, float *out2, int64_t ldo,
                                  void *stream) {
  DVIS_REQUIRE(dvis_x3_linear_supported(N, K, 1), "dvis_x3_linear_ln: (N, K) = (%d, %d) is not served (256, 256)", N, K);
  DVIS_REQUIRE(bias && res && gamma && beta, "dvis_x3_linear_ln: bias, res, gamma, beta are required");
  DVIS_REQUIRE(ldres % 4 == 0 && (uintptr_t)res % 16 == 0, "dvis_x3_linear_ln: res must be 16-byte aligned, ldres %% 4 == 0");
  DVIS_REQUIRE(pos == nullptr || (out2 && pos_rows > 0 && (uintptr_t)pos % 16 == 0 && (uintptr_t)out2 % 16 == 0),
               "dvis_x3_linear_ln: pos needs out2, pos_rows > 0 and 16-byte alignment");
  const int rc = x3_check_common(x, ldx, M, wp, out, ldo);
  if (rc != DVIS_OK) return rc;
  if (M == 0) return DVIS_OK;
  EpiArgs e = {};
  e.bias = bias, e.res = res, e.ldres = ldres, e.gamma = gamma, e.beta = beta, e.eps = eps, e.pos = pos, e.pos_rows = pos_rows;
  e.out = out, e.out2 = out2, e.ldo = ldo, e.inv = x3_pow2(-(xexp + wexp));
  const X3Guard gd = dvis_x3_guard();
  e.flag = gd.flag, e.tag = gd.tag;
  const float xs = x3_pow2(xexp);
  const int npass = 1;
  hipStream_t st = (hipStream_t)stream;
  DVIS_X3_STREAM(8, true, "dvis_x3_linear_ln", (size_t)3 * 256 * 4)
#undef DVIS_X3_STREAM
#undef DVIS_X3_STREAM_EX
#undef DVIS_X3_STREAM_FORM
}

DVIS_EXPORT int64_t dvis_x3_ffn_packed_bytes(int K, int H, int N) {
  if (K != 256 || N != 256 || H <= 0 || H % 128 != 0 || H > 4096) return -1;
  return (int64_t)(H / 128) * 8 * Ring<8>::kItemBytes;
}

DVIS_EXPORT int dvis_x3_ffn_pack(const float *w1, int64_t ldw1, const float *w2, int64_t ldw2, int K, int H, int N, int w1exp,
                                 int w2exp, void *packed, void *stream) {
  DVIS_REQUIRE(w1 && w2 && packed, "dvis_x3_ffn_pack: null operand");
  DVIS_REQUIRE(dvis_x3_ffn_packed_bytes(K, H, N) > 0, "dvis_x3_ffn_pack: needs K = N = 256, H %% 128 == 0 (K %d, H %d, N %d)", K, H, N);
  const int64_t fragments = (int64_t)(H / 128) * 8 * 16 * 64;
  hipLaunchKernelGGL(x3_ffn_pack_kernel, dim3((unsigned)((fragments + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1, ldw1,
                     w2, ldw2, H, x3_pow2(w1exp), x3_pow2(w2exp), (_Float16 *)packed, fragments);
  return dvis_check_launch("dvis_x3_ffn_pack");
}

DVIS_EXPORT int dvis_x3_ffn_ln(const float *x, int64_t ldx, int64_t M, int K, int H, int N, const void *wp, int xexp, int w1exp,
                               int hexp, int w2exp, const float *b1, const float *b2, const float *gamma, const float *beta,
                               float eps, const float *pos, int64_t pos_rows, float *out, float *out2, int64_t ldo,
                               void *stream) {
  DVIS_REQUIRE(dvis_x3_ffn_packed_bytes(K, H, N) > 0, "dvis_x3_ffn_ln: needs K = N = 256, H %% 128 == 0 (K %d, H %d, N %d)", K, H, N);
  DVIS_REQUIRE(b1 && b2 && gamma && beta, "dvis_x3_ffn_ln: biases, gamma, beta are required");
  DVIS_REQUIRE(pos == nullptr || (out2 && pos_rows > 0 && (uintptr_t)pos % 16 == 0 && (uintptr_t)out2 % 16 == 0),
               "dvis_x3_ffn_ln: pos needs out2, pos_rows > 0 and 16-byte alignment");
  const int rc = x3_check_common(x, ldx, M, wp, out, ldo);
  if (rc != DVIS_OK) return rc;
  if (M == 0) return DVIS_OK;
  EpiArgs e = {};
  e.bias = b2, e.res = x, e.ldres = ldx, e.gamma = gamma, e.beta = beta, e.eps = eps, e.pos = pos, e.pos_rows = pos_rows;
  e.out = out, e.out2 = out2, e.ldo = ldo, e.inv = x3_pow2(-(hexp + w2exp));
  const X3Guard gd = dvis_x3_guard();
  e.flag = gd.flag, e.tag = gd.tag;
  FfnArgs f = {b1, x3_pow2(-(xexp + w1exp)), x3_pow2(hexp), H / 128};
  static DvisLdsOptIn opted;
  return x3_launch(x3_ffn_kernel, &opted, kStages * Ring<8>::kItemBytes + kWaves * kScratch + (size_t)(H + 3 * 256) * 4, M,
                   (hipStream_t)stream, "dvis_x3_ffn_ln", x, ldx, M, wp, x3_pow2(xexp), f, e);
}
