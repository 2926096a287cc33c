// gemm_x3_tile.hip — the LARGE tall GEMMs of the ViT blocks (DINOv2 / ViT-Adapter backbones of BASELINE config #5,
// mask2former/modeling/backbones_vitAdapter: qkv / proj / fc1 / fc2 at 110 430 tokens x 1024 .. 4096 features per 30-frame clip)
// on split-f16 products (x3_common.h: every fp32 operand as two f16 terms, three matrix-core products per pair, fp32 accumulate).
//
// csrc/gemm_x3.hip was shaped for the deformable encoder (K = 256: a wave keeps its 32 tokens' fragments in registers and streams
// the weights; every pass over 256 output features re-reads and re-splits the tile's rows, every wave reads every weight fragment
// from LDS).  At K = 1024 .. 4096 that costs 12 - 16 passes over a 1 - 4 MB row tile that no cache holds (7 GB of HBM reads for
// the qkv projection of one block).  Here both operands are TILED:
//   workgroup = 128 tokens x 256 features, 4 waves (one per SIMD, 204 registers) as 2 (tokens) x 2 (features), a wave owns
//   64 x 128 = 8 accumulator blocks of 32 x 32; K advances in steps of 16 through a double-buffered LDS stage of 24 KB (A: the
//   tokens' rows, fetched two steps ahead into registers and split into hi / lo f16 ONCE per tile by the threads that stage them;
//   B: the packed weights, global -> LDS directly, no registers); per step a wave reads 4 + 8 fragments of 1 KB for 24 products
//   — half the LDS bytes per product of the streaming kernel — and a row tile is read once per 256 output features from the L2
//   of the XCD that owns it (workgroups that share a row tile are neighbours there).  48.5 KB of LDS per workgroup: TWO
//   workgroups share a CU, drift apart, and one's epilogue (a 128 KB tile store) runs under the other's products.
// Measured (profiles/r05_x3_tile.txt): 9.95 ms per ViT-L block of 30 frames (qkv + proj + fc1 + fc2) against 11.09 on the
// streaming kernel = 0.35 of the nominal f16 matrix peak — which is 0.55 of what the matrix pipes SUSTAIN on this part with
// random operands (tools/exp/ubench/mfma_peak.hip: 2.34 PFLOP/s with constant operands, 1.6 - 1.67 with random ones: the clock
// gives way under the toggling; GRBM_GUI_ACTIVE / time = 1.84 GHz inside these kernels).  An 8-wave 256 x 256 variant (one
// workgroup per CU, 132 KB) measured the same 9.95 ms and was dropped.
// LDS image of one operand part (hi or lo): 16-byte k-chunks (chunk c = k / 8), rows contiguous inside a chunk (a fragment read =
// 32 lanes x 16 contiguous bytes per lane half: conflict-free for ds_read_b128's lane groups); A's two chunks 2048 + 64 bytes apart
// (the staging writes of two lanes holding one row's chunks land 16 banks apart), B's 4096 + 32.
#include "dvis_common.h"
#include "x3_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kTN = 256;

struct TileArgs {
  const float *x;
  int64_t ldx, M;
  int K, N;
  const char *wp;                    // packed weights
  float xscale, inv;
  const float *bias, *radd;
  int64_t ldres;
  float *out;
  int64_t ldo;
  int act;                           // 0 none, 1 ReLU, 2 GELU (exact)
  int *flag;
  int tag;
  int tm, tn;                        // tiles along M / N
  // PACK form (the ViT blocks' qkv projection): instead of `out`, the two-term f16 images the split-f16 attention kernel reads
  // (csrc/attention.hip: [matrix q, k, v][batch-head][token][hi 64 | lo 64]); rows are (batch entry, token) with pack_L tokens per entry
  _Float16 *pack_ws;
  int pack_heads, pack_L;
  float pack_qscale;                 // Q's factor (softmax scale x log2(e) x 2^4); K and V carry 2^4
};

// Packed weights: [n-tile][k-tile of 16][hi, lo][chunk 0, 1][256 rows][8 halves] = 16 KB per (n-tile, k-tile).
constexpr int k2TM = 128, k2TK = 16;
constexpr int k2ChunkA = 2048 + 64;                 // 128 rows x 16 B (+ 64: the two chunks of a row land 16 banks apart when written)
constexpr int k2ChunkB = 4096 + 32;
constexpr int k2PartA = 2 * k2ChunkA, k2PartB = 2 * k2ChunkB;
constexpr int k2Stage = 2 * k2PartA + 2 * k2PartB;  // A hi | A lo | B hi | B lo = 24 896 bytes
constexpr int k2Lds = 2 * k2Stage;

__global__ void x3_tile_pack_kernel(const float *__restrict__ w, int64_t ldw, int N, int K, float scale, _Float16 *__restrict__ out,
                                     int64_t pieces) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per 16-byte piece (8 k of one row), hi and lo
  if (idx >= pieces) return;
  const int row = idx & 255, c = (idx >> 8) & 1;
  const int64_t t = idx >> 9;
  const int KT = K / k2TK;
  const int kt = (int)(t % KT), nt = (int)(t / KT);
  const int n = nt * kTN + row, k0 = kt * k2TK + 8 * c;
  _Float16 *o = out + t * (2 * 2 * 256 * 8) + (size_t)c * 256 * 8 + row * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = n < N ? w[(int64_t)n * ldw + k0 + e] * scale : 0.f;
    const _Float16 h = (_Float16)v;
    o[e] = h;
    o[2 * 256 * 8 + e] = (_Float16)(v - (float)h);
  }
}

template <bool GELU, bool PACK = false>
__global__ __launch_bounds__(256, 2) void x3_tile_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / a.tn) * 8 + xcd, nt = slot - (slot / a.tn) * a.tn;
  if (mt >= a.tm) return;
  const int64_t m0 = (int64_t)mt * k2TM;
  const int KT = a.K / k2TK;

  // staging: A row tid >> 1, k-chunk tid & 1 (8 floats), two k-tiles ahead in registers; B by LDS-DMA (16 pieces of 1 KB, 4 per wave)
  const int aq = tid & 1;
  const float *arow;
  {
    const int64_t m = m0 + (tid >> 1);
    arow = a.x + (m < a.M ? m : a.M - 1) * a.ldx + 8 * aq;
  }
  // (the B pieces as MUBUF LDS-DMA statements, x3_common.h: dma16 — hipcc's own waits for the row loads stay counted)
  const x3_u4 rsb = x3_stream_rsrc(a.wp);
  const unsigned bsrc = (unsigned)nt * (unsigned)KT * 16384u, lane16 = lane * 16;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds_b = lds_address(lds);
  f4 ra[2][2];
  auto fetch_a = [&](int kt, int set) { ra[set][0] = *(const f4 *)(arow + kt * k2TK), ra[set][1] = *(const f4 *)(arow + kt * k2TK + 4); };
  auto dma_b = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wave_u + 4 * i;                 // piece: part = p >> 3, chunk = (p >> 2) & 1, quarter = p & 3
      dma16(rsb, lane16, bsrc + (unsigned)kt * 16384u + (unsigned)p * 1024u,
            lds_b + (unsigned)((kt & 1) * k2Stage + 2 * k2PartA + (p >> 3) * k2PartB + ((p >> 2) & 1) * k2ChunkB + (p & 3) * 1024));
    }
  };
  auto stash_a = [&](int st, int set) {
    h8 hi, lo;
    split8(ra[set][0], ra[set][1], a.xscale, hi, lo);
    char *p = lds + st * k2Stage + aq * k2ChunkA + (tid >> 1) * 16;
    *(h8 *)p = hi;
    *(h8 *)(p + k2PartA) = lo;
  };

  f16v acc[2][4];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  dma_b(0);
  fetch_a(0, 0);
  if (KT > 1) fetch_a(1, 1);
  stash_a(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int aoff = (wm * 64 + r) * 16 + g * k2ChunkA, boff = 2 * k2PartA + (wn * 128 + r) * 16 + g * k2ChunkB;
  // fragments: B blocks one after the other (each serves the two A blocks), the next block's pair requested before the
  // current block's six products
  h8 ah[2], al[2], bh[2], bl[2];
  auto load_a = [&](const char *s) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const char *p = s + aoff + mb * 512;
      ah[mb] = *(const h8 *)p, al[mb] = *(const h8 *)(p + k2PartA);
    }
  };
  auto load_b = [&](const char *s, int nb, int buf) {
    const char *p = s + boff + nb * 512;
    bh[buf] = *(const h8 *)p, bl[buf] = *(const h8 *)(p + k2PartB);
  };
  load_a(lds);
  load_b(lds, 0, 0);
  auto tile = [&](int kt, auto set_c) {
    constexpr int SET = decltype(set_c)::value;      // tile kt + 1 is in set SET; tile kt + 2 goes to set SET ^ 1
    if (kt + 1 < KT) dma_b(kt + 1);
    if (kt + 2 < KT) fetch_a(kt + 2, SET ^ 1);
    const char *s = lds + (kt & 1) * k2Stage;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      if (nb + 1 < 4) load_b(s, nb + 1, (nb + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0], bh[nb & 1], acc[0][nb], 0, 0, 0);
      acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1], bh[nb & 1], acc[1][nb], 0, 0, 0);
      acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bl[nb & 1], acc[0][nb], 0, 0, 0);
      acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bl[nb & 1], acc[1][nb], 0, 0, 0);
      acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[nb & 1], acc[0][nb], 0, 0, 0);
      acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[nb & 1], acc[1][nb], 0, 0, 0);
      if (nb == 1 && kt + 1 < KT) stash_a((kt + 1) & 1, SET);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 2 < KT)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // tile kt + 1's B pieces landed; tile kt + 2's two row loads stay in flight
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) {
      const char *sn = lds + ((kt + 1) & 1) * k2Stage;
      load_a(sn);
      load_b(sn, 0, 0);
    }
  };
  for (int kt = 0; kt < KT; kt += 2) {
    tile(kt, std::integral_constant<int, 1>());
    if (kt + 1 < KT) tile(kt + 1, std::integral_constant<int, 0>());
  }

  float chk = 0.f;
  const bool full = m0 + k2TM <= a.M;
  if constexpr (PACK) {
    // qkv -> the attention kernel's operand images: column n = (matrix, head, dim), row m = (batch entry, token); 32 lanes hold 32
    // consecutive dims of one head: 64 contiguous bytes per store for the hi terms, 64 for the lo terms
    const int Cq = a.pack_heads * 64;
    const size_t BH = (size_t)((a.M + a.pack_L - 1) / a.pack_L) * a.pack_heads;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = nt * kTN + wn * 128 + nb * 32 + r;
      const float bv = a.bias ? a.bias[n] : 0.f;
      const int which = n / Cq, hd = (n - which * Cq) >> 6, dim = n & 63;
      const float f = which == 0 ? a.pack_qscale : 16.f;
      _Float16 *base = a.pack_ws + ((size_t)which * BH * a.pack_L) * 128 + dim;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int64_t mrow = m0 + wm * 64 + mb * 32 + 4 * g;
        const int64_t b0 = mrow / a.pack_L;
        const int t0 = (int)(mrow - b0 * a.pack_L);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int d = 8 * (i >> 2) + (i & 3);
          const int64_t m = mrow + d;
          int tok = t0 + d;
          int64_t be = b0;
          if (tok >= a.pack_L) tok -= a.pack_L, be += 1;       // (d < 32 <= pack_L: one wrap at most)
          const float t = acc[mb][nb][i] * a.inv + bv;
          chk = __builtin_fmaf(t, 0.f, chk);
          const float sv = t * f;
          const _Float16 h = (_Float16)sv;
          if (full || m < a.M) {
            _Float16 *dst = base + ((size_t)(be * a.pack_heads + hd) * a.pack_L + tok) * 128;
            dst[0] = h;
            dst[64] = (_Float16)(sv - (float)h);
          }
        }
      }
    }
    if (a.flag != nullptr && chk != chk) atomicCAS(a.flag, 0, a.tag);
    return;
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = nt * kTN + wn * 128 + nb * 32 + r;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      // the block's 16 residual values are requested TOGETHER, before any of them is used: one memory round trip per block
      // instead of one per element (load -> add -> store chains made the residual forms half again as slow as the plain ones)
      const int64_t mrow = m0 + wm * 64 + mb * 32 + 4 * g;
      float rv[16];
      if (a.radd) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t m = mrow + 8 * (i >> 2) + (i & 3);
          rv[i] = a.radd[(full || m < a.M ? m : a.M - 1) * a.ldres + n];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t m = mrow + 8 * (i >> 2) + (i & 3);
        const float t = acc[mb][nb][i] * a.inv + bv;
        chk = __builtin_fmaf(t, 0.f, chk);
        float v = GELU ? 0.5f * t * (1.f + erff(t * 0.70710678118654752440f)) : a.act == 1 ? fmaxf(t, 0.f) : t;
        if (a.radd) v += rv[i];
        if (full || m < a.M) a.out[m * a.ldo + n] = v;
      }
    }
  }
  if (a.flag != nullptr && chk != chk) atomicCAS(a.flag, 0, a.tag);
}

}  // namespace

DVIS_EXPORT int dvis_x3_tile_supported(int N, int K) { return N > 0 && K > 0 && N % kTN == 0 && K % 32 == 0; }

DVIS_EXPORT int64_t dvis_x3_tile_packed_bytes(int N, int K) {
  if (!dvis_x3_tile_supported(N, K)) return -1;
  return (int64_t)N * K * 4;
}

DVIS_EXPORT int dvis_x3_tile_pack(const float *w, int64_t ldw, int N, int K, int wexp, void *packed, void *stream) {
  DVIS_REQUIRE(dvis_x3_tile_supported(N, K), "dvis_x3_tile_pack: N %% 256 == 0 and K %% 32 == 0 are required (N %d, K %d)", N, K);
  DVIS_REQUIRE(w && packed && ldw >= K && (uintptr_t)packed % 16 == 0, "dvis_x3_tile_pack: bad operands");
  const int64_t pieces = (int64_t)N * K / 8;
  hipLaunchKernelGGL(x3_tile_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K,
                     ldexpf(1.f, wexp), (_Float16 *)packed, pieces);
  return dvis_check_launch("dvis_x3_tile_pack");
}

// The ViT blocks' qkv projection writing the split-f16 attention kernel's operand images directly (no fp32 qkv tensor, no pack
// pass): x (B * L rows) W^T + bias with N = 3 * heads * 64 columns ordered (q | k | v, head, dim) -> ws as attn_x3_pack_kernel
// leaves it.  qscale = softmax scale x log2(e) x 2^4.
DVIS_EXPORT int dvis_x3_tile_linear_qkv(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp,
                                        const float *bias, int heads, int L, float qscale, void *ws, void *stream) {
  DVIS_REQUIRE(dvis_x3_tile_supported(N, K) && heads > 0 && N == 3 * heads * 64, "dvis_x3_tile_linear_qkv: N must be 3 * heads * 64 (N %d, heads %d), K %% 32 == 0", N, heads);
  DVIS_REQUIRE(x && wp && ws && M >= 0 && L >= 32 && M % L == 0, "dvis_x3_tile_linear_qkv: rows must be whole batch entries of L >= 32 tokens (M %lld, L %d)", (long long)M, L);
  DVIS_REQUIRE(((uintptr_t)x | (uintptr_t)wp | (uintptr_t)ws) % 16 == 0 && ldx % 4 == 0 && ldx >= K, "dvis_x3_tile_linear_qkv: 16-byte alignment, ldx %% 4 == 0");
  if (M == 0) return DVIS_OK;
  TileArgs a = {};
  a.x = x, a.ldx = ldx, a.M = M, a.K = K, a.N = N, a.wp = (const char *)wp;
  a.xscale = ldexpf(1.f, xexp), a.inv = ldexpf(1.f, -(xexp + wexp));
  a.bias = bias;
  a.pack_ws = (_Float16 *)ws, a.pack_heads = heads, a.pack_L = L, a.pack_qscale = qscale;
  const X3Guard gd = dvis_x3_guard();
  a.flag = gd.flag, a.tag = gd.tag;
  const int64_t tm = (M + k2TM - 1) / k2TM;
  a.tm = (int)tm, a.tn = N / kTN;
  const int64_t grid = (tm + 7) / 8 * 8 * a.tn;
  DVIS_REQUIRE(grid < ((int64_t)1 << 31), "dvis_x3_tile_linear_qkv: too many tiles");
  static DvisLdsOptIn op;
  if (const int rc = dvis_lds_opt_in((const void *)x3_tile_kernel<false, true>, k2Lds, &op, "x3_tile_kernel")) return rc;
  hipLaunchKernelGGL((x3_tile_kernel<false, true>), dim3((unsigned)grid), dim3(256), k2Lds, (hipStream_t)stream, a);
  return dvis_check_launch("x3_tile_kernel (qkv pack)");
}

DVIS_EXPORT int dvis_x3_tile_linear(const float *x, int64_t ldx, int64_t M, int K, const void *wp, int N, int xexp, int wexp,
                                    const float *bias, int act, const float *res, int64_t ldres, float *out, int64_t ldo, void *stream) {
  DVIS_REQUIRE(dvis_x3_tile_supported(N, K), "dvis_x3_tile_linear: N %% 256 == 0 and K %% 32 == 0 are required (N %d, K %d)", N, K);
  DVIS_REQUIRE(x && wp && out && M >= 0, "dvis_x3_tile_linear: null pointer");
  DVIS_REQUIRE(((uintptr_t)x | (uintptr_t)wp) % 16 == 0 && ldx % 4 == 0 && ldx >= K && ldo >= N,
               "dvis_x3_tile_linear: x / packed weights must be 16-byte aligned, ldx %% 4 == 0");
  DVIS_REQUIRE(act >= 0 && act <= 2, "dvis_x3_tile_linear: act must be 0 (none), 1 (ReLU) or 2 (GELU)");
  DVIS_REQUIRE(res == nullptr || ldres >= N, "dvis_x3_tile_linear: residual row stride < N");
  if (M == 0) return DVIS_OK;
  TileArgs a = {};
  a.x = x, a.ldx = ldx, a.M = M, a.K = K, a.N = N, a.wp = (const char *)wp;
  a.xscale = ldexpf(1.f, xexp), a.inv = ldexpf(1.f, -(xexp + wexp));
  a.bias = bias, a.radd = res, a.ldres = ldres, a.out = out, a.ldo = ldo, a.act = act;
  const X3Guard gd = dvis_x3_guard();
  a.flag = gd.flag, a.tag = gd.tag;
  const int64_t tm = (M + k2TM - 1) / k2TM;
  a.tm = (int)tm, a.tn = N / kTN;
  const int64_t grid = (tm + 7) / 8 * 8 * a.tn;
  DVIS_REQUIRE(grid < ((int64_t)1 << 31), "dvis_x3_tile_linear: too many tiles");
  static DvisLdsOptIn o2, o2g;
  if (act == 2) {
    if (const int rc = dvis_lds_opt_in((const void *)x3_tile_kernel<true>, k2Lds, &o2g, "x3_tile_kernel")) return rc;
    hipLaunchKernelGGL(x3_tile_kernel<true>, dim3((unsigned)grid), dim3(256), k2Lds, (hipStream_t)stream, a);
  } else {
    if (const int rc = dvis_lds_opt_in((const void *)x3_tile_kernel<false>, k2Lds, &o2, "x3_tile_kernel")) return rc;
    hipLaunchKernelGGL(x3_tile_kernel<false>, dim3((unsigned)grid), dim3(256), k2Lds, (hipStream_t)stream, a);
  }
  return dvis_check_launch("x3_tile_kernel");
}
