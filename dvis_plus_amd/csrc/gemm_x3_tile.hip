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

#ifndef DVIS_TILE_ABLATION
#define DVIS_TILE_ABLATION 0      // development (timing only, results garbage): 1 no DMA after the prologue, 2 no fragment reads, 4 no barrier
#endif

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
  // ROW IMAGE forms (x3_tile_img_kernel): the row operand pre-split by its producer, [row tile of 128][k-tile of 16][hi, lo][chunk]
  // [128 rows][8 halves] = the LDS image of a stage, 8 KB per (row tile, k-tile); `oimg`: the output written the same way
  const char *ximg;
  char *oimg;
  float oscale;                      // 2^xexp of the consumer of `oimg`
};

// Packed weights: [n-tile][k-tile of 16][hi, lo][chunk 0, 1][256 rows][8 halves] = 16 KB per (n-tile, k-tile).
constexpr int k2TM = 128, k2TK = 16;
constexpr int k2ChunkA = 2048 + 64;                 // 128 rows x 16 B (+ 64: the two chunks of a row land 16 banks apart when written)
constexpr int k2ChunkB = 4096 + 32;
constexpr int k2PartA = 2 * k2ChunkA, k2PartB = 2 * k2ChunkB;
constexpr int k2Stage = 2 * k2PartA + 2 * k2PartB;  // A hi | A lo | B hi | B lo = 24 896 bytes
constexpr int k2Lds = 2 * k2Stage;

// k of slot e of chunk c of k-tile kt.  order 0: natural (16 kt + 8 c + e); order 1: the order in which a lane of the image-writing
// GEMM epilogue holds its 32 x 32 blocks (16 kt + 8 (e >> 2) + 4 c + (e & 3)); order 2: the split-f16 attention kernel's epilogue
// (head kt >> 2, lane pair jp = 2 (kt & 3) + c of its 16 x 16 blocks: 64 head + 16 (e & 3) + 2 jp + (e >> 2)).
__device__ __forceinline__ int x3_tile_k(int order, int kt, int c, int e) {
  if (order == 1) return 16 * kt + 8 * (e >> 2) + 4 * c + (e & 3);
  if (order == 2) return 64 * (kt >> 2) + 16 * (e & 3) + 2 * (2 * (kt & 3) + c) + (e >> 2);
  return 16 * kt + 8 * c + e;
}

__global__ void x3_tile_pack_kernel(const float *__restrict__ w, int64_t ldw, int N, int K, float scale, _Float16 *__restrict__ out,
                                     int64_t pieces, int order) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per 16-byte piece (8 k of one row), hi and lo
  if (idx >= pieces) return;
  const int row = idx & 255, c = (idx >> 8) & 1;
  const int64_t t = idx >> 9;
  const int KT = K / k2TK;
  const int kt = (int)(t % KT), nt = (int)(t / KT);
  const int n = nt * kTN + row;
  _Float16 *o = out + t * (2 * 2 * 256 * 8) + (size_t)c * 256 * 8 + row * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = n < N ? w[(int64_t)n * ldw + x3_tile_k(order, kt, c, e)] * scale : 0.f;
    const _Float16 h = (_Float16)v;
    o[e] = h;
    o[2 * 256 * 8 + e] = (_Float16)(v - (float)h);
  }
}

// ---- epilogues (accumulator block [mb][nb]: lane (r, g) holds column n = .. + 32 nb + r, rows .. + 32 mb + 8 (i >> 2) + (i & 3) + 4 g)

// qkv -> the attention kernel's operand images: column n = (matrix, head, dim), row m = (batch entry, token); 32 lanes hold 32
// consecutive dims of one head: 64 contiguous bytes per store for the hi terms, 64 for the lo terms
__device__ __forceinline__ void tile_store_qkv(const TileArgs &a, f16v (&acc)[2][4], int nt, int wm, int wn, int r, int g, int64_t m0) {
  float chk = 0.f;
  const bool full = m0 + k2TM <= a.M;
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
        const float sv = t * f;
        const _Float16 h = (_Float16)sv;
        if (full || m < a.M) {
          chk = __builtin_fmaf(t, 0.f, chk);
          _Float16 *dst = base + ((size_t)(be * a.pack_heads + hd) * a.pack_L + tok) * 128;
          dst[0] = h;
          dst[64] = (_Float16)(sv - (float)h);
        }
      }
    }
  }
  if (a.flag != nullptr && chk != chk) atomicCAS(a.flag, 0, a.tag);
}

// fp32 rows: act(acc + bias) + residual
template <bool GELU>
__device__ __forceinline__ void tile_store_rows(const TileArgs &a, f16v (&acc)[2][4], int nt, int wm, int wn, int r, int g, int64_t m0) {
  float chk = 0.f;
  const bool full = m0 + k2TM <= a.M;
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
        float v = GELU ? 0.5f * t * (1.f + erff(t * 0.70710678118654752440f)) : a.act == 1 ? fmaxf(t, 0.f) : t;
        if (a.radd) v += rv[i];
        if (full || m < a.M) {
          chk = __builtin_fmaf(t, 0.f, chk);
          a.out[m * a.ldo + n] = v;
        }
      }
    }
  }
  if (a.flag != nullptr && chk != chk) atomicCAS(a.flag, 0, a.tag);
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

  if constexpr (PACK)
    tile_store_qkv(a, acc, nt, wm, wn, r, g, m0);
  else
    tile_store_rows<GELU>(a, acc, nt, wm, wn, r, g, m0);
}

// ---- Row-image forms.  The row operand arrives pre-split (its producer's epilogue held the values anyway): both operands are
// LDS-DMA streams, no thread touches the rows before the matrix instructions, and the stages become a ring of THREE with the
// requests two k-tiles ahead (72.9 KB per workgroup, still two per CU).  MODE 0: fp32 rows out (bias, ReLU, residual); 1: the
// qkv form; 2: GELU, then the output as the next GEMM's row image — the products are issued with the operands SWAPPED, so that a
// lane holds one token and sixteen features of a block: two 16-byte fragments of hi terms and two of lo terms (k order 1).
constexpr int k3Stages = 3, k3Lds = k3Stages * k2Stage;
constexpr int kRowTile = 8192;                       // bytes of one (row tile, k-tile) of a row image

template <int MODE>
__global__ __launch_bounds__(256, 2) void x3_tile_img_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / a.tn) * 8 + xcd, nt = slot - (slot / a.tn) * a.tn;
  if (mt >= a.tm) return;
  const int64_t m0 = (int64_t)mt * k2TM;
  const int KT = a.K / k2TK;
#if DVIS_TILE_ABLATION & 8
  const x3_u4 rsa = x3_stream_rsrc(a.ximg), rsb = x3_stream_rsrc(a.wp);        // every workgroup streams the same (cache-resident) bytes
#elif DVIS_TILE_ABLATION & 16
  const x3_u4 rsa = x3_stream_rsrc(a.ximg + (size_t)mt * KT * kRowTile), rsb = x3_stream_rsrc(a.wp);      // ... the same weights only
#else
  const x3_u4 rsa = x3_stream_rsrc(a.ximg + (size_t)mt * KT * kRowTile), rsb = x3_stream_rsrc(a.wp + (size_t)nt * KT * 16384);
#endif
  const unsigned lane16 = lane * 16, lds0 = lds_address(lds);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // a k-tile = 8 row pieces + 16 weight pieces of 1 KB: 2 + 4 per wave
  auto dma = [&](int kt, int st) {
    const unsigned sb = lds0 + (unsigned)st * k2Stage;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = wave_u + 4 * i;                 // part = p >> 2, chunk = (p >> 1) & 1, half = p & 1
      dma16(rsa, lane16, (unsigned)kt * kRowTile + (unsigned)p * 1024u, sb + (unsigned)((p >> 2) * k2PartA + ((p >> 1) & 1) * k2ChunkA + (p & 1) * 1024));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = wave_u + 4 * i;                 // part = p >> 3, chunk = (p >> 2) & 1, quarter = p & 3
      dma16(rsb, lane16, (unsigned)kt * 16384u + (unsigned)p * 1024u,
            sb + (unsigned)(2 * k2PartA + (p >> 3) * k2PartB + ((p >> 2) & 1) * k2ChunkB + (p & 3) * 1024));
    }
  };

  f16v acc[2][4];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  dma(0, 0);
  if (KT > 1) {
    dma(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const int aoff = (wm * 64 + r) * 16 + g * k2ChunkA, boff = 2 * k2PartA + (wn * 128 + r) * 16 + g * k2ChunkB;
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
  int st = 0;                                       // stage of k-tile kt
  for (int kt = 0; kt < KT; ++kt) {
#if !(DVIS_TILE_ABLATION & 1)
    if (kt + 2 < KT) dma(kt + 2, st == 0 ? 2 : st - 1);       // (the stage k-tile kt - 1 was read from: released by the last barrier)
#endif
    const char *s = lds + st * k2Stage;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#if !(DVIS_TILE_ABLATION & 2)
      if (nb + 1 < 4) load_b(s, nb + 1, (nb + 1) & 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MODE == 2) {
        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nb & 1], al[0], acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nb & 1], al[1], acc[1][nb], 0, 0, 0);
        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nb & 1], ah[0], acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nb & 1], ah[1], acc[1][nb], 0, 0, 0);
        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nb & 1], ah[0], acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nb & 1], ah[1], acc[1][nb], 0, 0, 0);
      } else {
        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0], bh[nb & 1], acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1], bh[nb & 1], acc[1][nb], 0, 0, 0);
        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bl[nb & 1], acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bl[nb & 1], acc[1][nb], 0, 0, 0);
        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[nb & 1], acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[nb & 1], acc[1][nb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 2 < KT)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // k-tile kt + 1 landed; k-tile kt + 2's six pieces stay in flight
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(DVIS_TILE_ABLATION & 4)
    __syncthreads();
#endif
    st = st == 2 ? 0 : st + 1;
#if !(DVIS_TILE_ABLATION & 2)
    if (kt + 1 < KT) {
      const char *sn = lds + st * k2Stage;
      load_a(sn);
      load_b(sn, 0, 0);
    }
#endif
  }
#if DVIS_TILE_ABLATION
  a.flag = nullptr;
#endif

  if constexpr (MODE == 1) {
    tile_store_qkv(a, acc, nt, wm, wn, r, g, m0);
  } else if constexpr (MODE == 0) {
    tile_store_rows<false>(a, acc, nt, wm, wn, r, g, m0);
  } else {
    // lane (r, g) holds token 64 wm + 32 mb + r, features n0 + 32 nb + 8 (i >> 2) + (i & 3) + 4 g: accumulators 0 .. 7 are chunk g
    // of the output image's k-tile (n0 + 32 nb) / 16, accumulators 8 .. 15 of the next one
    float chk = 0.f;
    const int n0 = nt * kTN + wn * 128;
    const int OKT = a.N / k2TK;
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc(a.oimg + ((size_t)mt * OKT + (nt * kTN + (wave_u & 1) * 128) / 16) * kRowTile, 0, 8 * kRowTile, 0x00020000);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      f4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = a.bias ? *(const f4 *)(a.bias + n0 + nb * 32 + 8 * q + 4 * g) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float t = acc[mb][nb][i] * a.inv + bq[i >> 2][i & 3];
          chk = __builtin_fmaf(t, 0.f, chk);
          v[i] = 0.5f * t * (1.f + erff(t * 0.70710678118654752440f));
        }
        h8 h0, l0, h1, l1;
        split8(f4{v[0], v[1], v[2], v[3]}, f4{v[4], v[5], v[6], v[7]}, a.oscale, h0, l0);
        split8(f4{v[8], v[9], v[10], v[11]}, f4{v[12], v[13], v[14], v[15]}, a.oscale, h1, l1);
        const unsigned voff = (unsigned)(g * 2048 + (wm * 64 + mb * 32 + r) * 16), so = (unsigned)(2 * nb) * kRowTile;
        store_fragments4(ro, voff, h0, so, l0, so + 4096, h1, so + kRowTile, l1, so + kRowTile + 4096);
      }
    }
    // (rows past M of the last tile hold whatever the input image held there: stored, never read as results; they do not count
    // for the range guard either — the input image's tail is written as zeros by every producer)
    if (a.flag != nullptr && chk != chk) atomicCAS(a.flag, 0, a.tag);
  }
}

// fp32 rows -> row image (hi / lo f16 of x 2^xexp): the stand-alone producer (tests; callers without a fused one).  One workgroup
// per (k-tile, row tile): thread = (chunk, row).  Rows past M are written as zeros.
__global__ __launch_bounds__(256) void x3_rows_image_kernel(const float *__restrict__ x, int64_t ldx, int64_t M, int KT, float scale,
                                                            char *__restrict__ img) {
  const int kt = blockIdx.x, row = threadIdx.x & 127, c = threadIdx.x >> 7;
  const int64_t mt = blockIdx.y, m = mt * k2TM + row;
  f4 u = {0.f, 0.f, 0.f, 0.f}, v = u;
  if (m < M) {
    const float *p = x + m * ldx + kt * k2TK + 8 * c;
    u = *(const f4 *)p, v = *(const f4 *)(p + 4);
  }
  h8 hi, lo;
  split8(u, v, scale, hi, lo);
  char *o = img + ((size_t)mt * KT + kt) * kRowTile + c * 2048 + row * 16;
  *(h8 *)o = hi;
  *(h8 *)(o + 4096) = lo;
}

// LayerNorm of fp32 rows -> row image: one wave per row (C <= 1024 in registers, mean then centred variance as torch), eight rows
// per workgroup.  A lane's fragments go to LDS and leave it transposed — eight neighbouring lanes store the eight rows' fragments of
// one chunk = one whole 128-byte line (lane-per-fragment stores touched 64 lines per instruction: 0.35 ms per 110 430 x 1024 rows
// against 0.16 for the fp32 output; this way 0.2).  Rows past M (up to the row tile's end) are written as zeros.
constexpr int kLnPiece = 2 * 8 * 16 + 16;            // LDS bytes per 8-k piece: [hi, lo][8 rows][16 B] + 16 (bank spread)

__global__ __launch_bounds__(512) void layernorm_rows_image_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta, int64_t M, int C, float eps,
                                                                   float scale, char *__restrict__ img) {
  __shared__ __attribute__((aligned(16))) char stage[128 * kLnPiece];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t m8 = (int64_t)blockIdx.x * 8, m = m8 + w;
  const int KT = C / k2TK;
  f4 u[2][2];
  float sum = 0.f;
  const bool on = m < M;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = 8 * (lane + 64 * i);
    if (on && k < C) {
      const float *p = x + m * C + k;
      u[i][0] = *(const f4 *)p, u[i][1] = *(const f4 *)(p + 4);
      sum += ((u[i][0].x + u[i][0].y) + (u[i][0].z + u[i][0].w)) + ((u[i][1].x + u[i][1].y) + (u[i][1].z + u[i][1].w));
    } else {
      u[i][0] = u[i][1] = f4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (8 * (lane + 64 * i) < C) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f4 d = u[i][j] - mean;
        sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = 8 * (lane + 64 * i);
    if (k < C) {
      f4 o[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f4 gm = *(const f4 *)(gamma + k + 4 * j), bt = *(const f4 *)(beta + k + 4 * j);
        o[j] = on ? (u[i][j] - mean) * rstd * gm + bt : f4{0.f, 0.f, 0.f, 0.f};
      }
      h8 hi, lo;
      split8(o[0], o[1], scale, hi, lo);
      char *dst = stage + (lane + 64 * i) * kLnPiece + w * 16;
      *(h8 *)dst = hi;
      *(h8 *)(dst + 128) = lo;
    }
  }
  __syncthreads();
  // entry e = (piece, hi / lo, row): thread t takes entries t, t + 512, ...; eight consecutive threads = one 128-byte line
  char *tile = img + (size_t)(m8 / k2TM) * KT * kRowTile + (m8 % k2TM) * 16;
  const int entries = (C / 8) * 16;
  for (int e = threadIdx.x; e < entries; e += 512) {
    const int row = e & 7, hl = (e >> 3) & 1, piece = e >> 4;
    const h8 v = *(const h8 *)(stage + piece * kLnPiece + hl * 128 + row * 16);
    *(h8 *)(tile + (size_t)(piece >> 1) * kRowTile + hl * 4096 + (piece & 1) * 2048 + row * 16) = v;
  }
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
                     ldexpf(1.f, wexp), (_Float16 *)packed, pieces, 0);
  return dvis_check_launch("dvis_x3_tile_pack");
}

// The same image with the k slots of every k-tile in the order a row image's producer writes them (x3_tile_k): 0 natural (the
// stand-alone and LayerNorm producers), 1 the GELU form of dvis_x3_tile_linear_image, 2 dvis_attention_x3_packed_image (K = heads * 64).
DVIS_EXPORT int dvis_x3_tile_pack_order(const float *w, int64_t ldw, int N, int K, int wexp, int order, void *packed, void *stream) {
  DVIS_REQUIRE(dvis_x3_tile_supported(N, K), "dvis_x3_tile_pack_order: N %% 256 == 0 and K %% 32 == 0 are required (N %d, K %d)", N, K);
  DVIS_REQUIRE(w && packed && ldw >= K && (uintptr_t)packed % 16 == 0, "dvis_x3_tile_pack_order: bad operands");
  DVIS_REQUIRE(order >= 0 && order <= 2 && (order != 2 || K % 64 == 0), "dvis_x3_tile_pack_order: order 0 .. 2 (2: K %% 64 == 0)");
  const int64_t pieces = (int64_t)N * K / 8;
  hipLaunchKernelGGL(x3_tile_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K,
                     ldexpf(1.f, wexp), (_Float16 *)packed, pieces, order);
  return dvis_check_launch("dvis_x3_tile_pack_order");
}

// ---- row images (x3_tile_img_kernel above)
DVIS_EXPORT int64_t dvis_x3_rows_image_bytes(int64_t M, int K) {
  if (M < 0 || K <= 0 || K % 32 != 0) return -1;
  return (M + k2TM - 1) / k2TM * k2TM * (int64_t)K * 4;
}

DVIS_EXPORT int dvis_x3_rows_image(const float *x, int64_t ldx, int64_t M, int K, int xexp, void *image, void *stream) {
  DVIS_REQUIRE(M >= 0 && K > 0 && K % 32 == 0, "dvis_x3_rows_image: K %% 32 == 0 is required (K %d)", K);
  if (M == 0) return DVIS_OK;
  DVIS_REQUIRE(x && image && ((uintptr_t)x | (uintptr_t)image) % 16 == 0 && ldx % 4 == 0 && ldx >= K, "dvis_x3_rows_image: 16-byte alignment, ldx %% 4 == 0");
  const int64_t tm = (M + k2TM - 1) / k2TM;
  DVIS_REQUIRE(tm < 65536, "dvis_x3_rows_image: too many rows");
  hipLaunchKernelGGL(x3_rows_image_kernel, dim3(K / k2TK, (unsigned)tm), dim3(256), 0, (hipStream_t)stream, x, ldx, M, K / k2TK, ldexpf(1.f, xexp),
                     (char *)image);
  return dvis_check_launch("x3_rows_image_kernel");
}

// LayerNorm(x) (rows of C <= 1024 contiguous floats, affine) as the row image of the GEMM that consumes it (k order 0).
DVIS_EXPORT int dvis_layernorm_rows_image(const float *x, const float *gamma, const float *beta, int64_t M, int C, float eps, int xexp,
                                          void *image, void *stream) {
  DVIS_REQUIRE(M >= 0 && C > 0 && C % 32 == 0 && C <= 1024, "dvis_layernorm_rows_image: C %% 32 == 0, C <= 1024 (C %d)", C);
  if (M == 0) return DVIS_OK;
  DVIS_REQUIRE(x && gamma && beta && image && ((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)image) % 16 == 0,
               "dvis_layernorm_rows_image: null or misaligned pointer");
  const int64_t rows = (M + k2TM - 1) / k2TM * k2TM;
  hipLaunchKernelGGL(layernorm_rows_image_kernel, dim3((unsigned)(rows / 8)), dim3(512), 0, (hipStream_t)stream, x, gamma, beta, M, C, eps,
                     ldexpf(1.f, xexp), (char *)image);
  return dvis_check_launch("layernorm_rows_image_kernel");
}

static int tile_img_args(TileArgs &a, const void *ximg, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *bias,
                         const char *who) {
  DVIS_REQUIRE(dvis_x3_tile_supported(N, K), "%s: N %% 256 == 0 and K %% 32 == 0 are required (N %d, K %d)", who, N, K);
  DVIS_REQUIRE(ximg && wp && M >= 0 && ((uintptr_t)ximg | (uintptr_t)wp) % 16 == 0, "%s: null or misaligned operand", who);
  a.ximg = (const char *)ximg, a.M = M, a.K = K, a.N = N, a.wp = (const char *)wp;
  a.xscale = ldexpf(1.f, xexp), a.inv = ldexpf(1.f, -(xexp + wexp));
  a.bias = bias;
  const X3Guard gd = dvis_x3_guard();
  a.flag = gd.flag, a.tag = gd.tag;
  const int64_t tm = (M + k2TM - 1) / k2TM;
  a.tm = (int)tm, a.tn = N / kTN;
  DVIS_REQUIRE((tm + 7) / 8 * 8 * a.tn < ((int64_t)1 << 31), "%s: too many tiles", who);
  return DVIS_OK;
}

// dvis_x3_tile_linear with the rows given as a row image.  act 0 / 1: fp32 rows out (+ residual); act 2 (GELU): the result as the
// row image `oimg` (dvis_x3_rows_image_bytes(M, N) bytes, x 2^oexp, k order 1) of the GEMM that follows — `out` is not written.
DVIS_EXPORT int dvis_x3_tile_linear_image(const void *ximg, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *bias, int act,
                                          const float *res, int64_t ldres, float *out, int64_t ldo, void *oimg, int oexp, void *stream) {
  TileArgs a = {};
  if (const int rc = tile_img_args(a, ximg, M, K, wp, N, xexp, wexp, bias, "dvis_x3_tile_linear_image")) return rc;
  DVIS_REQUIRE(act >= 0 && act <= 2, "dvis_x3_tile_linear_image: act must be 0 (none), 1 (ReLU) or 2 (GELU -> row image)");
  if (M == 0) return DVIS_OK;
  const unsigned grid = (unsigned)((a.tm + 7) / 8 * 8 * a.tn);
  static DvisLdsOptIn o0, o2;
  if (act == 2) {
    DVIS_REQUIRE(oimg && (uintptr_t)oimg % 16 == 0 && res == nullptr, "dvis_x3_tile_linear_image: the GELU form writes a row image (no residual)");
    DVIS_REQUIRE(bias == nullptr || (uintptr_t)bias % 16 == 0, "dvis_x3_tile_linear_image: bias must be 16-byte aligned");
    a.oimg = (char *)oimg, a.oscale = ldexpf(1.f, oexp);
    if (const int rc = dvis_lds_opt_in((const void *)x3_tile_img_kernel<2>, k3Lds, &o2, "x3_tile_img_kernel")) return rc;
    hipLaunchKernelGGL(x3_tile_img_kernel<2>, dim3(grid), dim3(256), k3Lds, (hipStream_t)stream, a);
  } else {
    DVIS_REQUIRE(out && ldo >= N && (res == nullptr || ldres >= N), "dvis_x3_tile_linear_image: bad output / residual");
    a.radd = res, a.ldres = ldres, a.out = out, a.ldo = ldo, a.act = act;
    if (const int rc = dvis_lds_opt_in((const void *)x3_tile_img_kernel<0>, k3Lds, &o0, "x3_tile_img_kernel")) return rc;
    hipLaunchKernelGGL(x3_tile_img_kernel<0>, dim3(grid), dim3(256), k3Lds, (hipStream_t)stream, a);
  }
  return dvis_check_launch("x3_tile_img_kernel");
}

// dvis_x3_tile_linear_qkv with the rows given as a row image.
DVIS_EXPORT int dvis_x3_tile_linear_qkv_image(const void *ximg, int64_t M, int K, const void *wp, int N, int xexp, int wexp, const float *bias,
                                              int heads, int L, float qscale, void *ws, void *stream) {
  TileArgs a = {};
  if (const int rc = tile_img_args(a, ximg, M, K, wp, N, xexp, wexp, bias, "dvis_x3_tile_linear_qkv_image")) return rc;
  DVIS_REQUIRE(heads > 0 && N == 3 * heads * 64, "dvis_x3_tile_linear_qkv_image: N must be 3 * heads * 64 (N %d, heads %d)", N, heads);
  DVIS_REQUIRE(ws && (uintptr_t)ws % 16 == 0 && L >= 32 && M % L == 0, "dvis_x3_tile_linear_qkv_image: rows must be whole batch entries of L >= 32 tokens (M %lld, L %d)",
               (long long)M, L);
  if (M == 0) return DVIS_OK;
  a.pack_ws = (_Float16 *)ws, a.pack_heads = heads, a.pack_L = L, a.pack_qscale = qscale;
  static DvisLdsOptIn o1;
  if (const int rc = dvis_lds_opt_in((const void *)x3_tile_img_kernel<1>, k3Lds, &o1, "x3_tile_img_kernel")) return rc;
  hipLaunchKernelGGL(x3_tile_img_kernel<1>, dim3((unsigned)((a.tm + 7) / 8 * 8 * a.tn)), dim3(256), k3Lds, (hipStream_t)stream, a);
  return dvis_check_launch("x3_tile_img_kernel (qkv pack)");
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
