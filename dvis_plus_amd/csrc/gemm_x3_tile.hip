// gemm_x3_tile.hip — the LARGE tall GEMMs of the ViT blocks (DINOv2 / ViT-Adapter backbones of BASELINE config #5,
// mask2former/modeling/backbones_vitAdapter: qkv / proj / fc1 / fc2 at 110 430 tokens x 1024 .. 4096 features per 30-frame clip)
// on split-f16 products (x3_common.h: every fp32 operand as two f16 terms, three matrix-core products per pair, fp32 accumulate).
//
// csrc/gemm_x3.hip was shaped for the deformable encoder (K = 256: a wave keeps its 32 tokens' fragments in registers and streams
// the weights; every pass over 256 output features re-reads and re-splits the tile's rows, every wave reads every weight fragment
// from LDS).  At K = 1024 .. 4096 that costs 12 - 16 passes over a 1 - 4 MB row tile that no cache holds (7 GB of HBM reads for
// the qkv projection of one block) and leaves the matrix pipe 0.43 busy.  Here both operands are TILED:
//   workgroup = 256 tokens x 256 features, 8 waves as 2 (tokens) x 4 (features), a wave owns 128 x 64 = 8 accumulator blocks of
//   32 x 32; K advances in steps of 32 through a double-buffered LDS stage (A: the tokens' rows, split into hi / lo f16 ONCE per
//   tile by the threads that stage them; B: the packed weights, already split); per k-step of 16 a wave reads 8 + 4 fragments
//   of 1 KB and issues 24 products — half the LDS bytes per product of the streaming kernel, and no activation re-read: the
//   tile's rows come from memory once per 256 output features, from an L2 that holds them (workgroups that share a row tile are
//   neighbours on the same XCD).
// LDS image of one operand part (hi or lo), 256 rows x 32 k: four 16-byte k-chunks, chunk c = k / 8, rows contiguous inside a
// chunk (a fragment read = 32 lanes x 16 contiguous bytes per lane half: conflict-free for ds_read_b128's lane groups), chunks
// 4096 + 32 bytes apart (the staging writes of four lanes holding one row's four chunks land on different banks).
#include "dvis_common.h"
#include "x3_common.h"

#include <stdlib.h>

namespace {

constexpr int kTM = 256, kTN = 256, kTK = 32;
constexpr int kChunk = 4096 + 32;                  // bytes between the k-chunks of a part
constexpr int kPart = 4 * kChunk;                  // one operand part (hi or lo): 16 512 bytes
constexpr int kStage = 4 * kPart;                  // A hi | A lo | B hi | B lo: 66 048 bytes
constexpr int kTileLds = 2 * kStage;               // double buffered: 132 096 bytes

struct TileArgs {
  const float *x;
  int64_t ldx, M;
  int K, N;
  const char *wp;                    // packed weights: [n-tile][k-tile][hi, lo][chunk][256 rows][8 halves] = 32 KB per (n-tile, k-tile)
  float xscale, inv;
  const float *bias, *radd;
  int64_t ldres;
  float *out;
  int64_t ldo;
  int act;                           // 0 none, 1 ReLU, 2 GELU (exact)
  int *flag;
  int tag;
  int tm, tn;                        // tiles along M / N
  int dbg;                           // development: 1 = no global fetch in the loop, 2 = no LDS staging in the loop
};

__global__ void x3_tile_pack_kernel(const float *__restrict__ w, int64_t ldw, int N, int K, float scale, _Float16 *__restrict__ out,
                                    int64_t pieces) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one thread per 16-byte piece (8 k of one row), hi and lo
  if (idx >= pieces) return;
  const int row = idx & 255, c = (idx >> 8) & 3;
  const int64_t t = idx >> 10;
  const int KT = K / kTK;
  const int kt = (int)(t % KT), nt = (int)(t / KT);
  const int n = nt * kTN + row, k0 = kt * kTK + 8 * c;
  _Float16 *o = out + t * (2 * 4 * 256 * 8) + (size_t)c * 256 * 8 + row * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = n < N ? w[(int64_t)n * ldw + k0 + e] * scale : 0.f;
    const _Float16 h = (_Float16)v;
    o[e] = h;
    o[4 * 256 * 8 + e] = (_Float16)(v - (float)h);
  }
}

template <bool GELU>
__global__ __launch_bounds__(512) void x3_tile_kernel(TileArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, g = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  // tile of this workgroup, XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs; a row tile (and its tn feature
  // tiles) belongs to ONE XCD, whose L2 then serves the row tile's re-reads
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int mt = (slot / a.tn) * 8 + xcd, nt = slot - (slot / a.tn) * a.tn;
  if (mt >= a.tm) return;
  const int64_t m0 = (int64_t)mt * kTM;
  const int KT = a.K / kTK;

  // ---- staging: A rows (tid >> 2) and + 128, k-chunk tid & 3 (8 floats = two 16-byte loads per row); B: four 16-byte pieces
  const int aq = tid & 3;
  const float *arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m0 + (tid >> 2) + 128 * i;
    arow[i] = a.x + (m < a.M ? m : a.M - 1) * a.ldx + 8 * aq;
  }
  const char *bsrc = a.wp + (size_t)nt * KT * 32768 + tid * 16;
  f4 ra[2][2];
  h8 rb[4];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i][0] = *(const f4 *)(arow[i] + kt * kTK), ra[i][1] = *(const f4 *)(arow[i] + kt * kTK + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = *(const h8 *)(bsrc + (size_t)kt * 32768 + i * 8192);
  };
  auto stash = [&](int st) {
    char *s = lds + st * kStage;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      h8 hi, lo;
      split8(ra[i][0], ra[i][1], a.xscale, hi, lo);
      char *p = s + aq * kChunk + ((tid >> 2) + 128 * i) * 16;
      *(h8 *)p = hi;
      *(h8 *)(p + kPart) = lo;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 512 * i;               // 16-byte piece of the packed 32 KB: part = e >> 10, chunk = (e >> 8) & 3, row = e & 255
      *(h8 *)(s + 2 * kPart + (e >> 10) * kPart + ((e >> 8) & 3) * kChunk + (e & 255) * 16) = rb[i];
    }
  };

  f16v acc[4][2];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  fetch(0);
  stash(0);
  __syncthreads();
  const int aoff = (wm * 128 + r) * 16 + g * kChunk, boff = 2 * kPart + (wn * 64 + r) * 16 + g * kChunk;
  // Fragments just in time: all eight waves leave the barrier together, so a k-step that first reads its 12 fragments and then
  // issues its 24 products leaves the matrix pipes idle while the LDS serves 96 KB (768 cycles) and the LDS idle afterwards.  The
  // A fragments of block mb + 1 (and, behind the last block, the next k-step's B fragments and first A block) are requested
  // before block mb's six products are issued; a product group then waits only for what was requested one group earlier.
  h8 ah[2], al[2], bh[2][2], bl[2][2];
  auto load_b = [&](const char *s, int ks, int buf) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const char *p = s + boff + 2 * ks * kChunk + nb * 512;
      bh[buf][nb] = *(const h8 *)p, bl[buf][nb] = *(const h8 *)(p + kPart);
    }
  };
  auto load_a = [&](const char *s, int ks, int mb, int buf) {
    const char *p = s + aoff + 2 * ks * kChunk + mb * 512;
    ah[buf] = *(const h8 *)p, al[buf] = *(const h8 *)(p + kPart);
  };
  load_b(lds, 0, 0);
  load_a(lds, 0, 0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT && !(a.dbg & 1)) fetch(kt + 1);
    const char *s = lds + (kt & 1) * kStage;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int t = ks * 4 + mb;                 // A buffers alternate per block, B buffers per k-step
        if (!(a.dbg & 8)) {
          if (mb + 1 < 4)
            load_a(s, ks, mb + 1, (t + 1) & 1);
          else if (ks == 0)
            load_b(s, 1, 1), load_a(s, 1, 0, (t + 1) & 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // the three products of a block alternate between its two accumulators: a product never issues right behind the one
        // it accumulates onto
        acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t & 1], bh[ks][0], acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t & 1], bh[ks][1], acc[mb][1], 0, 0, 0);
        acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t & 1], bl[ks][0], acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t & 1], bl[ks][1], acc[mb][1], 0, 0, 0);
        acc[mb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t & 1], bh[ks][0], acc[mb][0], 0, 0, 0);
        acc[mb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t & 1], bh[ks][1], acc[mb][1], 0, 0, 0);
        if (ks == 0 && mb == 1 && kt + 1 < KT && !(a.dbg & 2)) stash((kt + 1) & 1);      // the other stage: free since the last barrier
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!(a.dbg & 4)) __syncthreads();
    if (kt + 1 < KT && !(a.dbg & 8)) {
      const char *sn = lds + ((kt + 1) & 1) * kStage;
      load_b(sn, 0, 0);
      load_a(sn, 0, 0, 0);
    }
  }

  // ---- epilogue: accumulator i of block (mb, nb) is row 8 (i / 4) + 4 g + i % 4, column r: 32 lanes store 128 contiguous bytes
  float chk = 0.f;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = nt * kTN + wn * 64 + nb * 32 + r;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int64_t m = m0 + wm * 128 + mb * 32 + 8 * (i >> 2) + 4 * g + (i & 3);
        const float t = acc[mb][nb][i] * a.inv + bv;
        chk = __builtin_fmaf(t, 0.f, chk);
        float v = GELU ? 0.5f * t * (1.f + erff(t * 0.70710678118654752440f)) : a.act == 1 ? fmaxf(t, 0.f) : t;
        if (m < a.M) {
          if (a.radd) v += a.radd[m * a.ldres + n];
          a.out[m * a.ldo + n] = v;
        }
      }
  }
  // range guard (x3_common.h): a split operand beyond the f16 range leaves a non-finite pre-activation value
  if (a.flag != nullptr && chk != chk) atomicCAS(a.flag, 0, a.tag);
}

}  // namespace

DVIS_EXPORT int dvis_x3_tile_supported(int N, int K) { return N > 0 && K > 0 && N % kTN == 0 && K % kTK == 0; }

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
  const int64_t tm = (M + kTM - 1) / kTM;
  a.tm = (int)tm, a.tn = N / kTN;
  static const int dbg = []() { const char *e = getenv("DVIS_X3_TILE_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  if (dbg) a.flag = nullptr;
  const int64_t grid = (tm + 7) / 8 * 8 * a.tn;
  DVIS_REQUIRE(grid < ((int64_t)1 << 31), "dvis_x3_tile_linear: too many tiles");
  static DvisLdsOptIn opted, opted_gelu;
  if (act == 2) {
    if (const int rc = dvis_lds_opt_in((const void *)x3_tile_kernel<true>, kTileLds, &opted_gelu, "x3_tile_kernel")) return rc;
    hipLaunchKernelGGL(x3_tile_kernel<true>, dim3((unsigned)grid), dim3(512), kTileLds, (hipStream_t)stream, a);
  } else {
    if (const int rc = dvis_lds_opt_in((const void *)x3_tile_kernel<false>, kTileLds, &opted, "x3_tile_kernel")) return rc;
    hipLaunchKernelGGL(x3_tile_kernel<false>, dim3((unsigned)grid), dim3(512), kTileLds, (hipStream_t)stream, a);
  }
  return dvis_check_launch("x3_tile_kernel");
}
