// Shared device code of the split-f16 matrix-core kernels (csrc/gemm_x3.hip, csrc/conv1x1_x3.hip): operand split, fragment
// MFMA loop over one item of the packed weight stream, the LDS ring that streams it.  See gemm_x3.hip for the design.
#pragma once
#pragma clang diagnostic ignored "-Winline-asm"      // (m0 on a clobber list: the LDS-DMA statement sets it)
#include "dvis_common.h"
#include <stdlib.h>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define DVIS_LDS __attribute__((address_space(3)))
#define DVIS_GLB __attribute__((address_space(1)))

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kTileTok = kWaves * 32;
constexpr int kStages = 3;
constexpr int kPiece = 1024;          // one operand fragment of a 32-row block: 64 lanes x 8 halves
constexpr int kScratch = 4096;        // per wave: 32 tokens x 32 floats, the epilogue's transposition buffer

__device__ __forceinline__ void glds16(const void *g, void *l) {
  __builtin_amdgcn_global_load_lds((const DVIS_GLB void *)g, (DVIS_LDS void *)l, 16, 0, 0);
}

// The same request in the MUBUF encoding, as an assembler statement: 1 KB of the stream (descriptor `rs`, scalar byte offset `so`,
// the lane's 16 bytes in `voff`) to LDS address `dst` + 16 lane.  hipcc's wait-count insertion treats the FLAT-encoded
// global_load_lds as "may touch LDS AND memory" and from then on turns every vmcnt / lgkmcnt wait it inserts into a full drain
// (s_waitcnt vmcnt(0) / lgkmcnt(0)) until both counters have been zero once; the assembler statement is not counted at all, so
// the compiler's waits for ITS loads stay counted (they only over-wait by the requests in flight, which complete in order),
// and the requests themselves are waited for explicitly (Ring::wait).
typedef unsigned x3_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ x3_u4 x3_stream_rsrc(const void *base) {
  const uintptr_t b = (uintptr_t)base;
  const x3_u4 r = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)b), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32)) & 0xffffu,
                   0xffffffffu, 0x00020000u};
  return r;
}
__device__ __forceinline__ void dma16(x3_u4 rs, unsigned voff, unsigned so, unsigned dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(rs), "s"(so) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_address(const void *p) { return (unsigned)(uintptr_t)(DVIS_LDS const char *)p; }

// v * s -> (hi, lo) for 8 values (round to nearest twice; s is a power of two, so v * s and the residual are exact)
__device__ __forceinline__ void split8(f4 a, f4 b, float s, h8 &hi, h8 &lo) {
  const float v[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f2 x = {v[2 * p], v[2 * p + 1]};
    const h2 h = __builtin_convertvector(x, h2);
    const f2 r = x - __builtin_convertvector(h, f2);
    const h2 l = __builtin_convertvector(r, h2);
    hi[2 * p] = h.x, hi[2 * p + 1] = h.y, lo[2 * p] = l.x, lo[2 * p + 1] = l.y;
  }
}

// Four 16-byte stores of operand-image fragments, then a pause before anything may touch their data registers.
// gfx950 / ROCm 7.2 hipcc: a buffer_store_dwordx4 reads its 4 data registers over ~16 cycles AFTER it has issued (a quad of
// lanes per row and cycle, dword by dword); a VALU instruction that overwrites one of them in the next few cycles — hipcc
// recycles the split's registers at once, e.g.  buffer_store_dwordx4 v[32:35], v67, s[56:59], s4 offen ; v_mul_f32 v32, s61, v40
// — reaches memory instead of the stored value in lanes 12 - 15 of every 16 (the last quads read), run-to-run different with two
// waves per SIMD.  hipcc's hazard recogniser allows that re-use after ONE wait state, and after none when the store carries a
// scalar offset register — measured here to be too little (tools/exp/bneck_debug3.py: stored low terms read 2.25, 7.5, 8.75 = the
// upper halves of the fp32 values 4.0, 65536.0, 229376.0 written by the NEXT instruction).  So: nothing is scheduled into the
// group, and 32 wait states follow it.  (8-byte stores showed the same, 4-byte stores — every other epilogue — never did.)
__device__ __forceinline__ void store_fragments4(__amdgpu_buffer_rsrc_t r, unsigned voff, h8 a, unsigned sa, h8 b, unsigned sb, h8 c,
                                                 unsigned sc, h8 d, unsigned sd) {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvis_v4u, a), r, voff, sa, 0);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvis_v4u, b), r, voff, sb, 0);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvis_v4u, c), r, voff, sc, 0);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvis_v4u, d), r, voff, sd, 0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void store_fragments2(__amdgpu_buffer_rsrc_t r, unsigned voff, h8 a, unsigned sa, h8 b, unsigned sb) {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvis_v4u, a), r, voff, sa, 0);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvis_v4u, b), r, voff, sb, 0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// One item of the weight stream: STEPS k-steps x NBL blocks of 32 output features, image [s][nb][hi, lo][lane][8 halves].
//   * The fragment pair of block t + 1 is requested before block t's three products are issued (hipcc on its own reads each
//     fragment into one register quad and waits for it right away: an LDS round trip per product pair with the matrix pipe idle
//     — one wave per SIMD has nobody else to fill it).
//   * `dma(i)`, i < NDMA: the caller's LDS-DMA requests for a LATER item, spread over the products instead of issued in one
//     burst after the barrier: a burst of 8 x 1 KB from each of the 4 waves queues 512 cycles of vector-memory issue in front
//     of the first product of every wave; between products the same requests cost their issue slot.
//   * `valu(t)`, t < STEPS * NBL: a few (<= 5) VALU instructions of the caller's, issued behind the FIRST product of block t:
//     the matrix pipe runs the block's three products (96 cycles) while the wave issues them — work that would otherwise
//     sit between two items with the pipe idle (the FFN's hidden-activation split).
template <int STEPS, int NBL, int NDMA, typename Dma, typename Valu>
__device__ __forceinline__ void mma_item(const char *stage, int lane, f16v *acc, const h8 *xh, const h8 *xl, Dma dma, Valu valu) {
  constexpr int T = STEPS * NBL;
  static_assert(NDMA <= T, "one request per block at most");
  const char *p = stage + lane * 16;
  h8 wh[2], wl[2];
  wh[0] = *(const h8 *)p, wl[0] = *(const h8 *)(p + kPiece);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int s = t / NBL, nb = t % NBL, c = t & 1;
    if (t + 1 < T) wh[c ^ 1] = *(const h8 *)(p + (t + 1) * 2 * kPiece), wl[c ^ 1] = *(const h8 *)(p + (t + 1) * 2 * kPiece + kPiece);
    if (NDMA > 0 && t * NDMA / T != (t + 1) * NDMA / T) dma(t * NDMA / T);
    __builtin_amdgcn_sched_barrier(0);      // the requests above stay above the products below
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[c], xh[s], acc[nb], 0, 0, 0);
    valu(t);
    __builtin_amdgcn_sched_barrier(0);
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[c], xl[s], acc[nb], 0, 0, 0);
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[c], xh[s], acc[nb], 0, 0, 0);
  }
}
template <int STEPS, int NBL, int NDMA, typename Dma>
__device__ __forceinline__ void mma_item(const char *stage, int lane, f16v *acc, const h8 *xh, const h8 *xl, Dma dma) {
  mma_item<STEPS, NBL, NDMA>(stage, lane, acc, xh, xl, dma, [](int) {});
}
template <int STEPS, int NBL>
__device__ __forceinline__ void mma_item(const char *stage, int lane, f16v *acc, const h8 *xh, const h8 *xl) {
  mma_item<STEPS, NBL, 0>(stage, lane, acc, xh, xl, [](int) {}, [](int) {});
}

}  // namespace
// Persistent grids: one workgroup per CU minus the reserve of dvis_x3_set_reserve (defined in csrc/gemm_x3.hip).
int dvis_x3_persistent_cus();
// Range guard (defined in csrc/gemm_x3.hip).  An activation beyond the f16 range after its 2^xexp scaling (|x| >= 65520 / 2^xexp)
// becomes (hi, lo) = (inf, -inf) in split8 and every product with it inf or NaN: the output element is NON-FINITE before its
// bias / ReLU (a ReLU would turn a NaN into 0 — the silent case).  The kernels' epilogues test exactly that at one fused
// multiply-add per output value (LayerNorm forms: at the row sum, for free) and store the launch's tag into the device word
// registered with dvis_x3_set_range_flag; the host reads it once per clip.  x3_guard(): the (flag, tag) of the next launch.
struct X3Guard {
  int *flag;      // NULL: no guard registered for the current device
  int tag;
};
X3Guard dvis_x3_guard();
namespace {

// EXTRA: ordinary loads the kernel keeps in flight, issued between an item's pieces and the next-but-one item's (the
// activation prefetch of csrc/conv1x1_x3.hip): they are newer than the item waited for, so the counted wait leaves them out too.
template <int PW, int EXTRA = 0, int NW = kWaves, int STAGES = kStages>      // PW: 1 KB pieces per wave and item (item bytes = NW * PW * 1024)
struct Ring {
  const char *src;
  char *lds;
  int period, total, it, st_cmp, st_iss, wave, lane;
  static constexpr int kItemBytes = NW * PW * kPiece;
  static constexpr int kAhead = STAGES - 1;                 // items requested ahead of the one being multiplied
  static constexpr int kLeave = (STAGES - 2) * PW + EXTRA;  // loads newer than the item waited for that may stay in flight

#ifdef DVIS_X3_DMA_MUBUF
  x3_u4 rs;
  unsigned lds0, voff;
#endif
  __device__ __forceinline__ void issue_at(size_t byte_offset) {
#ifdef DVIS_X3_DMA_MUBUF
#pragma unroll
    for (int p = 0; p < PW; ++p) dma16(rs, voff, (unsigned)byte_offset + (NW * p) * kPiece, lds0 + st_iss * kItemBytes + (NW * p) * kPiece);
#else
    const char *g = src + byte_offset + lane * 16;
    char *l = lds + st_iss * kItemBytes;
#pragma unroll
    for (int p = 0; p < PW; ++p) glds16(g + (wave + NW * p) * kPiece, l + (wave + NW * p) * kPiece);
#endif
    st_iss = st_iss + 1 == STAGES ? 0 : st_iss + 1;
  }
  __device__ __forceinline__ void init(const void *stream, char *ring, int period_, int total_, int wave_, int lane_) {
    src = (const char *)stream, lds = ring, period = period_, total = total_, it = 0, st_cmp = 0, st_iss = 0;
    wave = wave_, lane = lane_;
#ifdef DVIS_X3_DMA_MUBUF
    wave = __builtin_amdgcn_readfirstlane(wave_);
    rs = x3_stream_rsrc(stream), lds0 = lds_address(ring) + wave * kPiece, voff = lane * 16 + wave * kPiece;
#endif
  }
  __device__ __forceinline__ void start(const void *stream, char *ring, int period_, int total_, int wave_, int lane_) {
    init(stream, ring, period_, total_, wave_, lane_);
#pragma unroll
    for (int i = 0; i < kAhead; ++i)
      if (total > i) issue_at((size_t)(i % period) * kItemBytes);
  }
  // Make item `it` readable by every wave: wait for this wave's pieces of it (the kLeave newer loads stay in flight), then the
  // barrier.  drain: other vector-memory work (activation loads, the previous tile's stores) may be outstanding, or fewer
  // items than assumed are in flight — wait for everything.
  //   stage = ring.wait(drain); ring.begin(offset of the item kAhead after the one just waited for);
  //   mma_item<..., PW>(stage, ..., [&](int i) { ring.piece(i); });        // its requests, spread over the products
  __device__ __forceinline__ const char *wait(bool drain) {
    if (drain || it + kAhead > total)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLeave) : "memory");
    __builtin_amdgcn_s_barrier();
    const char *stage = lds + st_cmp * kItemBytes;
    st_cmp = st_cmp + 1 == STAGES ? 0 : st_cmp + 1;
    ++it;
    return stage;
  }
  const char *dma_src;
  char *dma_dst;
  unsigned dma_so, dma_ld;
  bool dma_on;
  __device__ __forceinline__ bool more() const { return it + kAhead - 1 < total; }      // after wait(): is there an item to request?
  __device__ __forceinline__ void begin(size_t byte_offset) {       // after wait(): `it` already counts the waited item
    dma_on = more();
#ifdef DVIS_X3_DMA_MUBUF
    dma_so = (unsigned)byte_offset, dma_ld = lds0 + st_iss * kItemBytes;
#else
    dma_src = src + byte_offset + lane * 16 + wave * kPiece;
    dma_dst = lds + st_iss * kItemBytes + wave * kPiece;
#endif
    if (dma_on) st_iss = st_iss + 1 == STAGES ? 0 : st_iss + 1;
  }
  __device__ __forceinline__ void begin_periodic() { begin((size_t)((it + kAhead - 1) % period) * kItemBytes); }
  __device__ __forceinline__ void piece(int i) {
#ifdef DVIS_X3_DMA_MUBUF
    if (dma_on) dma16(rs, voff, dma_so + i * (NW * kPiece), dma_ld + i * (NW * kPiece));
#else
    if (dma_on) glds16(dma_src + i * (NW * kPiece), dma_dst + i * (NW * kPiece));
#endif
  }
};


// Packing.  order 0: natural k (k-step S, lane half g, element e -> k = 16 S + 8 g + e); order 1: accumulator order
// (k = 32 (S >> 1) + 16 (S & 1) + 8 (e >> 2) + 4 g + (e & 3)): the order in which a lane holds the previous GEMM's output.
__device__ __forceinline__ int x3_k(int order, int S, int g, int e) {
  return order == 0 ? 16 * S + 8 * g + e : 32 * (S >> 1) + 16 * (S & 1) + 8 * (e >> 2) + 4 * g + (e & 3);
}

// rows [n0, n0 + 32 nbl) x k-steps [S0, S0 + steps) of W (N x K) -> one item image [s][nb][hi, lo][lane][8]
// taps > 1: a convolution weight (N, K / taps, taps) read as N x K with k = tap * (K / taps) + channel
__device__ __forceinline__ void x3_pack_fragment(const float *w, int64_t ldw, int N, int K, int n, int order, int S, int g,
                                                 float scale, _Float16 *hi, _Float16 *lo, int taps = 1) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = x3_k(order, S, g, e);
    const int ci = K / taps;
    const int src = taps == 1 ? k : (k % ci) * taps + k / ci;
    const float v = (n < N && k < K) ? w[(int64_t)n * ldw + src] * scale : 0.f;
    const _Float16 h = (_Float16)v;
    hi[e] = h;
    lo[e] = (_Float16)(v - (float)h);
  }
}

__global__ void x3_pack_kernel(const float *w, int64_t ldw, int N, int K, int NB, int order, float scale, _Float16 *out,
                               int64_t fragments, int taps = 1) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (pass, k-step, nb, lane)
  if (idx >= fragments) return;
  const int lane = idx & 63;
  int64_t t = idx >> 6;
  const int nb = t % NB;
  t /= NB;
  const int KS = K / 16;
  const int S = t % KS, pass = t / KS;
  _Float16 *o = out + (idx >> 6) * 1024 + lane * 8;
  x3_pack_fragment(w, ldw, N, K, 32 * (pass * NB + nb) + (lane & 31), order, S, lane >> 5, scale, o, o + 512, taps);
}

}  // namespace
