// Deterministic exact-fp32 GEMM  C = act(A W^T + bias + res)  — gfx950, v_mfma_f32_16x16x4_f32.
//
// Replaces, on the tracker / refiner stream, every library GEMM behind
//   nn.MultiheadAttention in / out projections, FFN linear1 / linear2, MLP layers     dvis_Plus/tracker.py:293-318,
//                                                                                       dvis_Plus/refiner.py:104-139
//   the cosine cost matrices of Noiser.match_embds (a batched A B^T)                   dvis_Plus/noiser.py:43-56
//   nn.Conv1d(k = 5 / 3, replicate 'same' padding) of the refiner as an im2col GEMM    dvis_Plus/refiner.py:42-54,116-119
// Why own code for "plain library GEMMs": stream() overlaps the tracker / refiner of clip i with the segmenter of clip
// i + 1 on two HIP streams, and two hipBLASLt stream-K kernels in flight on two streams can spin on each other forever
// (a stream-K workgroup waits for partial tiles of peers that only become resident when the other
// kernel's spinning workgroups leave).  A kernel here never waits for another workgroup, so whatever the library runs
// on the other stream always gets the CUs it waits for.  Second reason: run-to-run reproducibility — no atomics, no
// data-dependent split; the summation order of every output element is a function of (M, N, K, configuration) only.
//
// Work decomposition (not a classic LDS-staged tile GEMM): a workgroup owns a (16 RT) x (16 CT) tile of C and SPLITS K
// OVER ITS NW WAVES.  A wave's operand fragments are then needed by no other wave of the workgroup, so they go from
// global memory (L2 / MALL-resident weights and activations) straight into the MFMA operand layout — lane (i, g) loads
// 16 bytes = k-values 16 s + 4 g .. + 3 of row i of a 16-row tile, a 16-lane group reads 64 contiguous bytes of a row,
// the k-group index goes into the buffer load's SCALAR offset — with no LDS staging and no barrier in the main loop;
// PF k-groups of fragments are in flight per wave.  The K index inside a 16-group is permuted (MFMA c of a group sums
// k = 16 s + 4 g + c over the lane groups g), identically for both operands.  The NW partial tiles meet in LDS once, are
// summed in wave order by all threads, and leave through the epilogue (bias, residual, ReLU) as 16-byte stores.
// Rows / columns past M / N and k past K read zeros through the buffer descriptor's range check (no clamped re-reads).
#include "dvis_common.h"

namespace {

constexpr unsigned kOOB = 0x80000000u;   // + any in-tile offset stays out of range without wrapping (tiles < 2 GiB)

struct GemmArgs {
  const float *A, *W, *bias, *res;
  float *C;
  long long lda, ldw, ldres, ldc;          // row strides in floats
  long long sA, sW, sRes, sC, sBias;       // batch strides in floats (sBias: 0 = one bias for the whole batch)
  int M, N, K, act, row_blocks, col_blocks, vec_store, col_fastest;
  // head-major output: column n of row m goes to C[(n / head_d) * head_stride + m * head_d + n % head_d] — the (heads, M, d)
  // layout MSDeformAttn gathers from (a pixel's d channels of ONE head are a contiguous line, neighbouring pixels adjacent);
  // head_d == 0: plain row-major C with row stride ldc
  int head_d;
  long long head_stride;
};

// Register budget: the second __launch_bounds__ argument is the minimum number of waves per SIMD the allocator must leave
// room for.  Without it hipcc spends 176 registers (80 VGPR + 96 AGPR) on the 64 x 64 tile where 64 accumulators + two
// fragment slots need ~150: 2 instead of 3 waves per SIMD, and the prologue / epilogue of one wave is then covered by only
// one other wave's MFMAs.
template <int TILES> constexpr int kGemmMinWaves = TILES >= 32 ? 1 : (TILES >= 20 ? 2 : (TILES >= 12 ? 3 : 4));

template <int RT, int CT, int NW, int PF, bool K16>
__global__ __launch_bounds__(64 * NW, kGemmMinWaves<RT * CT>) void gemm_nt_kernel(const GemmArgs p) {
  extern __shared__ float lds[];   // [NW][16 RT][16 CT] partial tiles
  constexpr int BM = 16 * RT, BN = 16 * CT, T = 64 * NW;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  // Tile of this workgroup.  Skinny problems: row blocks fastest (workgroups that share a weight slab run together).
  // Tall problems (A does not fit the caches): column blocks fastest AND XCD-aware — dispatch puts block b on XCD b % 8,
  // so the logical tile id is remapped such that an XCD walks CONSECUTIVE logical tiles: the N / BN column blocks that
  // re-read one row block of A then follow each other on one XCD's L2 instead of being spread over the launch (A would
  // be re-fetched from HBM N / BN times) or over 8 L2s.
  int rb, cb;
  if (p.col_fastest) {
    const unsigned tiles = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned q8 = tiles >> 3, r8 = tiles & 7u;                       // bijective also when tiles % 8 != 0
    const unsigned logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    rb = (int)(logical / (unsigned)p.col_blocks), cb = (int)(logical % (unsigned)p.col_blocks);
  } else {
    rb = blockIdx.x % p.row_blocks, cb = blockIdx.x / p.row_blocks;
  }
  const int row0 = rb * BM, col0 = cb * BN;
  const long long batch = blockIdx.y;

  // ---- descriptors over exactly this tile's rows of A / W: anything outside reads 0
  const int rows_a = min(BM, p.M - row0), rows_w = min(BN, p.N - col0);
  const float *Ab = p.A + batch * p.sA + (long long)row0 * p.lda;
  const float *Wb = p.W + batch * p.sW + (long long)col0 * p.ldw;
  const __amdgpu_buffer_rsrc_t ra = dvis_make_rsrc_uniform(Ab, (unsigned)((((long long)rows_a - 1) * p.lda + p.K) * 4));
  const __amdgpu_buffer_rsrc_t rw = dvis_make_rsrc_uniform(Wb, (unsigned)((((long long)rows_w - 1) * p.ldw + p.K) * 4));
  unsigned aoff[RT], woff[CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) aoff[rt] = (unsigned)((rt * 16 + i) * p.lda * 4 + g * 16);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) woff[ct] = (unsigned)((ct * 16 + i) * p.ldw * 4 + g * 16);

  // ---- this wave's k-groups (16 k each)
  const int G = (p.K + 15) >> 4;
  const int g0 = (int)((long long)G * wv / NW), g1 = (int)((long long)G * (wv + 1) / NW);

  dvis_f4 a[PF][RT], w[PF][CT];
  auto load = [&](int slot, int grp) {   // grp is wave-uniform
    const bool in = grp < g1;            // (never both offsets out of range: 2^31 + 2^31 wraps to 0)
    const unsigned so = in ? (unsigned)grp * 64u : 0u;
    const bool kv = in && (K16 || grp * 16 + g * 4 < p.K);   // K % 4 == 0: a 16-byte piece is all in or all out
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      a[slot][rt] = __builtin_bit_cast(
          dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(ra, kv ? aoff[rt] : kOOB, so, 0));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      w[slot][ct] = __builtin_bit_cast(
          dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rw, kv ? woff[ct] : kOOB, so, 0));
  };

  dvis_f4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = dvis_f4{0.f, 0.f, 0.f, 0.f};

  auto contract = [&](int s) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][rt][c], w[s][ct][c], acc[rt][ct], 0, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) load(s, g0 + s);
  int grp = g0;
  // Steady state: rounds of PF groups whose refills are all in range — ONE basic block, no branch between a slot's
  // MFMAs and its refill, so the compiler's counted `s_waitcnt vmcnt(N)` leaves the other PF - 1 slots' loads in flight
  // (with a branch around each refill it waited vmcnt(0) at the loop head: every round stalled for a full load latency).
#pragma unroll 1
  for (; grp + 2 * PF <= g1; grp += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      contract(s);
      // pin the order "slot's MFMAs, then ITS refill": left alone the scheduler sinks all refills to the end of the round,
      // and the first slot's loads then have a fraction of a slot of MFMAs to arrive
      __builtin_amdgcn_sched_barrier(0);
      load(s, grp + s + PF);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (grp + PF <= g1) {   // last full round: refill only the slots the tail below still needs (wave-uniform branches)
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      contract(s);
      if (grp + s + PF < g1) load(s, grp + s + PF);
    }
    grp += PF;
  }
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (grp + s < g1) contract(s);   // tail: fewer than PF groups left (their slots were filled above / by the prologue)

  // ---- partial tiles meet in LDS.  Accumulator layout: column = lane & 15, row = (lane >> 4) * 4 + reg.
  float *mine = lds + wv * (BM * BN);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(rt * 16 + g * 4 + r) * BN + ct * 16 + i] = acc[rt][ct][r];
  __syncthreads();

  // ---- sum in wave order, epilogue, store: one 16-byte unit per thread and round
  constexpr int U = BM * BN / 4, UR = BN / 4;
  float *Cb = p.C + batch * p.sC;
  const float *Rb = p.res ? p.res + batch * p.sRes : nullptr;
  const float *Bb = p.bias ? p.bias + batch * p.sBias : nullptr;
#pragma unroll
  for (int u0 = 0; u0 < U; u0 += T) {
    const int u = u0 + tid;
    if (U % T != 0 && u >= U) break;
    const int row = u / UR, c4 = (u - row * UR) * 4;
    dvis_f4 s = *reinterpret_cast<const dvis_f4 *>(lds + row * BN + c4);
#pragma unroll
    for (int k = 1; k < NW; ++k) {
      const dvis_f4 t = *reinterpret_cast<const dvis_f4 *>(lds + k * (BM * BN) + row * BN + c4);
      s = dvis_f4{s[0] + t[0], s[1] + t[1], s[2] + t[2], s[3] + t[3]};
    }
    const int gr = row0 + row, gc = col0 + c4;
    if (gr >= p.M || gc >= p.N) continue;
    float *dst = p.head_d ? Cb + (long long)(gc / p.head_d) * p.head_stride + (long long)gr * p.head_d + gc % p.head_d
                          : Cb + (long long)gr * p.ldc + gc;
    if (p.vec_store) {   // N % 4 == 0, 16-byte aligned rows of C / bias / res
      if (Bb) {
        const dvis_f4 b = *reinterpret_cast<const dvis_f4 *>(Bb + gc);
        s = dvis_f4{s[0] + b[0], s[1] + b[1], s[2] + b[2], s[3] + b[3]};
      }
      if (Rb) {
        const dvis_f4 t = *reinterpret_cast<const dvis_f4 *>(Rb + (long long)gr * p.ldres + gc);
        s = dvis_f4{s[0] + t[0], s[1] + t[1], s[2] + t[2], s[3] + t[3]};
      }
      if (p.act) s = dvis_f4{fmaxf(s[0], 0.f), fmaxf(s[1], 0.f), fmaxf(s[2], 0.f), fmaxf(s[3], 0.f)};
      *reinterpret_cast<dvis_f4 *>(dst) = s;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (gc + e >= p.N) break;
        float v = s[e];
        if (Bb) v += Bb[gc + e];
        if (Rb) v += Rb[(long long)gr * p.ldres + gc + e];
        if (p.act) v = fmaxf(v, 0.f);
        dst[e] = v;
      }
    }
  }
}

// Tall problems, one wave per tile (NW = 1): PERSISTENT form.  A wave walks tiles t = first, first + grid, ... and requests
// the first fragment slots of its NEXT tile before the epilogue of the current one, so a tile's prologue latency (an HBM
// round trip for the A rows) and the LDS transpose + stores of the epilogue overlap instead of adding up — at K = 256 a
// tile is only 16 k-groups long and the two cost ~15 % of it.  Tile order as above: column-fastest and XCD-aware (the grid
// is a multiple of 8: in every sweep an XCD's workgroups take consecutive logical tiles).
template <int RT, int CT, int PF, bool K16>
__global__ __launch_bounds__(64, kGemmMinWaves<RT * CT>) void gemm_nt_persist_kernel(const GemmArgs p) {
  extern __shared__ float lds[];   // [16 RT][16 CT]: the epilogue's transpose buffer
  constexpr int BM = 16 * RT, BN = 16 * CT, T = 64;
  const int tid = threadIdx.x, lane = tid;
  const int i = lane & 15, g = lane >> 4;
  const long long total = (long long)p.row_blocks * p.col_blocks;
  const unsigned gsz = gridDim.x;                                             // multiple of 8, <= total
  long long t = (long long)(blockIdx.x & 7u) * (gsz >> 3) + (blockIdx.x >> 3);
  const long long batch = blockIdx.y;
  const float *A0 = p.A + batch * p.sA, *W0 = p.W + batch * p.sW;
  unsigned aoff[RT], woff[CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) aoff[rt] = (unsigned)((rt * 16 + i) * p.lda * 4 + g * 16);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) woff[ct] = (unsigned)((ct * 16 + i) * p.ldw * 4 + g * 16);
  const int g1 = (p.K + 15) >> 4;

  int row0 = (int)(t / p.col_blocks) * BM, col0 = (int)(t % p.col_blocks) * BN;
  auto desc_a = [&](int r0) {
    const int rows = min(BM, p.M - r0);
    return dvis_make_rsrc_uniform(A0 + (long long)r0 * p.lda, (unsigned)((((long long)rows - 1) * p.lda + p.K) * 4));
  };
  auto desc_w = [&](int c0) {
    const int rows = min(BN, p.N - c0);
    return dvis_make_rsrc_uniform(W0 + (long long)c0 * p.ldw, (unsigned)((((long long)rows - 1) * p.ldw + p.K) * 4));
  };
  __amdgpu_buffer_rsrc_t ra = desc_a(row0), rw = desc_w(col0);

  dvis_f4 a[PF][RT], w[PF][CT];
  auto load = [&](int slot, int grp) {
    const bool in = grp < g1;
    const unsigned so = in ? (unsigned)grp * 64u : 0u;
    const bool kv = in && (K16 || grp * 16 + g * 4 < p.K);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      a[slot][rt] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(ra, kv ? aoff[rt] : kOOB, so, 0));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      w[slot][ct] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rw, kv ? woff[ct] : kOOB, so, 0));
  };
  dvis_f4 acc[RT][CT];
  auto contract = [&](int s) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][rt][c], w[s][ct][c], acc[rt][ct], 0, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) load(s, s);

  float *Cb = p.C + batch * p.sC;
  const float *Rb = p.res ? p.res + batch * p.sRes : nullptr;
  const __amdgpu_buffer_rsrc_t rbias =
      dvis_make_rsrc_uniform(p.bias ? p.bias + batch * p.sBias : p.A, p.bias ? (unsigned)p.N * 4u : 0u);
#pragma unroll 1
  for (;;) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = dvis_f4{0.f, 0.f, 0.f, 0.f};
    int grp = 0;
    auto round = [&]() {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        contract(s);
        __builtin_amdgcn_sched_barrier(0);
        load(s, grp + s + PF);
        __builtin_amdgcn_sched_barrier(0);
      }
      grp += PF;
    };
    // The FIRST round of a tile is peeled out of the loop: on gfx950 loads and stores share the vmcnt counter and may
    // complete out of order with respect to each other, so with the previous tile's epilogue stores possibly pending the
    // compiler must wait vmcnt(0) — once, here.  Were this round inside the loop, the loop header would inherit "stores
    // may be pending" from its entry edge and wait vmcnt(0) in EVERY round (seen in the ISA: the 80 x 64 tile lost 12 %).
    if (grp + 2 * PF <= g1) round();
#pragma unroll 1
    while (grp + 2 * PF <= g1) round();
    if (grp + PF <= g1) {
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        contract(s);
        if (grp + s + PF < g1) load(s, grp + s + PF);
      }
      grp += PF;
    }
#pragma unroll
    for (int s = 0; s < PF; ++s)
      if (grp + s < g1) contract(s);

    // ---- the next tile's first slots go out BEFORE this tile's epilogue.  The epilogue's only load — the thread's four
    // bias values (its column quad is the same in every round: T is a multiple of BN / 4) — goes out before them: loads
    // return in order, and waiting for a bias requested behind the prefetch would drain the prefetch.
    const long long tn = t + gsz;
    const bool more = tn < total;   // wave-uniform
    const int row_e = row0, col_e = col0;
    static_assert(T % (BN / 4) == 0, "a thread keeps its column quad over the epilogue's rounds");
    const int c4_mine = (tid % (BN / 4)) * 4;
    // (through a descriptor over exactly the N bias values — none: zero records —, UNCONDITIONALLY and consumed on every
    // path below: a load that some path leaves un-waited would be "possibly pending" at the loop head and cost every round
    // of the main loop a vmcnt(0) before it may touch that register)
    const dvis_f4 bias4 = __builtin_bit_cast(
        dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (unsigned)(col_e + c4_mine) * 4u, 0, 0));
    if (more) {
      row0 = (int)(tn / p.col_blocks) * BM, col0 = (int)(tn % p.col_blocks) * BN;
      ra = desc_a(row0), rw = desc_w(col0);
#pragma unroll
      for (int s = 0; s < PF; ++s) load(s, s);
    }
    // ---- epilogue of tile (row_e, col_e): transpose through LDS, bias / residual / ReLU, 16-byte stores
    __syncthreads();   // the previous tile's reads of the buffer are done
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(rt * 16 + g * 4 + r) * BN + ct * 16 + i] = acc[rt][ct][r];
    __syncthreads();
    constexpr int U = BM * BN / 4, UR = BN / 4;
#pragma unroll 4
    for (int u = tid; u < U; u += T) {
      const int row = u / UR, c4 = (u - row * UR) * 4;
      dvis_f4 sv = *reinterpret_cast<const dvis_f4 *>(lds + row * BN + c4);
      const int gr = row_e + row, gc = col_e + c4;
      if (gr >= p.M || gc >= p.N) continue;
      float *dst = p.head_d ? Cb + (long long)(gc / p.head_d) * p.head_stride + (long long)gr * p.head_d + gc % p.head_d
                            : Cb + (long long)gr * p.ldc + gc;
      if (p.vec_store) {
        sv = dvis_f4{sv[0] + bias4[0], sv[1] + bias4[1], sv[2] + bias4[2], sv[3] + bias4[3]};
        if (Rb) {
          const dvis_f4 tq = *reinterpret_cast<const dvis_f4 *>(Rb + (long long)gr * p.ldres + gc);
          sv = dvis_f4{sv[0] + tq[0], sv[1] + tq[1], sv[2] + tq[2], sv[3] + tq[3]};
        }
        if (p.act) sv = dvis_f4{fmaxf(sv[0], 0.f), fmaxf(sv[1], 0.f), fmaxf(sv[2], 0.f), fmaxf(sv[3], 0.f)};
        *reinterpret_cast<dvis_f4 *>(dst) = sv;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (gc + e >= p.N) break;
          float v = sv[e] + bias4[e];
          if (Rb) v += Rb[(long long)gr * p.ldres + gc + e];
          if (p.act) v = fmaxf(v, 0.f);
          dst[e] = v;
        }
      }
    }
    if (!more) break;
    t = tn;
  }
}

struct Config {
  int rt, ct, nw;
  void (*kernel16)(const GemmArgs);
  void (*kernel4)(const GemmArgs);
  void (*persist16)(const GemmArgs);   // NW == 1 only
  void (*persist4)(const GemmArgs);
};
template <int RT, int CT, int NW, int PF, bool K16> constexpr auto persist_or_null() -> void (*)(const GemmArgs) {
  if constexpr (NW == 1) return gemm_nt_persist_kernel<RT, CT, PF, K16>;
  else return nullptr;
}

#define DVIS_GEMM_CFG(RT_, CT_, NW_, PF_)                                                                      \
  { RT_, CT_, NW_, gemm_nt_kernel<RT_, CT_, NW_, PF_, true>, gemm_nt_kernel<RT_, CT_, NW_, PF_, false>,            \
    persist_or_null<RT_, CT_, NW_, PF_, true>(), persist_or_null<RT_, CT_, NW_, PF_, false>() }
const Config kConfigs[] = {
    DVIS_GEMM_CFG(1, 1, 4, 4),   // 0: 16 x 16
    DVIS_GEMM_CFG(1, 1, 8, 4),   // 1
    DVIS_GEMM_CFG(2, 1, 8, 4),   // 2: 32 x 16
    DVIS_GEMM_CFG(2, 2, 4, 4),   // 3: 32 x 32
    DVIS_GEMM_CFG(2, 2, 8, 2),   // 4
    DVIS_GEMM_CFG(4, 1, 8, 4),   // 5: 64 x 16
    DVIS_GEMM_CFG(4, 2, 4, 2),   // 6: 64 x 32
    DVIS_GEMM_CFG(4, 2, 8, 2),   // 7
    DVIS_GEMM_CFG(4, 4, 4, 2),   // 8: 64 x 64
    DVIS_GEMM_CFG(4, 4, 2, 2),   // 9
    DVIS_GEMM_CFG(7, 1, 8, 2),   // 10: 112 x 16 (all rows of a 100-query block)
    DVIS_GEMM_CFG(7, 2, 4, 2),   // 11: 112 x 32
    DVIS_GEMM_CFG(4, 4, 1, 2),   // 12: 64 x 64, one wave, K not split (short K, tall M)
    DVIS_GEMM_CFG(8, 4, 2, 2),   // 13: 128 x 64
    DVIS_GEMM_CFG(8, 4, 1, 2),   // 14
    DVIS_GEMM_CFG(6, 4, 1, 2),   // 15: 96 x 64, one wave: 96 accumulators + two fragment slots fit 2 waves per SIMD
    DVIS_GEMM_CFG(5, 4, 1, 2),   // 16: 80 x 64
};
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);

// Tile configuration from the problem size alone (so that a shape always runs the same summation order).
int pick_config(int M, int N, int K, int batch) {
  auto wgs = [&](int c) {
    const long long rb = (M + 16 * kConfigs[c].rt - 1) / (16 * kConfigs[c].rt);
    const long long cb = (N + 16 * kConfigs[c].ct - 1) / (16 * kConfigs[c].ct);
    return rb * cb * batch;
  };
  // tall problems (A streams from HBM, tile order column-fastest): ONE wave per 128 x 64 tile, K unsplit — no LDS reduction,
  // every A row block read by N / 64 neighbouring workgroups — in the PERSISTENT form (gemm_nt_persist_kernel).  Measured at
  // 579 600 rows: 0.85 - 0.96 of hipBLASLt (the 80 x 64 / 64 x 64 tiles lose in that form: 0.80 / 0.53)
  if ((long long)M * K * 4 > (64ll << 20) && N >= 64) return 14;
  // the largest tile that still gives every CU a workgroup (half of them when a long K keeps each workgroup busy); below
  // that, smaller tiles with a deeper K split
  if (wgs(8) >= 256 || (K >= 1024 && wgs(8) >= 128)) return 8;
  if (wgs(6) >= 256) return 6;
  if (wgs(3) >= 192) return K >= 1024 ? 4 : 3;
  if (wgs(2) >= 96) return 2;
  return K >= 256 ? 1 : 0;
}

// Tile configuration inside ONE K-split family: `nw` waves split K, so every configuration this returns sums an output
// element's products in the same order — the result of a row depends on (N, K, nw) only, NOT on how many rows share the call.
// Phase A (the per-frame segmenter: frames are the batch, video_mask2former_transformer_decoder.py:327-335) takes its small
// GEMMs from here, so a frame gets the same bits alone, in a 30-frame clip or in a rank's 4-frame shard.
//   nw = 1: one-wave tiles, K unsplit (tall projections over every pixel: 64 x 64 / 128 x 64)
//   nw = 4: 16 x 16 / 32 x 32 / 64 x 32 / 64 x 64      nw = 8: 16 x 16 / 32 x 16 / 32 x 32 / 64 x 32
int pick_config_nw(int M, int N, int K, int batch, int nw) {
  auto wgs = [&](int c) {
    const long long rb = (M + 16 * kConfigs[c].rt - 1) / (16 * kConfigs[c].rt);
    const long long cb = (N + 16 * kConfigs[c].ct - 1) / (16 * kConfigs[c].ct);
    return rb * cb * batch;
  };
  if (nw == 1) return ((long long)M * K * 4 > (64ll << 20) && N >= 64) || wgs(14) >= 1024 ? 14 : 12;
  if (nw == 4) {
    if (wgs(8) >= 256 || (K >= 1024 && wgs(8) >= 128)) return 8;
    if (wgs(6) >= 256) return 6;
    if (wgs(3) >= 192) return 3;
    return 0;
  }
  if (wgs(7) >= 256) return 7;
  if (wgs(4) >= 192) return 4;
  if (wgs(2) >= 96) return 2;
  return 1;
}

}  // namespace

DVIS_EXPORT int dvis_gemm_num_configs(void) { return kNumConfigs; }

DVIS_EXPORT int dvis_gemm_pick_config_nw(int M, int N, int K, int batch, int nw) {
  if (nw != 1 && nw != 4 && nw != 8) return -1;
  return pick_config_nw(M, N, K, batch > 0 ? batch : 1, nw);
}

DVIS_EXPORT int dvis_gemm_config_waves(int config) { return config >= 0 && config < kNumConfigs ? kConfigs[config].nw : -1; }

DVIS_EXPORT int dvis_gemm_pick_config(int M, int N, int K, int batch) { return pick_config(M, N, K, batch > 0 ? batch : 1); }

static int gemm_launch(const float *A, int64_t lda, int64_t strideA, const float *W, int64_t ldw, int64_t strideW,
                       const float *bias, int64_t strideBias, const float *res, int64_t ldres, int64_t strideRes, float *C,
                       int64_t ldc, int64_t strideC, int M, int N, int K, int batch, int act, int config, int head_d,
                       int64_t head_stride, void *stream) {
  DVIS_REQUIRE(M >= 0 && N >= 0 && K > 0 && batch >= 0, "gemm_nt: bad sizes (M=%d N=%d K=%d batch=%d)", M, N, K, batch);
  if (M == 0 || N == 0 || batch == 0) return DVIS_OK;
  DVIS_REQUIRE(A && W && C, "gemm_nt: null pointer");
  DVIS_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && strideA % 4 == 0 && strideW % 4 == 0 &&
                   ((uintptr_t)A | (uintptr_t)W) % 16 == 0,
               "gemm_nt: needs K, lda, ldw, batch strides %% 4 == 0 and 16-byte aligned A / W (K=%d lda=%lld ldw=%lld)", K,
               (long long)lda, (long long)ldw);
  DVIS_REQUIRE(lda >= K && ldw >= K && (head_d || ldc >= N) && (!res || ldres >= N),
               "gemm_nt: row strides shorter than the rows");
  DVIS_REQUIRE(batch <= 65535, "gemm_nt: batch <= 65535");
  const int c = config >= 0 ? config : pick_config(M, N, K, batch);
  DVIS_REQUIRE(c < kNumConfigs, "gemm_nt: configuration %d does not exist", c);
  const Config &cf = kConfigs[c];
  const int BM = 16 * cf.rt, BN = 16 * cf.ct;
  DVIS_REQUIRE((long long)BM * lda * 4 < (1ll << 31) && (long long)BN * ldw * 4 < (1ll << 31),
               "gemm_nt: one tile's rows must span less than 2 GiB");
  GemmArgs p;
  p.A = A, p.W = W, p.bias = bias, p.res = res, p.C = C;
  p.lda = lda, p.ldw = ldw, p.ldres = ldres, p.ldc = ldc;
  p.sA = strideA, p.sW = strideW, p.sRes = strideRes, p.sC = strideC, p.sBias = strideBias;
  p.M = M, p.N = N, p.K = K, p.act = act;
  DVIS_REQUIRE(head_d == 0 || (head_d % 4 == 0 && N % head_d == 0 && !res && batch == 1 && head_stride >= (int64_t)M * head_d &&
                               head_stride % 4 == 0),
               "gemm_nt: head-major output needs head_d %% 4 == 0, N %% head_d == 0, no residual, batch 1");
  p.head_d = head_d, p.head_stride = head_stride;
  p.row_blocks = (M + BM - 1) / BM;
  p.col_blocks = (N + BN - 1) / BN;
  // A larger than what the caches keep between two passes over it (L2s 32 MB; the 256 MB MALL is shared with C and W)
  p.col_fastest = (long long)M * K * 4 > (64ll << 20) && p.col_blocks > 1;
  p.vec_store = N % 4 == 0 && ldc % 4 == 0 && strideC % 4 == 0 && (uintptr_t)C % 16 == 0 &&
                (!bias || ((uintptr_t)bias % 16 == 0 && strideBias % 4 == 0)) &&
                (!res || (ldres % 4 == 0 && strideRes % 4 == 0 && (uintptr_t)res % 16 == 0));
  const long long tiles = (long long)p.row_blocks * p.col_blocks;
  DVIS_REQUIRE(tiles < (1ll << 31), "gemm_nt: too many tiles");
  const size_t lds = (size_t)cf.nw * BM * BN * sizeof(float);
  auto kernel = K % 16 == 0 ? cf.kernel16 : cf.kernel4;
  static DvisLdsOptIn opted[kNumConfigs][2];
  if (const int rc = dvis_lds_opt_in(reinterpret_cast<const void *>(kernel), lds, &opted[c][K % 16 == 0 ? 0 : 1], "gemm_nt"))
    return rc;
  // persistent form: one-wave tiles of a tall problem (see gemm_nt_persist_kernel); DVIS_GEMM_PERSIST=0 switches it off
  static const bool persist_on = []() { const char *e = getenv("DVIS_GEMM_PERSIST"); return !(e && e[0] == '0'); }();
  auto pk = K % 16 == 0 ? cf.persist16 : cf.persist4;
  if (persist_on && pk != nullptr && p.col_fastest) {
    static int cus = 0;
    if (cus == 0) {
      int dev = 0, n = 0;
      hipGetDevice(&dev);
      cus = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
    }
    const int tiles_cnt = cf.rt * cf.ct;
    const int occ = tiles_cnt >= 32 ? 1 : (tiles_cnt >= 20 ? 2 : (tiles_cnt >= 12 ? 3 : 4));   // = kGemmMinWaves
    const size_t lds1 = (size_t)BM * BN * sizeof(float);
    long long per_cu = 4ll * occ;
    if ((long long)(160 * 1024 / lds1) < per_cu) per_cu = (long long)(160 * 1024 / lds1);
    long long grid = (long long)cus * per_cu;
    if (grid > tiles) grid = tiles;
    grid &= ~7ll;
    if (grid >= 8) {
      hipLaunchKernelGGL(pk, dim3((unsigned)grid, (unsigned)batch), dim3(64), lds1, (hipStream_t)stream, p);
      return dvis_check_launch("gemm_nt_persist_kernel");
    }
  }
  hipLaunchKernelGGL(kernel, dim3((unsigned)tiles, (unsigned)batch), dim3(64 * cf.nw), lds, (hipStream_t)stream, p);
  return dvis_check_launch("gemm_nt_kernel");
}

DVIS_EXPORT int dvis_gemm_nt_hm(const float *A, int64_t lda, int64_t strideA, const float *W, int64_t ldw, int64_t strideW,
                                const float *bias, const float *res, int64_t ldres, int64_t strideRes, float *C,
                                int64_t ldc, int64_t strideC, int M, int N, int K, int batch, int act, int config,
                                int head_d, int64_t head_stride, void *stream) {
  return gemm_launch(A, lda, strideA, W, ldw, strideW, bias, 0, res, ldres, strideRes, C, ldc, strideC, M, N, K, batch, act,
                     config, head_d, head_stride, stream);
}

DVIS_EXPORT int dvis_gemm_nt(const float *A, int64_t lda, int64_t strideA, const float *W, int64_t ldw, int64_t strideW,
                             const float *bias, const float *res, int64_t ldres, int64_t strideRes, float *C,
                             int64_t ldc, int64_t strideC, int M, int N, int K, int batch, int act, int config,
                             void *stream) {
  return gemm_launch(A, lda, strideA, W, ldw, strideW, bias, 0, res, ldres, strideRes, C, ldc, strideC, M, N, K, batch, act,
                     config, 0, 0, stream);
}

DVIS_EXPORT int dvis_gemm_nt_bb(const float *A, int64_t lda, int64_t strideA, const float *W, int64_t ldw, int64_t strideW,
                                const float *bias, int64_t strideBias, const float *res, int64_t ldres, int64_t strideRes,
                                float *C, int64_t ldc, int64_t strideC, int M, int N, int K, int batch, int act, int config,
                                void *stream) {
  return gemm_launch(A, lda, strideA, W, ldw, strideW, bias, strideBias, res, ldres, strideRes, C, ldc, strideC, M, N, K,
                     batch, act, config, 0, 0, stream);
}
