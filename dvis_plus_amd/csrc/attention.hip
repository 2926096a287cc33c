// softmax(Q K^T * scale [masked]) V for small-query / long-key attention — gfx950, exact-fp32 MFMA.
//
// Replaces the core of nn.MultiheadAttention as the reference uses it (need_weights=True slow path that
// materialises (B*heads, Q, HW) probabilities): masked cross-attention of the Mask2Former decoder
// (mask2former_video/modeling/transformer_decoder/video_mask2former_transformer_decoder.py:99-111, called from
// dvis_Plus/video_mask2former_transformer_decoder.py:299), the referring cross-attention and self-attention of
// the tracker (dvis_Plus/tracker.py:35-53) and the time / object / cross attention of the temporal refiner
// (dvis_Plus/refiner.py:104-139).  Projections stay library GEMMs; this kernel is the part in between.
//
// Shape regime: few queries (Q = 100/200, or T = 30), head dim 32 or 64, keys from 30 to 14 720.
//   * A wave owns 16 queries; a workgroup (8 waves) 128 queries of one (batch, head) and one key range
//     (flash-decoding style split over keys; partial (O, m, l) are merged by a second tiny kernel).
//   * S^T = K Q^T is computed instead of S (swapped operands): the accumulator then holds, per lane, 4 keys of
//     ONE query, which (a) makes the row max/sum a 2-shuffle reduction and (b) is already the A-operand layout
//     of P V — probabilities never leave registers.  K index permutation as in mask_gemm.hip (lane group g sums
//     dims [g*d/4, (g+1)*d/4)) so Q / K fragments are contiguous 16-byte reads.
//   * K / V stages of 64 (d=32) or 32 (d=64) keys go through LDS once per workgroup (all 8 waves reuse them), next stage prefetched
//     into registers under the current stage's MFMAs.
//   * mask: uint8 (1 = blocked), ONE copy per frame shared by all heads (the reference repeats it 8x);
//     a row whose allowed_count is 0 ignores the mask = the reference's "fully masked row" reset (…:297), done
//     on the device without the torch.where host sync.
//   * v_mfma_f32_16x16x4_f32 = fp32 fma chain, online softmax in fp32 (base 2, v_exp_f32): parity well inside 1e-3.
#include <math.h>

#include <type_traits>

#include "dvis_common.h"
#include "x3_common.h"

namespace {

// The softmax runs in base 2: q is pre-scaled by scale * log2(e), so exp(s - m) is ONE v_exp_f32 (a quarter-rate
// transcendental) per score instead of expf's range reduction around it — the VALU work per 16 x 16 score tile was
// 154 instructions against 16 MFMAs (PMC, decoder cross-attention), i.e. as many issue cycles as the matrix work.
// Row statistics (m, and the combine kernel's weights) live in the same base-2 units; the normalised output does not care.
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

constexpr int kKT = 64;   // keys per LDS stage

struct dvis_strides {
  int64_t b, h, r;   // floats between batch entries / heads / rows (last dim contiguous)
};

struct SplitPlan {
  int nsplit, keys_per_split, qchunks;
};

// How the keys of one (batch, head) are split over workgroups — a function of (Lq, Lk) ONLY, never of the batch: the merged
// result depends on where the splits fall, and a frame of the per-frame segmenter (frames = batch) must get the same bits
// whether it is computed alone, in a 30-frame clip or in a rank's 4-frame shard (north_star's frame sharding).  (Rounds 1 - 4
// sized the split from the number of (batch, head) pairs to fill the chip at every batch size: batch-dependent bits.)
SplitPlan plan_split(int /*BH*/, int Lq, int Lk) {
  SplitPlan p;
  p.qchunks = (Lq + 127) / 128;
  int ns = (16 + p.qchunks - 1) / p.qchunks;            // >= 16 workgroups per (batch, head) from chunks x splits
  const int max_ns = (Lk + 4 * kKT - 1) / (4 * kKT);   // at least 256 keys per split
  if (ns > max_ns) ns = max_ns;
  if (ns < 1) ns = 1;
  int kps = ((Lk + ns - 1) / ns + kKT - 1) / kKT * kKT;
  p.nsplit = (Lk + kps - 1) / kps;
  p.keys_per_split = kps;
  return p;
}

template <int DH, bool SHORT>
__global__ __launch_bounds__(512) void attn_fwd_kernel(
    const float *__restrict__ q, dvis_strides qs, const float *__restrict__ k, dvis_strides ks_, const float *__restrict__ v,
    dvis_strides vs, float *__restrict__ out, dvis_strides os, const uint8_t *__restrict__ mask,
    const int *__restrict__ allowed, int heads, int Lq, int Lk, float scale, int nsplit, int keys_per_split,
    float *__restrict__ ws_o, float *__restrict__ ws_ml) {
  constexpr int DQ = DH / 4;        // dims per lane group
  constexpr int NT = DH / 16;       // output N tiles
  constexpr int LS = DH + 4;        // LDS row stride (floats): 16-B aligned, V rows of lane groups 0/1 split banks
  // keys per LDS stage.  Long key sequences: 64 (d=32) / 32 (d=64) with a register prefetch of the next stage.
  // (64 keys at d=64 was measured on the ViT-L shape — 3681 tokens, 16 heads: 17.83 ms either way, 93 TFLOP/s — so
  // the smaller stage is kept for its 17 KB of LDS.  An earlier form of the prefetch — float4 arrays with a zero-fill
  // branch — crashed hipcc 7.2's machine-copy-propagation pass at F4 = 2; the scalar form below compiles.)  SHORT (Lk <= 128: tracker / refiner / decoder self-attention): ONE 128-key stage written
  // straight to LDS, so the whole call pays a single global-load latency instead of one per 32-key stage.
  constexpr int KT = SHORT ? 128 : (DH == 64 ? 32 : 64);
  constexpr int F4 = SHORT ? 1 : KT * DH / 4 / 512;   // float4 per thread per matrix per stage (prefetch registers)
  __shared__ float k_lds[KT * LS];
  __shared__ float v_lds[KT * LS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int split = blockIdx.x, bh = blockIdx.y;
  const int bi = bh / heads, hi = bh - bi * heads;   // (batch entry, head)
  const int q0 = blockIdx.z * 128 + wv * 16;
  const bool wave_on = q0 < Lq;
  const int myq = q0 + j;                // the query this lane's accumulator COLUMN belongs to
  const bool q_ok = myq < Lq;
  const int key_lo = split * keys_per_split;
  const int key_hi = min(Lk, key_lo + keys_per_split);

  // ---- B operand of S^T: Q[myq][g*DQ + kk] * scale (torch scales q before the product)
  float qf[DQ];
  {
    const float *qrow = q + (size_t)bi * qs.b + (size_t)hi * qs.h + (size_t)(q_ok ? myq : 0) * qs.r + g * DQ;
#pragma unroll
    for (int c = 0; c < DQ / 4; ++c) {
      const float4 t = *reinterpret_cast<const float4 *>(qrow + 4 * c);
      qf[4 * c] = q_ok ? t.x * (scale * kLog2e) : 0.f;
      qf[4 * c + 1] = q_ok ? t.y * (scale * kLog2e) : 0.f;
      qf[4 * c + 2] = q_ok ? t.z * (scale * kLog2e) : 0.f;
      qf[4 * c + 3] = q_ok ? t.w * (scale * kLog2e) : 0.f;
    }
  }
  const int mb = bi;
  const bool use_mask = mask != nullptr && q_ok && (allowed == nullptr || allowed[(size_t)mb * Lq + myq] != 0);
  const uint8_t *mrow = mask ? mask + ((size_t)mb * Lq + (q_ok ? myq : 0)) * Lk : nullptr;
  const bool lk4 = (Lk & 3) == 0;

  dvis_f4 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = dvis_f4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;

  const float *kb = k + (size_t)bi * ks_.b + (size_t)hi * ks_.h;
  const float *vb = v + (size_t)bi * vs.b + (size_t)hi * vs.h;
  // prefetch registers as scalars (F4 <= 2): as arrays captured by a lambda hipcc keeps them in scratch
  static_assert(F4 == 1 || F4 == 2, "stage size");
  float4 pk0, pv0, pk1, pv1;
  pk0 = pv0 = pk1 = pv1 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage_addr = [&](int i, int &row, int &c4) {
    const int e = tid + 512 * i;                // float4 index within the stage
    row = e / (DH / 4);
    c4 = e - row * (DH / 4);
  };
  auto prefetch = [&](int ks) {
    // rows past the split's last key re-read that last key: they are masked out of the softmax below (p = 0 exactly)
    // and real rows are finite, so no zero fill / branch is needed
    int row, c4;
    stage_addr(0, row, c4);
    int key = min(ks + row, key_hi - 1);
    pk0 = *reinterpret_cast<const float4 *>(kb + (size_t)key * ks_.r + 4 * c4);
    pv0 = *reinterpret_cast<const float4 *>(vb + (size_t)key * vs.r + 4 * c4);
    if (F4 > 1) {
      stage_addr(1, row, c4);
      key = min(ks + row, key_hi - 1);
      pk1 = *reinterpret_cast<const float4 *>(kb + (size_t)key * ks_.r + 4 * c4);
      pv1 = *reinterpret_cast<const float4 *>(vb + (size_t)key * vs.r + 4 * c4);
    }
  };
  if (!SHORT) prefetch(key_lo);
  for (int ks = key_lo; ks < key_hi; ks += KT) {
    __syncthreads();
    if (SHORT) {
      for (int e = tid; e < KT * DH / 4; e += 512) {
        const int row = e / (DH / 4), c4 = e - row * (DH / 4);
        const int key = ks + row;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (key < key_hi) {
          a = *reinterpret_cast<const float4 *>(kb + (size_t)key * ks_.r + 4 * c4);
          b = *reinterpret_cast<const float4 *>(vb + (size_t)key * vs.r + 4 * c4);
        }
        *reinterpret_cast<float4 *>(&k_lds[row * LS + 4 * c4]) = a;
        *reinterpret_cast<float4 *>(&v_lds[row * LS + 4 * c4]) = b;
      }
    } else {
      int row, c4;
      stage_addr(0, row, c4);
      *reinterpret_cast<float4 *>(&k_lds[row * LS + 4 * c4]) = pk0;
      *reinterpret_cast<float4 *>(&v_lds[row * LS + 4 * c4]) = pv0;
      if (F4 > 1) {
        stage_addr(1, row, c4);
        *reinterpret_cast<float4 *>(&k_lds[row * LS + 4 * c4]) = pk1;
        *reinterpret_cast<float4 *>(&v_lds[row * LS + 4 * c4]) = pv1;
      }
    }
    __syncthreads();
    if (!SHORT && ks + KT < key_hi) prefetch(ks + KT);
    if (!wave_on) continue;
    // Software pipeline over the key tiles of the stage: the score MFMAs of tile kt + 1 are ISSUED before the softmax
    // VALU work of tile kt, so the matrix pipe executes them in the shadow of that VALU work (an MFMA is asynchronous;
    // independent VALU instructions of the same wave issue while it runs).  VALU and MFMA of one wave did not overlap at
    // all before: PMC showed their busy cycles adding up to the kernel time.
    auto score_tile = [&](int kt) -> dvis_f4 {     // S^T tile: rows = 16 keys, cols = 16 queries
      dvis_f4 sc = dvis_f4{0.f, 0.f, 0.f, 0.f};
      const float *krow = &k_lds[(kt * 16 + j) * LS + g * DQ];
#pragma unroll
      for (int c = 0; c < DQ / 4; ++c) {
        const float4 kk = *reinterpret_cast<const float4 *>(krow + 4 * c);
        sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.x, qf[4 * c], sc, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.y, qf[4 * c + 1], sc, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.z, qf[4 * c + 2], sc, 0, 0, 0);
        sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.w, qf[4 * c + 3], sc, 0, 0, 0);
      }
      return sc;
    };
    // mask bytes of this lane's 4 keys of a FULL tile (one dword), fetched one tile ahead like the scores
    auto mask_word = [&](int kt) -> unsigned {
      const int k0 = ks + kt * 16;
      return (mask != nullptr && lk4 && k0 + 16 <= key_hi) ? *reinterpret_cast<const unsigned *>(mrow + k0 + 4 * g) : 0u;
    };
    dvis_f4 s_next = score_tile(0);
    unsigned mw_next = mask_word(0);
#pragma unroll 1
    for (int kt = 0; kt < KT / 16; ++kt) {
      const int key0 = ks + kt * 16;
      if (key0 >= key_hi) break;   // uniform
      const dvis_f4 s = s_next;
      const unsigned mw_cur = mw_next;
      if (kt + 1 < KT / 16 && key0 + 16 < key_hi) {      // uniform; in flight under the softmax below
        s_next = score_tile(kt + 1);
        mw_next = mask_word(kt + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      // lane (j, g) now holds S[query myq][key0 + 4g + r], r = 0..3.  Dead keys (masked, or past the split's end) are
      // set to -inf ONCE, branch-free for the common full tile; 2^(-inf - m) is exactly 0, so the probabilities need no
      // second select.  (Uniform branches only: `mask`, the tail tile.)
      const int kbase = key0 + 4 * g;
      float sv[4] = {s[0], s[1], s[2], s[3]};
      const bool full = key0 + 16 <= key_hi;                  // wave-uniform
      if (mask != nullptr) {                                  // wave-uniform
        unsigned mw;
        if (lk4 && full) {
          mw = mw_cur;
        } else {
          mw = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) mw |= (kbase + r < key_hi && mrow[kbase + r] != 0) ? (0xffu << (8 * r)) : 0u;
        }
        mw = use_mask ? mw : 0u;                              // rows that ignore the mask (allowed_count == 0)
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = (mw & (0xffu << (8 * r))) ? -INFINITY : sv[r];
      }
      if (!full) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = kbase + r >= key_hi ? -INFINITY : sv[r];
      }
      float tmax = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run, tmax);
      // alpha = 2^(m_run - m_new): 1 unless this tile raised the row maximum (rare after the first tiles) — the
      // transcendental is only issued when some lane needs it (wave-uniform branch)
      float alpha = 1.f;
      if (!__all(m_new == m_run)) alpha = (m_new == -INFINITY) ? 1.f : ex2(m_run - m_new);
      const float m_sub = (m_new == -INFINITY) ? 0.f : m_new;   // no live key yet: -inf - 0 = -inf, not NaN
      float p[4], psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = ex2(sv[r] - m_sub);
        psum += p[r];
      }
      l_part = l_part * alpha + psum;
      m_run = m_new;
      // rescale O rows: row (4g + r) of the accumulator belongs to query q0 + 4g + r, whose alpha lives in lane 4g + r
      if (!__all(alpha == 1.f)) {
        const dvis_f4 av = dvis_f4{__shfl(alpha, 4 * g), __shfl(alpha, 4 * g + 1), __shfl(alpha, 4 * g + 2),
                                   __shfl(alpha, 4 * g + 3)};
#pragma unroll
        for (int n = 0; n < NT; ++n) o[n] = o[n] * av;
      }
      // ---- O += P V : A = P (already in A layout), B = V rows key0 + 4g + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float *vrow = &v_lds[(kt * 16 + 4 * g + r) * LS + j];
#pragma unroll
        for (int n = 0; n < NT; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], vrow[16 * n], o[n], 0, 0, 0);
      }
    }
  }
  if (!wave_on) return;

  // ---- epilogue: l over the 4 lane groups; stats of query (4g + r) come from lane 4g + r
  float l_tot = l_part + __shfl_xor(l_part, 16);
  l_tot += __shfl_xor(l_tot, 32);
  float lr[4], mr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lr[r] = __shfl(l_tot, 4 * g + r);
    mr[r] = __shfl(m_run, 4 * g + r);
  }
  if (nsplit == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < Lq) {
        const float inv = lr[r] > 0.f ? 1.f / lr[r] : 0.f;
        float *orow = out + (size_t)bi * os.b + (size_t)hi * os.h + (size_t)qq * os.r;
#pragma unroll
        for (int n = 0; n < NT; ++n) orow[16 * n + j] = o[n][r] * inv;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < Lq) {
        const size_t row = ((size_t)bh * nsplit + split) * Lq + qq;
#pragma unroll
        for (int n = 0; n < NT; ++n) ws_o[row * DH + 16 * n + j] = o[n][r];
        if (j == 0) {
          ws_ml[row * 2] = mr[r];
          ws_ml[row * 2 + 1] = lr[r];
        }
      }
    }
  }
}

// Few queries, long key sequences, d = 32 (the decoder's masked cross-attention: 100 / 200 queries, 920 ... 14 720 keys):
// partitioned by KEYS.  A wave owns a key range of one (batch, head) and ALL query tiles of its chunk (QT = 7 tiles of 16
// = 112 queries): Q fragments (56 VGPRs), O accumulators (56) and the row statistics of all 7 tiles stay in registers,
// and a 16-key tile of K and V — needed by no other wave — goes from global memory straight into the MFMA operand
// layouts, fetched one tile ahead: no LDS, no barrier.  Every wave is its own flash-decoding split (partials merged by
// attn_combine_kernel).  Against the query-partitioned kernel above: 100 of 112 issued rows are useful instead of 100 of
// 128 with one wave in eight idle, and a K/V fragment is loaded once per 112 MFMAs.
// The row maximum is LAZY: probabilities are taken relative to a reference m_ref that is only raised (cross-lane
// reduction + rescale of O and l) when some score of the tile exceeds it by more than 2^kLazy — softmax is invariant to
// the reference, and the common tile then costs no lane exchange at all (2 ds_bpermute + waits per tile before).
constexpr float kLazy = 16.f;   // scores are in log2 units: p <= 2^16 between two raises, l <= Lk * 2^16

struct KeySplitPlan {
  int nsplit, keys_per_split, qchunks;
};
constexpr int kQT = 7;

// EIGHT splits of the keys (at least 8 tiles of 16 keys each), whatever the batch (see plan_split): 7 / 8 / 8 splits at the R50
// levels of 920 / 3680 / 14 720 keys — what the batch-sized rule of rounds 1 - 4 chose at the benchmark's 30 frames x 8 heads
// (1680 - 1920 waves for the chip's 2048 wave slots); a single frame runs 56 - 64 waves.  Measured at 30 frames
// (profiles/r05_attn_key_splits.txt): splits of 32 tiles (29 at the finest level) 691 / 156 / 139 us per level, of 128 tiles
// 585 / 496 / 254 us: the finest level wants few long splits (partials: 13 KB written and re-read per split), the coarse ones
// want their 7 - 8.  DVIS_ATTN_KEY_SPLITS (development): another split count.
KeySplitPlan plan_keysplit(int /*BH*/, int Lq, int Lk) {
  KeySplitPlan p;
  p.qchunks = ((Lq + 15) / 16 + kQT - 1) / kQT;
  const int tiles = (Lk + 15) / 16;
  static const int want = []() { const char *e = getenv("DVIS_ATTN_KEY_SPLITS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
  int ns = tiles / 8 < want ? tiles / 8 : want;
  if (ns < 1) ns = 1;
  const int kps = (tiles + ns - 1) / ns * 16;
  p.nsplit = (Lk + kps - 1) / kps;
  p.keys_per_split = kps;
  return p;
}

bool keysplit_applies(int Lq, int Lk, int d, bool has_mask, int64_t k_row, int64_t v_row) {
  return d == 32 && Lq > 64 && Lk >= 512 && (!has_mask || (Lk & 3) == 0) && (long long)Lk * k_row * 4 < (1ll << 31) &&
         (long long)Lk * v_row * 4 < (1ll << 31) && (long long)Lq * Lk < (1ll << 31);
}

__global__ __launch_bounds__(512) void attn_keysplit_kernel(
    const float *__restrict__ q, dvis_strides qs, const float *__restrict__ k, dvis_strides ks_, const float *__restrict__ v,
    dvis_strides vs, float *__restrict__ out, dvis_strides os, const uint8_t *__restrict__ mask,
    const int *__restrict__ allowed, int heads, int Lq, int Lk, float scale, int nsplit, int keys_per_split, int qchunks,
    int total_units, float *__restrict__ ws_o, float *__restrict__ ws_ml) {
  constexpr int DH = 32, DQ = 8, NT = 2, QT = kQT;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int unit = blockIdx.x * 8 + wv;
  if (unit >= total_units) return;   // wave-uniform; the kernel has no barrier
  const int split = unit % nsplit;
  const int t = unit / nsplit;
  const int qc = t % qchunks, bh = t / qchunks;
  const int bi = bh / heads, hi = bh - bi * heads;
  const int q_base = qc * (QT * 16);
  const int key_lo = split * keys_per_split;
  const int key_hi = min(Lk, key_lo + keys_per_split);

  // ---- B operands of S^T for all query tiles: Q[q_base + 16 qt + j][g*DQ + kk] * scale * log2(e)
  float qf[QT][DQ];
  unsigned use_mask_bits = 0;
  unsigned moff[QT];   // byte offset of this lane's query row in the frame's mask, + 4g
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int myq = q_base + qt * 16 + j;
    const bool q_ok = myq < Lq;
    const float *qrow = q + (size_t)bi * qs.b + (size_t)hi * qs.h + (size_t)(q_ok ? myq : 0) * qs.r + g * DQ;
#pragma unroll
    for (int c = 0; c < DQ / 4; ++c) {
      const float4 tq = *reinterpret_cast<const float4 *>(qrow + 4 * c);
      qf[qt][4 * c] = q_ok ? tq.x * (scale * kLog2e) : 0.f;
      qf[qt][4 * c + 1] = q_ok ? tq.y * (scale * kLog2e) : 0.f;
      qf[qt][4 * c + 2] = q_ok ? tq.z * (scale * kLog2e) : 0.f;
      qf[qt][4 * c + 3] = q_ok ? tq.w * (scale * kLog2e) : 0.f;
    }
    const bool um = mask != nullptr && q_ok && (allowed == nullptr || allowed[(size_t)bi * Lq + myq] != 0);
    use_mask_bits |= um ? (1u << qt) : 0u;
    moff[qt] = (unsigned)((q_ok ? myq : 0) * Lk + 4 * g);
  }
  const __amdgpu_buffer_rsrc_t rk = dvis_make_rsrc_uniform(k + (size_t)bi * ks_.b + (size_t)hi * ks_.h, 0x7FFFFFFFu);
  const __amdgpu_buffer_rsrc_t rv = dvis_make_rsrc_uniform(v + (size_t)bi * vs.b + (size_t)hi * vs.h, 0x7FFFFFFFu);
  // mask rows of this frame; a word that starts past the last row's end reads 0 (tail tile of the last query)
  const __amdgpu_buffer_rsrc_t rm =
      dvis_make_rsrc_uniform(mask ? mask + (size_t)bi * Lq * Lk : nullptr, mask ? (unsigned)((size_t)Lq * Lk) : 0u);
  const unsigned k_row_bytes = (unsigned)(ks_.r * 4), v_row_bytes = (unsigned)(vs.r * 4);

  // fragments of one 16-key tile: K as A operand (lane (j, g): key j, dims g*8 ..), V as B operand (lane (j, g): keys
  // 4g + r, dims j and 16 + j), the 7 mask words (bytes = this lane's 4 keys of query tile qt).  Rows past the split's
  // last key re-read that last key: finite, and masked out of the softmax below.
  struct Frag {
    dvis_f4 k0, k1;
    float v[4][NT];
    unsigned mw[QT];
  };
  auto load_tile = [&](int key0, Frag &f) {
    const unsigned ko = (unsigned)min(key0 + j, key_hi - 1) * k_row_bytes + (unsigned)(g * DQ * 4);
    f.k0 = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rk, ko, 0, 0));
    f.k1 = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rk, ko + 16u, 0, 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned vo = (unsigned)min(key0 + 4 * g + r, key_hi - 1) * v_row_bytes + (unsigned)(j * 4);
#pragma unroll
      for (int n = 0; n < NT; ++n)
        f.v[r][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, vo + 64u * n, 0, 0));
    }
    if (mask != nullptr) {   // uniform
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) f.mw[qt] = __builtin_amdgcn_raw_buffer_load_b32(rm, moff[qt] + (unsigned)key0, 0, 0);
    } else {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) f.mw[qt] = 0u;
    }
  };
  // S^T tile: rows = 16 keys, cols = the 16 queries of tile qt.  The accumulator starts from the lane's mask BIAS
  // (0, or a huge negative number for a dead key), so masking costs no select after the product.
  // Two independent half chains (dims 0-3 / 4-7 of the lane group), summed at the end: a dependent fp32 MFMA waits for
  // its predecessor's 8 passes, and one 8-long chain next to the two 4-long P V chains left the matrix pipe idle.
  auto score = [&](const Frag &f, int qt, dvis_f4 sc) -> dvis_f4 {
    dvis_f4 s2 = dvis_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.k0[c], qf[qt][c], sc, 0, 0, 0);
      s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.k1[c], qf[qt][4 + c], s2, 0, 0, 0);
    }
    return sc + s2;
  };
  // Dead keys (masked, or past the split's end) get the bias 0xFF000000 = -1.7e38: finite, absorbs any score, and
  // 2^(bias - m_ref) is exactly 0 because m_ref never goes below kNoRef = -1e30.  Bit 0 of the key's mask byte, sign-
  // extended and masked (v_bfe_i32 + v_and_b32), is that pattern — no compare / select and none of their wait states.
  auto bias_of = [&](unsigned mw) -> dvis_f4 {
    dvis_f4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      bv[r] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_sbfe((int)mw, 8 * r, 1) & 0xFF000000u);
    return bv;
  };
  auto mask_word = [&](const Frag &f, int qt, unsigned dead) -> unsigned {
    // (rows that ignore the mask — allowed_count == 0 — keep only the tile's own dead keys)
    return (((use_mask_bits >> qt) & 1u) ? f.mw[qt] : 0u) | dead;
  };
  // keys of a tile past the split's end, as a mask word (bytes = this lane's 4 keys)
  auto dead_word = [&](int key0) -> unsigned {
    unsigned dead = 0u;
    if (key0 + 16 > key_hi) {   // wave-uniform
#pragma unroll
      for (int r = 0; r < 4; ++r) dead |= key0 + 4 * g + r >= key_hi ? (0xffu << (8 * r)) : 0u;
    }
    return dead;
  };

  // Row state per query tile as SEPARATE variables: as arrays (float[7], or 2-vectors[7]) hipcc keeps the seven
  // references in one wide value and copies all of it around the raise branch of every tile (8-14 v_mov per query tile).
  // m_ref starts at a finite -1e30, not -inf: no special case for "no live key yet", and the first live score raises
  // the reference with alpha = 2^(-1e30 - m) = 0 on an all-zero state.
  constexpr float kNoRef = -1e30f;
  struct Row {
    float m, l;
  };
  Row st0{kNoRef, 0.f}, st1{kNoRef, 0.f}, st2{kNoRef, 0.f}, st3{kNoRef, 0.f}, st4{kNoRef, 0.f}, st5{kNoRef, 0.f},
      st6{kNoRef, 0.f};
  dvis_f4 o[QT][NT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int n = 0; n < NT; ++n) o[qt][n] = dvis_f4{0.f, 0.f, 0.f, 0.f};

  dvis_f4 s_next;
  // one query tile of one key tile (fragments `cur`); `next_score` issues the score MFMAs of the following (query, key)
  // tile, which the compiler places next to this tile's P V MFMAs
  auto tile_body = [&](auto qt_c, Row &st, const Frag &cur, auto next_score) {
    constexpr int qt = decltype(qt_c)::value;
    const dvis_f4 s = s_next;
    s_next = next_score();
    __builtin_amdgcn_sched_barrier(0);
    // lane (j, g) holds S[query q_base + 16 qt + j][key0 + 4g + r] (+ bias), r = 0..3
    const float tmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
    if (__any(tmax > st.m + kLazy)) {   // wave-uniform, rare after the first tiles: raise the reference
      float m_new = fmaxf(tmax, __shfl_xor(tmax, 16));
      m_new = fmaxf(m_new, __shfl_xor(m_new, 32));
      m_new = fmaxf(m_new, st.m);       // (a row with no live key so far keeps kNoRef: its tmax is the bias)
      const float alpha = ex2(st.m - m_new);
      // row (4g + r) of the accumulator belongs to query 4g + r of the tile, whose alpha lives in lane 4g + r
      const dvis_f4 av = dvis_f4{__shfl(alpha, 4 * g), __shfl(alpha, 4 * g + 1), __shfl(alpha, 4 * g + 2),
                                 __shfl(alpha, 4 * g + 3)};
#pragma unroll
      for (int n = 0; n < NT; ++n) o[qt][n] = o[qt][n] * av;
      st.m = m_new;
      st.l *= alpha;
    }
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = ex2(s[r] - st.m);
    st.l += (p[0] + p[1]) + (p[2] + p[3]);
    // ---- O += P V : A = P (already in A layout), B = V rows key0 + 4g + r
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) o[qt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], cur.v[r][n], o[qt][n], 0, 0, 0);
  };
  // all 7 query tiles of the key tile held in `cur`; `nxt` is the following key tile, requested by `fetch_next` after the
  // third query tile (it is first needed by the seventh)
  auto key_tile = [&](const Frag &cur, unsigned dead_cur, const Frag &nxt, unsigned &dead_nxt, auto fetch_next) {
#define DVIS_QT(N_, ST_)                                                                                  \
  tile_body(std::integral_constant<int, N_>{}, ST_, cur,                                                  \
            [&]() { return score(cur, N_ + 1, bias_of(mask_word(cur, N_ + 1, dead_cur))); })
    DVIS_QT(0, st0);
    DVIS_QT(1, st1);
    DVIS_QT(2, st2);
    fetch_next();
    DVIS_QT(3, st3);
    DVIS_QT(4, st4);
    DVIS_QT(5, st5);
#undef DVIS_QT
    tile_body(std::integral_constant<int, 6>{}, st6, cur,
              [&]() { return score(nxt, 0, bias_of(mask_word(nxt, 0, dead_nxt))); });   // the next key tile's first
  };

  // Two fragment sets, ping-pong over an unrolled pair of key tiles (no register copies).  WHERE the next tile is
  // requested matters: vmcnt counts loads in issue order, so the first use of a fragment of the CURRENT tile waits until
  // at most (loads issued after it) are outstanding — requested at the top of the tile, the 17 loads of the next tile
  // sit behind every such first use in the first query tile and `s_waitcnt vmcnt(15..10)` there waited for loads issued
  // a few hundred cycles earlier (timing-only ablation, tools/exp/abl: 125 of 653 us came back with the loads removed).
  // Requested after the third query tile, every fragment has had at least four query tiles (~1.5 us) to arrive.
  Frag fa, fb;
  load_tile(key_lo, fa);
  unsigned dead_a = dead_word(key_lo), dead_b = 0u;
  s_next = score(fa, 0, bias_of(mask_word(fa, 0, dead_a)));
#pragma unroll 1
  for (int key0 = key_lo; key0 < key_hi; key0 += 32) {
    const int kb1 = key0 + 16 < key_hi ? key0 + 16 : key0;   // (past the end: re-read the same tile, result unused)
    key_tile(fa, dead_a, fb, dead_b, [&]() { load_tile(kb1, fb); dead_b = dead_word(kb1); });
    if (key0 + 16 >= key_hi) break;                           // wave-uniform
    const int ka1 = key0 + 32 < key_hi ? key0 + 32 : key0 + 16;
    key_tile(fb, dead_b, fa, dead_a, [&]() { load_tile(ka1, fa); dead_a = dead_word(ka1); });
  }

  // ---- epilogue: l over the 4 lane groups; stats of query (4g + r) come from lane 4g + r
  const Row *rows[QT] = {&st0, &st1, &st2, &st3, &st4, &st5, &st6};
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float lq = rows[qt]->l, mq = rows[qt]->m;
    float l_tot = lq + __shfl_xor(lq, 16);
    l_tot += __shfl_xor(l_tot, 32);
    const float m_out = mq == kNoRef ? -INFINITY : mq;   // no live key in this split: weight 0 in the merge
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float lr = __shfl(l_tot, 4 * g + r), mr = __shfl(m_out, 4 * g + r);
      const int qq = q_base + qt * 16 + 4 * g + r;
      if (qq >= Lq) continue;
      if (nsplit == 1) {
        const float inv = lr > 0.f ? 1.f / lr : 0.f;
        float *orow = out + (size_t)bi * os.b + (size_t)hi * os.h + (size_t)qq * os.r;
#pragma unroll
        for (int n = 0; n < NT; ++n) orow[16 * n + j] = o[qt][n][r] * inv;
      } else {
        const size_t row = ((size_t)bh * nsplit + split) * Lq + qq;
#pragma unroll
        for (int n = 0; n < NT; ++n) ws_o[row * DH + 16 * n + j] = o[qt][n][r];
        if (j == 0) {
          ws_ml[row * 2] = mr;
          ws_ml[row * 2 + 1] = lr;
        }
      }
    }
  }
}

// Latency kernel for short key sequences with few (batch, head) pairs — the tracker's 100x100 attentions at batch 1,
// which are strictly sequential, so latency is what counts.  With the mapping above one (batch, head) sits on ONE CU and
// its 7 query tiles x 224 fp32 MFMAs (32 clk each) take ~6 us of pure MFMA issue; only 8 CUs are busy.  Here a workgroup is one 16-query tile of one head and its 4 waves (one per SIMD) split
// the key tiles; all keys are resident in LDS, so there is no online softmax: each wave does S^T for its tiles, one
// max / exp / sum pass, P V, and the four partial (m, l, O) are merged once through LDS.
template <int DH>
__global__ __launch_bounds__(256) void attn_short_kernel(
    const float *__restrict__ q, dvis_strides qs, const float *__restrict__ k, dvis_strides ks_, const float *__restrict__ v,
    dvis_strides vs, float *__restrict__ out, dvis_strides os, const uint8_t *__restrict__ mask,
    const int *__restrict__ allowed, int heads, int Lq, int Lk, float scale) {
  constexpr int DQ = DH / 4, NT = DH / 16, LS = DH + 4, NKT = 2;   // Lk <= 128: at most 2 key tiles per wave
  // dynamic LDS sized to the padded key count (a 30-key refiner call should not pin 69 KB and halve the occupancy):
  // [K rows | V rows | (m, l) merge]; the K region is at least 4*16*DH floats because it is reused for the O merge.
  extern __shared__ float short_lds[];
  const int nrows = (Lk + 15) / 16 * 16;
  const int kregion = max(nrows * LS, 64 * DH);
  float *k_lds = short_lds;
  float *v_lds = short_lds + kregion;
  float *ml_lds = v_lds + nrows * LS;
  float *o_lds = k_lds;                  // K is dead once every wave has its S tiles

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y;
  const int bi = bh / heads, hi = bh - bi * heads;
  const int q0 = blockIdx.x * 16;
  const int myq = q0 + j;
  const bool q_ok = myq < Lq;

  float qf[DQ];
  float4 qraw[DQ / 4];
  {
    const float *qrow = q + (size_t)bi * qs.b + (size_t)hi * qs.h + (size_t)(q_ok ? myq : 0) * qs.r + g * DQ;
#pragma unroll
    for (int c = 0; c < DQ / 4; ++c) qraw[c] = *reinterpret_cast<const float4 *>(qrow + 4 * c);
  }
  const float *kb = k + (size_t)bi * ks_.b + (size_t)hi * ks_.h;
  const float *vb = v + (size_t)bi * vs.b + (size_t)hi * vs.h;
  // K / V -> LDS.  A link of a launch-bound chain can afford ONE memory round trip: every 16-byte piece of K and V this
  // thread stages is requested before the first one is written to LDS (a load -> store loop serialises into one round trip
  // per iteration: 7 of them at 100 keys).
  // Through buffer descriptors over this head's Lk rows (rows past Lk read zeros in hardware): no branch around a load —
  // with conditional loads hipcc merges the "zero" and "loaded" values through register copies that wait for the load.
  constexpr int NIT = 128 * (DH / 4) / 256;   // Lk <= 128
  dvis_f4 kr[NIT], vr[NIT];
  {
    const __amdgpu_buffer_rsrc_t rk = dvis_make_rsrc_uniform(kb, (unsigned)(((size_t)(Lk - 1) * ks_.r + DH) * 4));
    const __amdgpu_buffer_rsrc_t rv = dvis_make_rsrc_uniform(vb, (unsigned)(((size_t)(Lk - 1) * vs.r + DH) * 4));
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = tid + 256 * it;
      const int row = e / (DH / 4), c4 = e - row * (DH / 4);
      const bool in = row < Lk;
      kr[it] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(
                                               rk, in ? (unsigned)(((size_t)row * ks_.r + 4 * c4) * 4) : 0x80000000u, 0, 0));
      vr[it] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(
                                               rv, in ? (unsigned)(((size_t)row * vs.r + 4 * c4) * 4) : 0x80000000u, 0, 0));
    }
  }
  // (scheduling fence: left alone hipcc hoists the q scaling between the K / V requests and waits for the q rows there —
  // the second half of the requests then leaves one round trip late)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < DQ / 4; ++c) {
    const float4 t = qraw[c];
    qf[4 * c] = q_ok ? t.x * (scale * kLog2e) : 0.f;
    qf[4 * c + 1] = q_ok ? t.y * (scale * kLog2e) : 0.f;
    qf[4 * c + 2] = q_ok ? t.z * (scale * kLog2e) : 0.f;
    qf[4 * c + 3] = q_ok ? t.w * (scale * kLog2e) : 0.f;
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = tid + 256 * it;
    const int row = e / (DH / 4), c4 = e - row * (DH / 4);
    if (row < nrows) {
      *reinterpret_cast<dvis_f4 *>(&k_lds[row * LS + 4 * c4]) = kr[it];
      *reinterpret_cast<dvis_f4 *>(&v_lds[row * LS + 4 * c4]) = vr[it];
    }
  }
  const bool use_mask = mask != nullptr && q_ok && (allowed == nullptr || allowed[(size_t)bi * Lq + myq] != 0);
  const uint8_t *mrow = mask ? mask + ((size_t)bi * Lq + (q_ok ? myq : 0)) * Lk : nullptr;
  __syncthreads();

  const int ntiles = nrows / 16;
  // this wave's key tiles: wv, wv + 4
  dvis_f4 sa[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) sa[t] = dvis_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < DQ / 4; ++c) {
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const int kt = wv + 4 * t;
      if (kt < ntiles) {   // wave-uniform
        const float4 kk = *reinterpret_cast<const float4 *>(&k_lds[(kt * 16 + j) * LS + g * DQ + 4 * c]);
        sa[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.x, qf[4 * c], sa[t], 0, 0, 0);
        sa[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.y, qf[4 * c + 1], sa[t], 0, 0, 0);
        sa[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.z, qf[4 * c + 2], sa[t], 0, 0, 0);
        sa[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kk.w, qf[4 * c + 3], sa[t], 0, 0, 0);
      }
    }
  }
  // lane (j, g): sa[t][r] = S[query myq][key 16 * (wv + 4 t) + 4 g + r]
  float tmax = -INFINITY;
  bool dead[NKT][4];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int kbase = (wv + 4 * t) * 16 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bool d = kbase + r >= Lk;
      if (use_mask && !d) d = mrow[kbase + r] != 0;
      dead[t][r] = d;
      if (!d) tmax = fmaxf(tmax, sa[t][r]);
    }
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
  float lsum = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pr = dead[t][r] ? 0.f : ex2(sa[t][r] - tmax);
      sa[t][r] = pr;
      lsum += pr;
    }
  lsum += __shfl_xor(lsum, 16);
  lsum += __shfl_xor(lsum, 32);
  dvis_f4 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = dvis_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int kt = wv + 4 * t;
    if (kt < ntiles) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float *vrow = &v_lds[(kt * 16 + 4 * g + r) * LS + j];
#pragma unroll
        for (int n = 0; n < NT; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[t][r], vrow[16 * n], o[n], 0, 0, 0);
      }
    }
  }
  // ---- merge the 4 waves' partials.  (m, l) of query jq live in lanes with (lane & 15) == jq.
  if (g == 0) {
    ml_lds[(wv * 16 + j) * 2] = tmax;
    ml_lds[(wv * 16 + j) * 2 + 1] = lsum;
  }
  __syncthreads();           // every wave is past its K reads: k_lds may be overwritten
  // accumulator rows are queries 4 g + r
  float inv[4], wgt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int jq = 4 * g + r;
    float M = -INFINITY;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) M = fmaxf(M, ml_lds[(w2 * 16 + jq) * 2]);
    float L = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const float mw = ml_lds[(w2 * 16 + jq) * 2];
      L += (mw == -INFINITY) ? 0.f : ml_lds[(w2 * 16 + jq) * 2 + 1] * ex2(mw - M);
    }
    const float mine = ml_lds[(wv * 16 + jq) * 2];
    wgt[r] = (mine == -INFINITY) ? 0.f : ex2(mine - M);
    inv[r] = L > 0.f ? 1.f / L : 0.f;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) o_lds[(wv * 16 + 4 * g + r) * DH + 16 * n + j] = o[n][r] * wgt[r];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + 4 * g + r;
      if (qq < Lq) {
        float *orow = out + (size_t)bi * os.b + (size_t)hi * os.h + (size_t)qq * os.r;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float acc = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) acc += o_lds[(w2 * 16 + 4 * g + r) * DH + 16 * n + j];
          orow[16 * n + j] = acc * inv[r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Long self-attention at head dim 64 on the F16 matrix cores with split fp32 operands (round 5): the DINOv2 / ViT-Adapter blocks
// of BASELINE config #5 (3681 tokens x 16 heads x 24 blocks: 40 TFLOP per 30-frame clip, 38 % of that clip on the fp32 kernel
// above at 0.71 of the fp32 matrix peak).  Same partition as attn_fwd_kernel — 8 waves x 16 queries per workgroup, K / V staged
// through LDS for all of them, online softmax in base 2 with the score tile of key tile kt + 1 issued before the softmax of tile
// kt — with the arithmetic of csrc/gemm_x3.hip: every fp32 operand v is carried as hi = rn16(v 2^e), lo = rn16(v 2^e - hi) and
// every product as lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x16_f16 (fp32 accumulation): 12 matrix instructions of 8 cycles per
// 16-key tile and product where the fp32 form issues 16 of 32 cycles.
//   S^T tile (16 keys x 16 queries):  A = K[key j][16 c + 4 g ..+3], B = Q[query j][16 c + 4 g ..+3], c = 0..3
//     -> lane (j, g) holds S[query j][key0 + 4 g + r] = the A layout of the next product: P[query j][keys 4 g ..+3]
//   O tile (16 queries x 16 dims):    A = P (from registers), B = V[key0 + 4 g ..+3][16 n + j]: four CONSECUTIVE KEYS of one
//     dim per lane, so V is staged TRANSPOSED ([dim][key], split once per stage by the threads that load it).
// Scales: Q (already times scale log2 e), K, V by 2^4 (|v| < 4094; elements down to 2^-7 keep 22 bits), P (in [0, 1]) by 2^10;
// the score tile is scaled back by 2^-8, the output by 2^-14 inside its 1 / l.  No mask, no key split (callers: nsplit == 1).
typedef _Float16 ah4 __attribute__((ext_vector_type(4)));
constexpr int kX3KT = 64;             // keys per stage
constexpr int kX3Row = 144;           // halves per LDS row of K ([hi 64 | lo 64] + 16: ds_read_b128 is served in the 16-lane groups {0-3, 12-15,
                                      // 20-27}, ... — rows 72 dwords apart put each group's 16 fragments (8 of lane group g, 8 of g + 1)
                                      // on 16 different 4-bank slots; at 68 dwords one pair collides)
constexpr int kX3RowV = 144;          // ... of V, read through ds_read_b64_tr_b16: eight key rows x 32 B tile the 64 banks at 72 dwords per row
typedef __fp16 dvis_fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
// The LDS transpose read of gfx950: within a group of 16 lanes, lane i names 8 bytes (row i / 4, 16-bit columns 4 (i % 4) ..+3 of a
// [4][16] block) and receives column i of the block's four rows.  V stays [key][dim] in LDS — the layout it is staged in with
// 8-byte writes — and still arrives as the k-contiguous B operand of P V (four keys of one dim per lane).
__device__ __forceinline__ ah4 lds_read_tr16(const _Float16 *p) {
  return __builtin_bit_cast(ah4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) dvis_fp4 *)(p)));
}

typedef _Float16 ah8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ ah8 x3_cat(ah4 a, ah4 b) { return ah8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
__device__ __forceinline__ void x3_split4(float a, float b, float c, float d, ah4 &hi, ah4 &lo) {
  const _Float16 h0 = (_Float16)a, h1 = (_Float16)b, h2 = (_Float16)c, h3 = (_Float16)d;
  hi = ah4{h0, h1, h2, h3};
  lo = ah4{(_Float16)(a - (float)h0), (_Float16)(b - (float)h1), (_Float16)(c - (float)h2), (_Float16)(d - (float)h3)};
}

// Pass 1: Q, K, V -> two-term f16 images in the workspace, [matrix][batch-head][token][hi 64 | lo 64] (256 B per token and matrix:
// what one fp32 row takes).  Q carries scale * log2(e) * 2^4, K and V 2^4.  The split then costs one pass over the operands
// instead of one per workgroup that stages them (29 query blocks stream the same K / V at 3681 tokens) — in the main kernel the
// splitting arithmetic was half of all vector instructions.
__global__ __launch_bounds__(256) void attn_x3_pack_kernel(const float *__restrict__ q, dvis_strides qs, const float *__restrict__ k,
                                                           dvis_strides ks_, const float *__restrict__ v, dvis_strides vs,
                                                           _Float16 *__restrict__ ws, int heads, int Lq, int Lk, float qscale) {
  const int which = blockIdx.z, bh = blockIdx.y, bi = bh / heads, hi_ = bh - bi * heads;
  const int L = which == 0 ? Lq : Lk;
  const int row = blockIdx.x * 32 + (threadIdx.x >> 3), c8 = threadIdx.x & 7;
  if (row >= L) return;
  const float *src = which == 0 ? q : which == 1 ? k : v;
  const dvis_strides st = which == 0 ? qs : which == 1 ? ks_ : vs;
  const float f = which == 0 ? qscale : 16.f;
  const float *p = src + (size_t)bi * st.b + (size_t)hi_ * st.h + (size_t)row * st.r + 8 * c8;
  const float4 x = *reinterpret_cast<const float4 *>(p), y = *reinterpret_cast<const float4 *>(p + 4);
  ah4 h0, l0, h1, l1;
  x3_split4(x.x * f, x.y * f, x.z * f, x.w * f, h0, l0);
  x3_split4(y.x * f, y.y * f, y.z * f, y.w * f, h1, l1);
  const size_t BH = gridDim.y;
  _Float16 *dst = ws + (which == 0 ? (size_t)0 : which == 1 ? BH * Lq * 128 : BH * ((size_t)Lq + Lk) * 128) + ((size_t)bh * L + row) * 128 +
                  8 * c8;
  *reinterpret_cast<ah8 *>(dst) = x3_cat(h0, h1);
  *reinterpret_cast<ah8 *>(dst + 64) = x3_cat(l0, l1);
}

// Pass 2.  QT query tiles of 16 per wave (a workgroup covers 128 QT queries): every K / V fragment read from LDS serves QT matrix
// instructions, and a staged K / V tile QT times the queries.
// IMG: the output as the ROW IMAGE of the GEMM that consumes it (csrc/gemm_x3_tile.hip: [row tile of 128][k-tile of 16][hi, lo][chunk]
// [128 rows][8 halves], rows = (batch entry, token), k = (head, dim)) x img_scale: a lane holds four dims of a query, two neighbouring
// lanes one 16-byte fragment (k order 2 of x3_tile_k) — no fp32 attention output, no split in the projection.
template <int QT, bool IMG = false>
__global__ __launch_bounds__(512) void attn_x3_kernel(const _Float16 *__restrict__ ws, float *__restrict__ out, dvis_strides os, int BH,
                                                      int heads, int Lq, int Lk, int *__restrict__ guard_flag, int guard_tag,
                                                      char *__restrict__ img = nullptr, float img_scale = 1.f) {
  constexpr int KT = kX3KT, NT = KT / 16, RS = kX3Row, RV = kX3RowV;
  __shared__ __attribute__((aligned(16))) _Float16 k_lds[KT * RS], v_lds[KT * RV];      // [key][hi 64 | lo 64 | pad]; V is read transposed
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  // Workgroup -> (batch-head, query block), XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, each with its own
  // L2; all query blocks of one (batch, head) stream the same K / V (1.9 MB at 3681 keys), so they are given to ONE XCD and to
  // consecutive slots there — K / V then come from HBM once and from that L2 for the other blocks.  (With the plain (bh, block)
  // grid every resident workgroup streamed a different head's K / V: 8.7 GB of L2 misses per ViT-L block at 10 frames.)
  constexpr int QB = 128 * QT;
  const int nqb = (Lq + QB - 1) / QB;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / nqb) * 8 + xcd, qb = slot - (slot / nqb) * nqb;
  if (bh >= BH) return;
  const int bi = bh / heads, hi_ = bh - bi * heads;
  const int q0 = qb * QB + wv * (16 * QT);
  const bool wave_on = q0 < Lq;
  constexpr float kOp = 16.f, kLogP = 10.f;        // operands x 2^4; probabilities x 2^10 (folded into the exponent)
  const _Float16 *wq = ws + (size_t)bh * Lq * 128;
  const _Float16 *wk = ws + (size_t)BH * Lq * 128 + (size_t)bh * Lk * 128;
  const _Float16 *wvv = ws + (size_t)BH * ((size_t)Lq + Lk) * 128 + (size_t)bh * Lk * 128;

  // ---- B operands of S^T (v_mfma_f32_16x16x32_f16: lane (j, g) holds k = 8 g ..+7): Q[q0 + 16 t + j][32 c + 8 g ..+7]
  ah8 qh[QT][2], ql[QT][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int myq = min(q0 + 16 * t + j, Lq - 1);          // (rows past Lq repeat the last query: computed, never stored)
    const _Float16 *qrow = wq + (size_t)myq * 128 + 8 * g;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      qh[t][c] = *reinterpret_cast<const ah8 *>(qrow + 32 * c);
      ql[t][c] = *reinterpret_cast<const ah8 *>(qrow + 64 + 32 * c);
    }
  }
  dvis_f4 o[QT][4];
  float m_run[QT], l_part[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m_run[t] = -INFINITY, l_part[t] = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) o[t][n] = dvis_f4{0.f, 0.f, 0.f, 0.f};
  }

  // stage = 64 keys x 256 B per matrix = 1024 16-byte pieces: two per thread and matrix, prefetched one stage ahead
  ah8 pk[2], pv[2];
  auto prefetch = [&](int ks) {          // rows past the last key re-read it: their scores are set to -inf below
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 512 * i, key = min(ks + (id >> 4), Lk - 1);
      pk[i] = *reinterpret_cast<const ah8 *>(wk + (size_t)key * 128 + 8 * (id & 15));
      pv[i] = *reinterpret_cast<const ah8 *>(wvv + (size_t)key * 128 + 8 * (id & 15));
    }
  };
  prefetch(0);
  for (int ks = 0; ks < Lk; ks += KT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + 512 * i;
      *reinterpret_cast<ah8 *>(&k_lds[(id >> 4) * RS + 8 * (id & 15)]) = pk[i];
      *reinterpret_cast<ah8 *>(&v_lds[(id >> 4) * RV + 8 * (id & 15)]) = pv[i];
    }
    __syncthreads();
    if (ks + KT < Lk) prefetch(ks + KT);
    if (!wave_on) continue;
    // ---- the stage's four S^T tiles per query tile (rows = 16 keys, cols = 16 queries; scaled by 2^8), ONE running-max update per
    //      64 keys: the cross-lane reduction and the exp chain are paid once per stage, not per tile
    dvis_f4 s[QT][NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
      for (int t = 0; t < QT; ++t) s[t][kt] = dvis_f4{0.f, 0.f, 0.f, 0.f};
      const _Float16 *kr = &k_lds[(kt * 16 + j) * RS + 8 * g];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const ah8 a_h = *reinterpret_cast<const ah8 *>(kr + 32 * c), a_l = *reinterpret_cast<const ah8 *>(kr + 64 + 32 * c);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, qh[t][c], s[t][kt], 0, 0, 0);
          s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, ql[t][c], s[t][kt], 0, 0, 0);
          s[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, qh[t][c], s[t][kt], 0, 0, 0);
        }
      }
    }
    ah8 a_h[QT][NT / 2], a_l[QT][NT / 2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (ks + KT > Lk) {                             // the ragged last stage (uniform)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[t][kt][r] = ks + kt * 16 + 4 * g + r >= Lk ? -INFINITY : s[t][kt][r];
      }
      float tmax = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) tmax = fmaxf(tmax, fmaxf(fmaxf(s[t][kt][0], s[t][kt][1]), fmaxf(s[t][kt][2], s[t][kt][3])));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run[t], tmax * (1.f / 256.f));      // every stage holds at least one key: finite
      const bool grew = !__all(m_new == m_run[t]);
      float alpha = 1.f;
      if (grew) alpha = ex2(m_run[t] - m_new);        // first stage: ex2(-inf) = 0 on zero accumulators
      const float shift = kLogP - m_new;
      ah4 ph[NT], pl[NT];
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] = ex2(__builtin_fmaf(s[t][kt][r], 1.f / 256.f, shift));
          psum += p[r];
        }
        x3_split4(p[0], p[1], p[2], p[3], ph[kt], pl[kt]);
      }
      l_part[t] = l_part[t] * alpha + psum;
      m_run[t] = m_new;
      if (grew) {
        const dvis_f4 av = dvis_f4{__shfl(alpha, 4 * g), __shfl(alpha, 4 * g + 1), __shfl(alpha, 4 * g + 2), __shfl(alpha, 4 * g + 3)};
#pragma unroll
        for (int n = 0; n < 4; ++n) o[t][n] = o[t][n] * av;
      }
#pragma unroll
      for (int pb = 0; pb < NT / 2; ++pb) a_h[t][pb] = x3_cat(ph[2 * pb], ph[2 * pb + 1]), a_l[t][pb] = x3_cat(pl[2 * pb], pl[2 * pb + 1]);
    }
    // ---- O += P V, 32 keys per matrix instruction.  The k index of the instruction is a free permutation as long as both operands
    //      use it: slot 8 g + e is key 4 g + e of tile 2 pb (e < 4) or of tile 2 pb + 1 (e >= 4) — exactly the probabilities this
    //      lane holds (A = P[query j][slot]); B = V[key(slot)][dim 16 n + j]: two transpose reads, one per tile.
#pragma unroll
    for (int pb = 0; pb < NT / 2; ++pb) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int voff = (pb * 32 + 4 * g + (j >> 2)) * RV + 16 * n + 4 * (j & 3);      // this lane's 8 bytes of the group's block
        const ah8 b_h = x3_cat(lds_read_tr16(&v_lds[voff]), lds_read_tr16(&v_lds[voff + 16 * RV]));
        const ah8 b_l = x3_cat(lds_read_tr16(&v_lds[voff + 64]), lds_read_tr16(&v_lds[voff + 64 + 16 * RV]));
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          o[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l[t][pb], b_h, o[t][n], 0, 0, 0);
          o[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[t][pb], b_l, o[t][n], 0, 0, 0);
          o[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h[t][pb], b_h, o[t][n], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_on) return;
  float chk = 0.f;
  if constexpr (IMG) {
    const int KT = heads * 4, jp = j >> 1;
    char *base = img + (size_t)(4 * hi_ + (jp >> 1)) * 8192 + (jp & 1) * 2048 + (j & 1) * 8;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      float l_tot = l_part[t] + __shfl_xor(l_part[t], 16);
      l_tot += __shfl_xor(l_tot, 32);
      ah4 vh[4], vl[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float lr = __shfl(l_tot, 4 * g + r);
        const float inv = lr > 0.f ? (1.f / kOp) / lr : 0.f;
        float v[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          v[n] = o[t][n][r] * inv;
          chk = __builtin_fmaf(v[n], 0.f, chk);
          v[n] *= img_scale;
        }
        x3_split4(v[0], v[1], v[2], v[3], vh[r], vl[r]);
      }
      // (nothing may be scheduled into the stores and a pause follows them: a vector store reads its data registers after it has
      // issued — x3_common.h, store_fragments4)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = q0 + 16 * t + 4 * g + r;
        if (qq < Lq) {
          const size_t m = (size_t)bi * Lq + qq;
          char *dst = base + (m >> 7) * KT * 8192 + (m & 127) * 16;
          *reinterpret_cast<ah4 *>(dst) = vh[r];
          *reinterpret_cast<ah4 *>(dst + 4096) = vl[r];
        }
      }
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    if (guard_flag != nullptr && chk != chk) atomicCAS(guard_flag, 0, guard_tag);
    return;
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l_tot = l_part[t] + __shfl_xor(l_part[t], 16);
    l_tot += __shfl_xor(l_tot, 32);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float lr = __shfl(l_tot, 4 * g + r);
      const int qq = q0 + 16 * t + 4 * g + r;
      const float inv = lr > 0.f ? (1.f / kOp) / lr : 0.f;        // l carries the probabilities' 2^10
      if (qq < Lq) {
        float *orow = out + (size_t)bi * os.b + (size_t)hi_ * os.h + (size_t)qq * os.r;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const float val = o[t][n][r] * inv;
          chk = __builtin_fmaf(val, 0.f, chk);
          orow[16 * n + j] = val;
        }
      }
    }
  }
  // range guard (x3_common.h): an operand beyond the f16 range (|v| >= 4094) leaves non-finite outputs
  if (guard_flag != nullptr && chk != chk) atomicCAS(guard_flag, 0, guard_tag);
}

// Merge the per-split partials: O = sum_s O_s e^{m_s - M} / sum_s l_s e^{m_s - M}.
__global__ __launch_bounds__(256) void attn_combine_kernel(const float *__restrict__ ws_o, const float *__restrict__ ws_ml,
                                                           int nsplit, int Lq, int DH, int heads, size_t total,
                                                           float *__restrict__ out, dvis_strides os) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int d = (int)(idx % DH);
  const size_t t = idx / DH;
  const int qq = (int)(t % Lq);
  const size_t bh = t / Lq;
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, ws_ml[((bh * nsplit + s) * Lq + qq) * 2]);
  float num = 0.f, den = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const size_t row = (bh * nsplit + s) * Lq + qq;
    const float ms = ws_ml[row * 2];
    const float wgt = (ms == -INFINITY) ? 0.f : ex2(ms - M);
    num += ws_o[row * DH + d] * wgt;
    den += ws_ml[row * 2 + 1] * wgt;
  }
  const size_t bi = bh / heads, hi = bh - bi * heads;
  out[bi * os.b + hi * os.h + (size_t)qq * os.r + d] = den > 0.f ? num / den : 0.f;
}

}  // namespace

DVIS_EXPORT int64_t dvis_attention_ws_bytes_k(int BH, int Lq, int Lk, int d, int kernel) {
  if (kernel != 2) return dvis_attention_ws_bytes(BH, Lq, Lk, d);
  if (BH <= 0 || Lq <= 0 || Lk <= 0 || d != 64) return 0;
  return (int64_t)BH * ((int64_t)Lq + 2 * (int64_t)Lk) * 128 * (int64_t)sizeof(_Float16);       // the two-term f16 images of Q, K, V
}

DVIS_EXPORT int64_t dvis_attention_ws_bytes(int BH, int Lq, int Lk, int d) {
  if (BH <= 0 || Lq <= 0 || Lk <= 0 || d <= 0) return 0;
  const SplitPlan p = plan_split(BH, Lq, Lk);
  int ns = p.nsplit;
  if (keysplit_applies(Lq, Lk, d, false, 0, 0)) ns = std::max(ns, plan_keysplit(BH, Lq, Lk).nsplit);   // whichever runs
  if (ns == 1) return 0;
  return (int64_t)BH * ns * Lq * (d + 2) * (int64_t)sizeof(float);
}

static int attention_launch(const float *q, const int64_t *q_strides, const float *k, const int64_t *k_strides,
                            const float *v, const int64_t *v_strides, float *out, const int64_t *o_strides,
                            const uint8_t *mask, const int32_t *allowed_count, int B, int heads, int Lq, int Lk, int d,
                            float scale, void *ws, void *stream, int kernel) {
  DVIS_REQUIRE(B >= 0 && heads > 0 && Lq >= 0 && Lk > 0, "attention: bad sizes");
  if (B == 0 || Lq == 0) return DVIS_OK;
  DVIS_REQUIRE(q && k && v && out && q_strides && k_strides && v_strides && o_strides, "attention: null pointer");
  DVIS_REQUIRE(d == 32 || d == 64, "attention: head dim must be 32 or 64 (got %d)", d);
  const int BH = B * heads;
  DVIS_REQUIRE(BH <= 65535, "attention: batch*heads must be <= 65535");
  const dvis_strides qs{q_strides[0], q_strides[1], q_strides[2]}, ks{k_strides[0], k_strides[1], k_strides[2]};
  const dvis_strides vs{v_strides[0], v_strides[1], v_strides[2]}, os{o_strides[0], o_strides[1], o_strides[2]};
  const uintptr_t al = (uintptr_t)q | (uintptr_t)k | (uintptr_t)v;
  DVIS_REQUIRE((al & 15) == 0 && ((qs.b | qs.h | qs.r | ks.b | ks.h | ks.r | vs.b | vs.h | vs.r) & 3) == 0,
               "attention: q/k/v must be 16-byte aligned with strides that are multiples of 4 floats");
  DVIS_REQUIRE(mask == nullptr || ((uintptr_t)mask & 3) == 0, "attention: mask must be 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  DVIS_REQUIRE(kernel == 0 || kernel == 2 || (kernel == 1 && Lk <= 128), "attention: kernel 1 (short keys) needs Lk <= 128 (Lk=%d)", Lk);
  if (kernel == 2) {      // split-f16 long self-attention (ViT blocks): d = 64, no mask, enough query chunks that keys need no split
    DVIS_REQUIRE(d == 64 && mask == nullptr && Lk >= 128, "attention: kernel 2 (split-f16) serves d = 64 without a mask, Lk >= 128");
    DVIS_REQUIRE(ws != nullptr, "attention: kernel 2 needs its workspace (dvis_attention_ws_bytes_k)");
    const X3Guard gd = dvis_x3_guard();
    _Float16 *wsh = (_Float16 *)ws;
    hipLaunchKernelGGL(attn_x3_pack_kernel, dim3((std::max(Lq, Lk) + 31) / 32, BH, 3), dim3(256), 0, st, q, qs, k, ks, v, vs, wsh, heads, Lq,
                       Lk, scale * kLog2e * 16.f);
    if (const int rc = dvis_check_launch("attn_x3_pack_kernel")) return rc;
    static const int qt = []() { const char *e = getenv("DVIS_ATTN_X3_QT"); return e ? atoi(e) : 1; }();     // (development)
    if (qt == 2)
      hipLaunchKernelGGL(attn_x3_kernel<2>, dim3(((BH + 7) / 8) * 8 * ((Lq + 255) / 256)), dim3(512), 0, st, wsh, out, os, BH, heads, Lq, Lk,
                         gd.flag, gd.tag);
    else
      hipLaunchKernelGGL(attn_x3_kernel<1>, dim3(((BH + 7) / 8) * 8 * ((Lq + 127) / 128)), dim3(512), 0, st, wsh, out, os, BH, heads, Lq, Lk,
                         gd.flag, gd.tag);
    return dvis_check_launch("attn_x3_kernel");
  }
  kernel = kernel == 1 ? 1 : 0;
  if (kernel == 0 && keysplit_applies(Lq, Lk, d, mask != nullptr, ks.r, vs.r)) {
    const KeySplitPlan kp = plan_keysplit(BH, Lq, Lk);
    DVIS_REQUIRE(kp.nsplit == 1 || ws, "attention: workspace required (dvis_attention_ws_bytes)");
    float *kws_o = (float *)ws;
    float *kws_ml = kws_o ? kws_o + (size_t)BH * kp.nsplit * Lq * d : nullptr;
    const long long units = (long long)BH * kp.qchunks * kp.nsplit;
    hipLaunchKernelGGL(attn_keysplit_kernel, dim3((unsigned)((units + 7) / 8)), dim3(512), 0, st, q, qs, k, ks, v, vs, out, os,
                       mask, allowed_count, heads, Lq, Lk, scale, kp.nsplit, kp.keys_per_split, kp.qchunks, (int)units, kws_o,
                       kws_ml);
    int rc = dvis_check_launch("attn_keysplit_kernel");
    if (rc != DVIS_OK || kp.nsplit == 1) return rc;
    const size_t total = (size_t)BH * Lq * d;
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, kws_o, kws_ml, kp.nsplit,
                       Lq, d, heads, total, out, os);
    return dvis_check_launch("attn_combine_kernel");
  }
  const SplitPlan p = plan_split(BH, Lq, Lk);
  DVIS_REQUIRE(p.nsplit == 1 || ws, "attention: workspace required (dvis_attention_ws_bytes)");
  float *ws_o = (float *)ws;
  float *ws_ml = ws_o ? ws_o + (size_t)BH * p.nsplit * Lq * d : nullptr;
  const dim3 grid(p.nsplit, BH, p.qchunks), block(512);
// Lk <= 128 (decoder / tracker / refiner self- and cross-attention over queries or frames): one workgroup per (batch, head,
  // 16-query tile) with the keys split over 4 waves (8.0 vs 14.8 us for the tracker's batch-1 call) — for EVERY batch size: the
  // kernels differ in their summation order, and choosing between them by the number of (batch, head) pairs (rounds 1 - 4: <= 384
  // workgroups; at B = 30 the other kernel is 6 us faster per call) made a frame's bits depend on its batch mates.
  // DVIS_ATTN_SHORT_MAX (development): the old rule.
  static const long long short_max = []() { const char *e = getenv("DVIS_ATTN_SHORT_MAX"); return e ? atoll(e) : -1ll; }();
  if (kernel == 1 || (Lk <= 128 && p.nsplit == 1 && (short_max < 0 || (long long)BH * ((Lq + 15) / 16) <= short_max))) {
    const dim3 sgrid((Lq + 15) / 16, BH);
    const int nrows = (Lk + 15) / 16 * 16, ls = d + 4;
    const size_t lds = sizeof(float) * ((size_t)std::max(nrows * ls, 64 * d) + (size_t)nrows * ls + 128);
    static DvisLdsOptIn opted32, opted64;       // (d = 64 with 113 .. 128 keys needs 70 KB)
    if (const int rc = d == 32 ? dvis_lds_opt_in((const void *)attn_short_kernel<32>, lds, &opted32, "attn_short_kernel")
                               : dvis_lds_opt_in((const void *)attn_short_kernel<64>, lds, &opted64, "attn_short_kernel"))
      return rc;
    if (d == 32)
      hipLaunchKernelGGL((attn_short_kernel<32>), sgrid, dim3(256), lds, st, q, qs, k, ks, v, vs, out, os, mask, allowed_count,
                         heads, Lq, Lk, scale);
    else
      hipLaunchKernelGGL((attn_short_kernel<64>), sgrid, dim3(256), lds, st, q, qs, k, ks, v, vs, out, os, mask, allowed_count,
                         heads, Lq, Lk, scale);
    return dvis_check_launch("attn_short_kernel");
  }
  #define DVIS_ATTN(DH_, SHORT_)                                                                                     \
  hipLaunchKernelGGL((attn_fwd_kernel<DH_, SHORT_>), grid, block, 0, st, q, qs, k, ks, v, vs, out, os, mask, allowed_count, \
                     heads, Lq, Lk, scale, p.nsplit, p.keys_per_split, ws_o, ws_ml)
  const bool shrt = Lk <= 128 && p.nsplit == 1;
  if (d == 32 && shrt) DVIS_ATTN(32, true);
  else if (d == 32) DVIS_ATTN(32, false);
  else if (shrt) DVIS_ATTN(64, true);
  else DVIS_ATTN(64, false);
#undef DVIS_ATTN
  int rc = dvis_check_launch("attn_fwd_kernel");
  if (rc != DVIS_OK || p.nsplit == 1) return rc;
  const size_t total = (size_t)BH * Lq * d;
  hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws_o, ws_ml, p.nsplit,
                     Lq, d, heads, total, out, os);
  return dvis_check_launch("attn_combine_kernel");
}

DVIS_EXPORT int dvis_attention_forward(const float *q, const int64_t *q_strides, const float *k, const int64_t *k_strides,
                                       const float *v, const int64_t *v_strides, float *out, const int64_t *o_strides,
                                       const uint8_t *mask, const int32_t *allowed_count, int B, int heads, int Lq,
                                       int Lk, int d, float scale, void *ws, void *stream) {
  return attention_launch(q, q_strides, k, k_strides, v, v_strides, out, o_strides, mask, allowed_count, B, heads, Lq, Lk, d,
                          scale, ws, stream, 0);
}

// Kernel 2's second launch alone: Q, K, V are already in `ws` as two-term f16 images (written by dvis_x3_tile_linear_qkv, the qkv
// projection's epilogue) — self-attention of B x heads (batch, head) pairs over L tokens at d = 64, no mask.
DVIS_EXPORT int dvis_attention_x3_packed(const void *ws, float *out, const int64_t *o_strides, int B, int heads, int L, void *stream) {
  DVIS_REQUIRE(ws && out && o_strides && B >= 0 && heads > 0 && L >= 128, "attention_x3_packed: bad arguments (L >= 128)");
  DVIS_REQUIRE((uintptr_t)ws % 16 == 0, "attention_x3_packed: workspace must be 16-byte aligned");
  if (B == 0) return DVIS_OK;
  const int BH = B * heads;
  const dvis_strides os{o_strides[0], o_strides[1], o_strides[2]};
  const X3Guard gd = dvis_x3_guard();
  hipLaunchKernelGGL(attn_x3_kernel<1>, dim3(((BH + 7) / 8) * 8 * ((L + 127) / 128)), dim3(512), 0, (hipStream_t)stream, (const _Float16 *)ws, out,
                     os, BH, heads, L, L, gd.flag, gd.tag);
  return dvis_check_launch("attn_x3_kernel");
}

// Rows [M, end of M's row tile) of a row image as zeros (every k-tile): what a producer that only writes whole results leaves open.
__global__ __launch_bounds__(256) void rows_image_tail_kernel(char *__restrict__ img, size_t M, int KT) {
  const int kt = blockIdx.x, part = threadIdx.x >> 7, row = threadIdx.x & 127;
  if (row < (int)(M & 127)) return;
  char *p = img + ((M >> 7) * KT + kt) * 8192 + row * 16;
  const dvis_f4 z = {0.f, 0.f, 0.f, 0.f};
  *reinterpret_cast<dvis_f4 *>(p + part * 2048) = z;
  *reinterpret_cast<dvis_f4 *>(p + 4096 + part * 2048) = z;
}

// dvis_attention_x3_packed with the output as the row image of the out-projection (dvis_x3_tile_linear_image, weights packed with
// dvis_x3_tile_pack_order(order 2)): rows (batch entry, token), K = heads * 64, values x 2^xexp; `image` holds
// dvis_x3_rows_image_bytes(B * L, heads * 64) bytes.
DVIS_EXPORT int dvis_attention_x3_packed_image(const void *ws, void *image, int B, int heads, int L, int xexp, void *stream) {
  DVIS_REQUIRE(ws && image && B >= 0 && heads > 0 && L >= 128, "attention_x3_packed_image: bad arguments (L >= 128)");
  DVIS_REQUIRE(((uintptr_t)ws | (uintptr_t)image) % 16 == 0, "attention_x3_packed_image: workspace and image must be 16-byte aligned");
  if (B == 0) return DVIS_OK;
  const int BH = B * heads;
  const size_t M = (size_t)B * L;
  const X3Guard gd = dvis_x3_guard();
  if (M & 127) hipLaunchKernelGGL(rows_image_tail_kernel, dim3(heads * 4), dim3(256), 0, (hipStream_t)stream, (char *)image, M, heads * 4);
  hipLaunchKernelGGL((attn_x3_kernel<1, true>), dim3(((BH + 7) / 8) * 8 * ((L + 127) / 128)), dim3(512), 0, (hipStream_t)stream, (const _Float16 *)ws,
                     (float *)nullptr, dvis_strides{0, 0, 0}, BH, heads, L, L, gd.flag, gd.tag, (char *)image, ldexpf(1.f, xexp));
  return dvis_check_launch("attn_x3_kernel (row image)");
}

DVIS_EXPORT int dvis_attention_forward_k(const float *q, const int64_t *q_strides, const float *k, const int64_t *k_strides,
                                         const float *v, const int64_t *v_strides, float *out, const int64_t *o_strides,
                                         const uint8_t *mask, const int32_t *allowed_count, int B, int heads, int Lq,
                                         int Lk, int d, float scale, void *ws, void *stream, int kernel) {
  return attention_launch(q, q_strides, k, k_strides, v, v_strides, out, o_strides, mask, allowed_count, B, heads, Lq, Lk, d,
                          scale, ws, stream, kernel);
}
