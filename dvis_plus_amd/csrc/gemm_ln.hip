// GEMM whose A operand is NORMALISED ON THE WAY IN:  C = act( LN2( LN1(A) + ADD ) W^T + bias + res )  — gfx950,
// v_mfma_f32_16x16x4_f32, deterministic (no atomics, no inter-workgroup waiting).
//
// Why: the referring tracker (dvis_Plus/tracker.py:277-318) is a strictly sequential chain of post-norm blocks over
// 100 x 512 activations — `tgt = norm(identity + attn(...))`, `tgt = norm(tgt + ffn(tgt))` (tracker.py:45-52,
// video_mask2former_transformer_decoder.py:47-50, 166-170) — i.e. GEMM -> add -> LayerNorm -> GEMM, three launches per
// seam at 5-9 us each where the arithmetic is worth 1 us.  Here the producing GEMM keeps its residual epilogue and writes
// the RAW sum; the LayerNorm happens in the CONSUMING GEMM's prologue: a workgroup of csrc/gemm.hip's decomposition (a
// (16 RT) x (16 CT) tile of C, K split over its NW waves, fragments straight from global memory into the MFMA operand
// layout) holds its tile's A rows COMPLETE — K <= 16 NW GW, every wave keeps its GW k-groups of all 16 RT rows in
// registers — so the row statistics are two LDS exchanges between the waves (mean, then centred variance: the two-pass
// form of torch's LayerNorm) while the weight fragments, requested first, are still on their way.  The seam costs no
// launch, no extra pass over the activations and no cross-workgroup hand-off (an in-launch "last arriver normalises"
// combine needs an agent-scope release + acquire, ~3.4 us on this chip — more than the LayerNorm launch it would replace).
// The tracker's chain shrinks from ~65 to 35 launches per frame (dvis_plus_amd/tracker.py).
//
// The two-norm form serves the seam between two tracker layers: x = LN_ffn(y) is the previous layer's output,
// LN_cross(x + t) the next layer's cross-attention block whose attention term t does not depend on the layer chain
// (tracker.py:293-318: q = reference, k / v = the frame's queries for every layer).
// a_out: the normalised rows are also an OUTPUT (the residual of the following block, the frame's result): the workgroups
// of a row block share the store, one k-group each.
#include "dvis_common.h"

namespace {

constexpr unsigned kOOB = 0x80000000u;

struct GemmLnArgs {
  const float *A, *add, *g1, *b1, *g2, *b2, *W, *bias, *res;
  float *a_out, *C;
  long long lda, ldadd, ldaout, ldw, ldres, ldc;
  int M, N, K, act, row_blocks, col_blocks;
  float eps1, eps2;
};

// Sum of v over the four 16-lane rows of the wave (lanes i, i + 16, i + 32, i + 48 hold the four k-quads of tile row i),
// returned in every lane, as two VALU swaps instead of two ds_bpermute round trips through the LDS pipe:
//   v_permlane16_swap a, b (a = b = v)   ->  a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]   (odd rows of a <-> even rows of b)
//   v_permlane32_swap a, b (a = b = s)   ->  a = [s01 s01],     b = [s23 s23]       (upper half of a <-> lower half of b)
// Every lane computes (r0 + r1) + (r2 + r3): one summation order.  (Inline asm: through the builtin hipcc 7.2 folds the
// two results of a swap that is fed one value twice into one; the s_nop cover the VALU -> permlane read hazard the
// compiler does not see inside asm.)
__device__ __forceinline__ float sum_rows4(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  const float s = a + b;
  a = s, b = s;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}

// LN1 / ADD / LN2: which parts of the prologue exist (compile time: a plain call carries no dead loads or registers).
template <int RT, int CT, int NW, int GW, bool LN1, bool ADD, bool LN2>
__global__ __launch_bounds__(64 * NW) void gemm_ln_kernel(const GemmLnArgs p) {
  // ONE LDS array (a second __shared__ object makes hipcc drain vmcnt before LDS reads): [2][NW][BM] statistics exchange
  // (double-buffered: one barrier per reduction), later reused as [NW][BM][BN] partial tiles
  extern __shared__ float lds[];
  constexpr int BM = 16 * RT, BN = 16 * CT, T = 64 * NW;
  constexpr int U = BM * BN / 4, UR = BN / 4, NR = (U + T - 1) / T;   // epilogue: 16-byte units, rounds per thread
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  // Tile of this workgroup.  The row blocks that share a weight slab should also share an L2: dispatch puts block b on XCD
  // b % 8 (observed placement; a wrong guess costs speed only), so with col_blocks % 8 == 0 XCD x takes the column blocks
  // == x (mod 8) and walks each one's row blocks — every weight line is then fetched from MALL / HBM by ONE XCD instead of
  // by min(row_blocks, 8) of them (the weights are the one operand that is never L2-resident: 104 MB per frame cycle).
  int rb, cb;
  if ((p.col_blocks & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    rb = slot % p.row_blocks, cb = (slot / p.row_blocks) * 8 + xcd;
  } else {
    rb = blockIdx.x % p.row_blocks, cb = blockIdx.x / p.row_blocks;
  }
  const int row0 = rb * BM, col0 = cb * BN;

  const int rows_a = min(BM, p.M - row0), rows_w = min(BN, p.N - col0);
  const __amdgpu_buffer_rsrc_t rw =
      dvis_make_rsrc_uniform(p.W + (long long)col0 * p.ldw, (unsigned)((((long long)rows_w - 1) * p.ldw + p.K) * 4));
  const __amdgpu_buffer_rsrc_t ra =
      dvis_make_rsrc_uniform(p.A + (long long)row0 * p.lda, (unsigned)((((long long)rows_a - 1) * p.lda + p.K) * 4));

  // this wave's k-groups: wv * GW + s; K % 16 == 0 (host); groups past K / 16 read zeros (descriptor range) and take no
  // part.  EVERY load of the kernel is requested up front — one memory round trip per launch is the budget of a chain link
  // — in the order they are needed: loads return in order, so the rows and norm parameters go out BEFORE the weights (the
  // long pole: MALL / HBM) and the row statistics run while the weight fragments are still on their way.
  const int KG = p.K >> 4;
  dvis_f4 w[GW][CT], a[GW][RT], ad[ADD ? GW : 1][ADD ? RT : 1];
  dvis_f4 ga1[LN1 ? GW : 1], be1[LN1 ? GW : 1], ga2[LN2 ? GW : 1], be2[LN2 ? GW : 1];
  unsigned koff[GW];   // byte offset of this lane's 16-byte piece inside a row, or out of range
#pragma unroll
  for (int s = 0; s < GW; ++s) koff[s] = wv * GW + s < KG ? (unsigned)((wv * GW + s) * 64 + g * 16) : kOOB;
#pragma unroll
  for (int s = 0; s < GW; ++s)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
      a[s][rt] = __builtin_bit_cast(
          dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)((rt * 16 + i) * p.lda * 4) + koff[s], 0, 0));
  if constexpr (LN1) {
    const __amdgpu_buffer_rsrc_t rg = dvis_make_rsrc_uniform(p.g1, (unsigned)p.K * 4u);
    const __amdgpu_buffer_rsrc_t rb_ = dvis_make_rsrc_uniform(p.b1, (unsigned)p.K * 4u);
#pragma unroll
    for (int s = 0; s < GW; ++s) {
      ga1[s] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rg, koff[s], 0, 0));
      be1[s] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rb_, koff[s], 0, 0));
    }
  }
  if constexpr (ADD) {
    const __amdgpu_buffer_rsrc_t rd = dvis_make_rsrc_uniform(p.add + (long long)row0 * p.ldadd,
                                                             (unsigned)((((long long)rows_a - 1) * p.ldadd + p.K) * 4));
#pragma unroll
    for (int s = 0; s < GW; ++s)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        ad[s][rt] = __builtin_bit_cast(
            dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rd, (unsigned)((rt * 16 + i) * p.ldadd * 4) + koff[s], 0, 0));
  }
  if constexpr (LN2) {
    const __amdgpu_buffer_rsrc_t rg = dvis_make_rsrc_uniform(p.g2, (unsigned)p.K * 4u);
    const __amdgpu_buffer_rsrc_t rb_ = dvis_make_rsrc_uniform(p.b2, (unsigned)p.K * 4u);
#pragma unroll
    for (int s = 0; s < GW; ++s) {
      ga2[s] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rg, koff[s], 0, 0));
      be2[s] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rb_, koff[s], 0, 0));
    }
  }
#pragma unroll
  for (int s = 0; s < GW; ++s)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      w[s][ct] = __builtin_bit_cast(
          dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)((ct * 16 + i) * p.ldw * 4) + koff[s], 0, 0));
  // the epilogue's operands as well (its thread -> unit mapping is fixed): bias and residual of this thread's units
  dvis_f4 bia[NR], rsd[NR];
  {
    const __amdgpu_buffer_rsrc_t rbias = dvis_make_rsrc_uniform(p.bias ? p.bias : p.A, p.bias ? (unsigned)p.N * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rres = dvis_make_rsrc_uniform(
        p.res ? p.res + (long long)row0 * p.ldres : p.A, p.res ? (unsigned)((((long long)rows_a - 1) * p.ldres + p.N) * 4) : 0u);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int u = r * T + tid;
      const int row = u / UR, c4 = (u - row * UR) * 4;
      const bool ok = u < U && col0 + c4 < p.N;
      bia[r] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rbias, ok ? (unsigned)(col0 + c4) * 4u : kOOB, 0, 0));
      rsd[r] = __builtin_bit_cast(
          dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rres, ok ? (unsigned)((row * p.ldres + col0 + c4) * 4) : kOOB, 0, 0));
    }
  }

  // ---- row statistics over the waves: partial per (wave, row) -> LDS -> every lane sums the NW partials in wave order
  int round = 0;
  auto row_sums = [&](float (&part)[RT], float (&tot)[RT]) {
    float *buf = lds + (round & 1) * (NW * BM);
    ++round;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const float v = sum_rows4(part[rt]);
      if (g == 0) buf[wv * BM + rt * 16 + i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) t += buf[k * BM + rt * 16 + i];
      tot[rt] = t;
    }
  };
  const float invK = 1.f / (float)p.K;
  auto layer_norm = [&](const dvis_f4 *ga, const dvis_f4 *be, float eps) {
    float part[RT], mean[RT], rstd[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < GW; ++s) v += (a[s][rt][0] + a[s][rt][1]) + (a[s][rt][2] + a[s][rt][3]);
      part[rt] = v;
    }
    row_sums(part, mean);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      mean[rt] *= invK;
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < GW; ++s) {
        if (wv * GW + s < KG) {   // wave-uniform; groups past K are zeros, not (0 - mean)
          const float d0 = a[s][rt][0] - mean[rt], d1 = a[s][rt][1] - mean[rt], d2 = a[s][rt][2] - mean[rt],
                      d3 = a[s][rt][3] - mean[rt];
          v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
      }
      part[rt] = v;
    }
    row_sums(part, rstd);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      rstd[rt] = rsqrtf(rstd[rt] * invK + eps);
#pragma unroll
      for (int s = 0; s < GW; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[s][rt][c] = (a[s][rt][c] - mean[rt]) * rstd[rt] * ga[s][c] + be[s][c];
      // (groups past K: gamma = beta = 0 through the descriptor -> the fragment stays 0)
    }
  };
  if constexpr (LN1) layer_norm(ga1, be1, p.eps1);
  if constexpr (ADD) {
#pragma unroll
    for (int s = 0; s < GW; ++s)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < 4; ++c) a[s][rt][c] += ad[s][rt][c];
  }
  if constexpr (LN2) layer_norm(ga2, be2, p.eps2);

  // a_out: every workgroup of a row block holds the same normalised rows; each k-group is stored by ONE of them (column
  // block kg mod min(col_blocks, KG)) — one workgroup storing all of it (64 KB at 32 rows x 512) finishes ~1.2 us after
  // the rest of the launch
  if (p.a_out) {
    const int nst = min(p.col_blocks, KG);
#pragma unroll
    for (int s = 0; s < GW; ++s) {
      const int kg = wv * GW + s;
      if (kg < KG && kg % nst == cb) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int gr = row0 + rt * 16 + i;
          if (gr < p.M) *reinterpret_cast<dvis_f4 *>(p.a_out + (long long)gr * p.ldaout + kg * 16 + g * 4) = a[s][rt];
        }
      }
    }
  }

  // ---- contraction: lane (i, g) holds k = 16 kg + 4 g + c of row i (A) / column i (W); MFMA c of a group sums over g
  dvis_f4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = dvis_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < GW; ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][rt][c], w[s][ct][c], acc[rt][ct], 0, 0, 0);

  // ---- partial tiles meet in LDS (it aliases the statistics buffers: every wave must be past its last read of them)
  if constexpr (LN1 || LN2) __syncthreads();
  float *mine = lds + wv * (BM * BN);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(rt * 16 + g * 4 + r) * BN + ct * 16 + i] = acc[rt][ct][r];
  __syncthreads();

#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int u = r * T + tid;
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
    s = dvis_f4{s[0] + bia[r][0], s[1] + bia[r][1], s[2] + bia[r][2], s[3] + bia[r][3]};   // (none: zeros)
    s = dvis_f4{s[0] + rsd[r][0], s[1] + rsd[r][1], s[2] + rsd[r][2], s[3] + rsd[r][3]};
    if (p.act) s = dvis_f4{fmaxf(s[0], 0.f), fmaxf(s[1], 0.f), fmaxf(s[2], 0.f), fmaxf(s[3], 0.f)};
    *reinterpret_cast<dvis_f4 *>(p.C + (long long)gr * p.ldc + gc) = s;
  }
}

typedef void (*LnKernel)(const GemmLnArgs);
struct LnConfig {
  int rt, ct, nw, gw;
  LnKernel kernel[4];   // [0] plain, [1] LN1, [2] ADD + LN2, [3] LN1 + ADD + LN2 (null: not built for this configuration)
};
#define DVIS_GEMM_LN_CFG(RT_, CT_, NW_, GW_)                                                                   \
  { RT_, CT_, NW_, GW_,                                                                                        \
    { gemm_ln_kernel<RT_, CT_, NW_, GW_, false, false, false>, gemm_ln_kernel<RT_, CT_, NW_, GW_, true, false, false>, \
      gemm_ln_kernel<RT_, CT_, NW_, GW_, false, true, true>, gemm_ln_kernel<RT_, CT_, NW_, GW_, true, true, true> } }
#define DVIS_GEMM_PLAIN_CFG(RT_, CT_, NW_, GW_) \
  { RT_, CT_, NW_, GW_, { gemm_ln_kernel<RT_, CT_, NW_, GW_, false, false, false>, nullptr, nullptr, nullptr } }
const LnConfig kLnConfigs[] = {
    DVIS_GEMM_LN_CFG(2, 1, 8, 4),      // 0: 32 x 16 tile, K <= 512
    DVIS_GEMM_LN_CFG(2, 2, 8, 4),      // 1: 32 x 32
    DVIS_GEMM_LN_CFG(1, 2, 8, 4),      // 2: 16 x 32
    DVIS_GEMM_LN_CFG(1, 1, 8, 4),      // 3: 16 x 16
    DVIS_GEMM_LN_CFG(2, 1, 4, 4),      // 4: 32 x 16, 4 waves, K <= 256
    DVIS_GEMM_LN_CFG(2, 2, 4, 4),      // 5: 32 x 32, 4 waves, K <= 256
    DVIS_GEMM_PLAIN_CFG(1, 1, 8, 16),  // 6: 16 x 16, K <= 2048 (the FFN's second projection: its 128 k-groups in ONE round trip)
    DVIS_GEMM_PLAIN_CFG(2, 1, 8, 16),  // 7: 32 x 16, K <= 2048
    DVIS_GEMM_PLAIN_CFG(1, 2, 8, 16),  // 8: 16 x 32, K <= 2048
    DVIS_GEMM_PLAIN_CFG(1, 1, 16, 8),  // 9: 16 x 16, 16 waves, K <= 2048
};
constexpr int kNumLnConfigs = sizeof(kLnConfigs) / sizeof(kLnConfigs[0]);

// From the sizes alone (a shape always runs the same summation order): the skinny problems of the tracker want every CU
// to have a workgroup; K decides how deep the split is.
int pick_ln_config(int M, int N, int K) {
  auto wgs = [&](int c) {
    const long long rb = (M + 16 * kLnConfigs[c].rt - 1) / (16 * kLnConfigs[c].rt);
    const long long cb = (N + 16 * kLnConfigs[c].ct - 1) / (16 * kLnConfigs[c].ct);
    return rb * cb;
  };
  // measured on MI355X as links of a hipGraph chain at M = 100 (tools/gemm_ln_time.py): the smallest tile while the launch
  // is at most one workgroup per CU, then 32 x 32, then 16 x 32
  if (K > 512) return 9;
  if (K <= 256) return wgs(5) >= 192 ? 5 : 4;
  if (wgs(3) <= 256) return 3;
  if (wgs(1) <= 256) return 1;
  return 2;
}

}  // namespace

DVIS_EXPORT int dvis_gemm_ln_num_configs(void) { return kNumLnConfigs; }

DVIS_EXPORT int dvis_gemm_ln_pick_config(int M, int N, int K) { return pick_ln_config(M, N, K); }

// norms != 0: the call carries a LayerNorm (or the ADD operand) — K <= 512; plain calls reach K <= 2048
DVIS_EXPORT int dvis_gemm_ln_supported(int M, int N, int K, int norms) {
  return M > 0 && N > 0 && K > 0 && K % 16 == 0 && K <= (norms ? 512 : 2048) && N % 4 == 0;
}

DVIS_EXPORT int dvis_gemm_ln(const float *A, int64_t lda, const float *add, int64_t ldadd, const float *gamma1,
                             const float *beta1, float eps1, const float *gamma2, const float *beta2, float eps2,
                             float *a_out, int64_t ldaout, const float *W, int64_t ldw, const float *bias, const float *res,
                             int64_t ldres, float *C, int64_t ldc, int M, int N, int K, int act, int config, void *stream) {
  DVIS_REQUIRE(M >= 0 && N >= 0 && K > 0, "gemm_ln: bad sizes (M=%d N=%d K=%d)", M, N, K);
  if (M == 0 || N == 0) return DVIS_OK;
  DVIS_REQUIRE(A && W && C, "gemm_ln: null pointer");
  DVIS_REQUIRE((gamma1 == nullptr) == (beta1 == nullptr) && (gamma2 == nullptr) == (beta2 == nullptr),
               "gemm_ln: a LayerNorm needs both gamma and beta");
  DVIS_REQUIRE((add == nullptr) == (gamma2 == nullptr), "gemm_ln: the forms are LN1, ADD + LN2, LN1 + ADD + LN2 or none");
  const int variant = (gamma1 ? 1 : 0) + (gamma2 ? 2 : 0);
  DVIS_REQUIRE(dvis_gemm_ln_supported(M, N, K, variant),
               "gemm_ln: needs K %% 16 == 0, K <= 512 (2048 without norms), N %% 4 == 0 (N=%d K=%d)", N, K);
  DVIS_REQUIRE(lda % 4 == 0 && ldw % 4 == 0 && ldc % 4 == 0 && lda >= K && ldw >= K && ldc >= N &&
                   (!add || (ldadd % 4 == 0 && ldadd >= K)) && (!a_out || (ldaout % 4 == 0 && ldaout >= K)) &&
                   (!res || (ldres % 4 == 0 && ldres >= N)),
               "gemm_ln: row strides must be multiples of 4 floats and cover the rows");
  const uintptr_t al = (uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)add | (uintptr_t)a_out | (uintptr_t)bias |
                       (uintptr_t)res | (uintptr_t)gamma1 | (uintptr_t)beta1 | (uintptr_t)gamma2 | (uintptr_t)beta2;
  DVIS_REQUIRE(al % 16 == 0, "gemm_ln: every operand must be 16-byte aligned");
  const int c = config >= 0 ? config : pick_ln_config(M, N, K);
  DVIS_REQUIRE(c < kNumLnConfigs, "gemm_ln: configuration %d does not exist", c);
  const LnConfig &cf = kLnConfigs[c];
  DVIS_REQUIRE(K <= 16 * cf.nw * cf.gw, "gemm_ln: configuration %d holds K <= %d (K=%d)", c, 16 * cf.nw * cf.gw, K);
  const LnKernel kernel = cf.kernel[variant];
  DVIS_REQUIRE(kernel != nullptr, "gemm_ln: configuration %d holds K <= %d without norms only", c, 16 * cf.nw * cf.gw);
  const int BM = 16 * cf.rt, BN = 16 * cf.ct;
  DVIS_REQUIRE((long long)BM * lda * 4 < (1ll << 31) && (long long)BN * ldw * 4 < (1ll << 31) &&
                   (!add || (long long)BM * ldadd * 4 < (1ll << 31)) && (!res || (long long)BM * ldres * 4 < (1ll << 31)),
               "gemm_ln: one tile's rows must span less than 2 GiB");
  GemmLnArgs p;
  p.A = A, p.add = add, p.g1 = gamma1, p.b1 = beta1, p.g2 = gamma2, p.b2 = beta2, p.W = W, p.bias = bias, p.res = res;
  p.a_out = a_out, p.C = C;
  p.lda = lda, p.ldadd = add ? ldadd : 0, p.ldaout = ldaout, p.ldw = ldw, p.ldres = res ? ldres : 0, p.ldc = ldc;
  p.M = M, p.N = N, p.K = K, p.act = act, p.eps1 = eps1, p.eps2 = eps2;
  p.row_blocks = (M + BM - 1) / BM;
  p.col_blocks = (N + BN - 1) / BN;
  const long long tiles = (long long)p.row_blocks * p.col_blocks;
  DVIS_REQUIRE(tiles < (1ll << 31), "gemm_ln: too many tiles");
  size_t lds = (size_t)cf.nw * BM * BN * sizeof(float);
  const size_t stats = (size_t)2 * cf.nw * BM * sizeof(float);
  if (lds < stats) lds = stats;
  static DvisLdsOptIn opted[kNumLnConfigs][4];
  if (const int rc = dvis_lds_opt_in(reinterpret_cast<const void *>(kernel), lds, &opted[c][variant], "gemm_ln")) return rc;
  hipLaunchKernelGGL(kernel, dim3((unsigned)tiles), dim3(64 * cf.nw), lds, (hipStream_t)stream, p);
  return dvis_check_launch("gemm_ln_kernel");
}
