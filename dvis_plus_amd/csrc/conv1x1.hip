// 1x1 stride-1 convolution on NCHW with its epilogue in the same pass — gfx950, exact-fp32 MFMA.
//
//   out[n, m, p] = relu?( sum_k w[m, k] * x[n, k, p] + bias[m] (+ res[n, m, p]) )
//
// Replaces, in the detectron2-style bottleneck of the R50 front-end (conv1 / conv3 / the stride-1 shortcut; FrozenBN folded
// into w and bias by backbone.py), a library contraction FOLLOWED by dvis_bias_act: at the large maps (184x320, 92x160)
// those layers are memory-bound — conv3 of a res2 block (64 -> 256 channels) writes 1.8 GB per 30-frame clip, and the
// separate bias + shortcut + ReLU pass read and wrote it again (MIOpen 1.34 ms + epilogue 1.20 ms against 4.06 GB of
// compulsory traffic = 0.8 ms at 5 TB/s).  MIOpen's own fused conv + bias + activation was measured slower than the two
// separate kernels.
//
// Structure = mask_gemm.hip's pixel partition: a wave owns 16 * NT pixels and ALL QT row tiles of the pass; its B operand
// (activations) goes from global memory straight into the MFMA operand layout (lane (j, g): NT consecutive pixels of
// channel g*CQ + u), the A operand (weights, the same for every frame) is loaded into LDS ONCE per workgroup as
// [lane group][row][CQ + 4].  Passes over rows (Co > 16 QT) re-read the activations; the channels of one pass must fit
// the LDS (K <= 128 at 256 rows, <= 256 at 128 rows) — deeper / wider layers stay with the library, which is compute-bound
// there and good at it.
#include "dvis_common.h"

namespace {

// Out-of-range buffer offset.  It is used together with SCALAR offsets (channel / row), and the hardware adds the two
// before its range check, in 32 bits: 0xFFFFFF00 + a row offset wrapped back INTO the slab (a "dropped" load returned
// stale memory).  2 GiB + any offset into a < 2 GiB slab neither wraps nor lands inside it (the host checks the sizes).
constexpr unsigned kOOB = 0x80000000u;
constexpr int kPad = 4;                  // floats added to an LDS row (16-lane groups of b128 reads hit all banks)

template <int NT> struct PixVec;
template <> struct PixVec<4> { typedef dvis_f4 type; };
template <> struct PixVec<2> { typedef float type __attribute__((ext_vector_type(2))); };

template <int QT, int NT>
__global__ __launch_bounds__(512) void conv1x1_kernel(
    const float *__restrict__ w, const float *__restrict__ x, const float *__restrict__ bias,
    const float *__restrict__ res, float *__restrict__ out, int M, int mbeg, int K, int CQ, long long HW, int relu, int gpf,
    int spf, int total_steps) {
  extern __shared__ float lds[];   // A: [4 lane groups][QT * 16 rows][CQ + kPad]
  constexpr int ROWS = QT * 16;
  constexpr int PG = 16 * NT;      // pixels per wave group
  typedef typename PixVec<NT>::type pix_t;
  const int arow_len = CQ + kPad;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const unsigned chan_bytes = (unsigned)(HW * sizeof(float));

  // ---- A: rows mbeg .. mbeg + ROWS of the weights, once (they do not depend on the frame)
  {
    const int n4 = CQ >> 2, total4 = 4 * ROWS * n4;
    for (int base = 0; base < total4; base += 4 * 512) {   // 4 granules in flight per thread
      dvis_f4 v[4];
      int dst[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = base + k * 512 + tid;
        const int u4 = idx % n4, rest = idx / n4;
        const int row = rest % ROWS, gg = rest / ROWS;
        const int m = mbeg + row, c = gg * CQ + u4 * 4;
        dst[k] = idx < total4 ? (gg * ROWS + row) * arow_len + u4 * 4 : -1;
        const bool live = idx < total4 && m < M && c < K;   // K % 4 == 0: a granule is all in or all out
        v[k] = live ? *reinterpret_cast<const dvis_f4 *>(w + (size_t)m * K + c) : dvis_f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (dst[k] >= 0) *reinterpret_cast<dvis_f4 *>(&lds[dst[k]]) = v[k];
    }
  }
  __syncthreads();

  const int s_lo = (int)((long long)blockIdx.x * total_steps / gridDim.x);
  const int s_hi = (int)((long long)(blockIdx.x + 1) * total_steps / gridDim.x);
  const int ulim = K - g * CQ;   // this lane group's channels are u < ulim
  const float *arow = lds + (g * ROWS + j) * arow_len;

  // pixel columns of a wave group: byte offset of this lane's NT pixels, lane group's first channel included (a column
  // that does not exist gets an out-of-range offset and reads 0); the channel of a k-step goes into the SCALAR offset
  auto columns = [&](int group) -> unsigned {
    const long long p = (long long)group * PG + NT * j;
    return p < HW ? (unsigned)(g * CQ) * chan_bytes + (unsigned)(p * 4) : kOOB;
  };
  pix_t b0[4], b1[4];
  bool have_b0 = false;   // wave-uniform: b0 already holds k-steps 0..3 of this step's group — requested under the last
                          // MFMAs of the previous step (a 64-channel layer is 512 MFMAs per group: not long enough to
                          // sit out a global round trip at its start)

#pragma unroll 1
  for (int step = s_lo; step < s_hi; ++step) {
    const int b = step / spf, sidx = step - b * spf;
    const int group = sidx * 8 + wv;
    if (group >= gpf) continue;   // wave-uniform; the loop has no barrier (have_b0 is false: only existing groups are prefetched)
    const __amdgpu_buffer_rsrc_t rs = dvis_make_rsrc_uniform(x + (size_t)b * K * HW, (unsigned)((size_t)K * HW * sizeof(float)));
    auto load_b = [&](int u0, unsigned voff, pix_t(&dst)[4]) {   // channels g*CQ + u0 .. + 3
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned so = (unsigned)(u0 + i) * chan_bytes;
        const unsigned vo = (u0 + i < ulim) ? voff : kOOB;
        if constexpr (NT == 4)
          dst[i] = __builtin_bit_cast(pix_t, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0));
        else
          dst[i] = __builtin_bit_cast(pix_t, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, 0));
      }
    };
    const long long p0 = (long long)group * PG + NT * j;
    const bool ok = p0 < HW;      // HW % NT == 0: the lane's NT pixels exist together
    const unsigned voff = columns(group);
    if (!have_b0) load_b(0, voff, b0);
    have_b0 = false;

    dvis_f4 acc[QT][NT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[qt][n] = dvis_f4{0.f, 0.f, 0.f, 0.f};
    // 4 k-steps against the B fragment `bf`, QH row tiles at a time (all 16 tiles' A fragments at once would be 64 VGPRs
    // on top of 128 accumulators)
    constexpr int QH = QT > 8 ? 8 : QT;
    auto contract = [&](int u0, const pix_t(&bf)[4]) {
#pragma unroll
      for (int h = 0; h < QT / QH; ++h) {
        dvis_f4 af[QH];
#pragma unroll
        for (int q = 0; q < QH; ++q) af[q] = *reinterpret_cast<const dvis_f4 *>(arow + (h * QH + q) * 16 * arow_len + u0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < QH; ++q)
#pragma unroll
            for (int n = 0; n < NT; ++n)
              acc[h * QH + q][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[q][i], bf[i][n], acc[h * QH + q][n], 0, 0, 0);
      }
    };

#pragma unroll 1
    for (int u0 = 0; u0 < CQ; u0 += 8) {   // CQ % 8 == 0
      load_b(u0 + 4, voff, b1);
      contract(u0, b0);
      if (u0 + 8 < CQ) {   // uniform
        load_b(u0 + 8, voff, b0);
      } else if (step + 1 < s_hi) {
        const int nb = (step + 1) / spf, ng = (step + 1 - nb * spf) * 8 + wv;
        if (nb == b && ng < gpf) {   // same frame (same descriptor), and the group exists
          load_b(0, columns(ng), b0);
          have_b0 = true;
        }
      }
      contract(u0 + 4, b1);
    }

    // ---- epilogue: + bias (+ shortcut) (ReLU), NT consecutive pixels per store.
    // Accumulator layout: column (pixel) = lane & 15, row = (lane >> 4) * 4 + reg.
    const size_t frame = (size_t)b * M * HW;
    const __amdgpu_buffer_rsrc_t ro = dvis_make_rsrc_uniform(out + frame, (unsigned)((size_t)M * HW * sizeof(float)));
    const __amdgpu_buffer_rsrc_t rr =
        dvis_make_rsrc_uniform(res ? res + frame : out + frame, (unsigned)((size_t)M * HW * sizeof(float)));
    const unsigned vo = (unsigned)(((size_t)(mbeg + g * 4) * HW + p0) * 4);
    // Shortcut values are requested a CHUNK of row tiles ahead of their use (EP tiles = 4 EP loads in flight per lane):
    // requested two rows at a time the epilogue was a chain of 32 dependent global round trips per pixel group.
    constexpr int EP = QT >= 4 ? 4 : QT;
    static_assert(QT % EP == 0, "epilogue chunk");
#pragma unroll
    for (int qc = 0; qc < QT; qc += EP) {
      pix_t rv[EP][4];
      dvis_f4 bv[EP];
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const int qt = qc + e;
        const int row0 = mbeg + qt * 16 + g * 4;
        // bias of this lane's 4 rows (M % 4 == 0: all four exist or none)
        bv[e] = (bias != nullptr && row0 < M) ? *reinterpret_cast<const dvis_f4 *>(bias + row0) : dvis_f4{0.f, 0.f, 0.f, 0.f};
        const unsigned v = (ok && row0 < M) ? vo : kOOB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned so = (unsigned)(qt * 16 + r) * chan_bytes;
          if (res == nullptr) {   // uniform
#pragma unroll
            for (int n = 0; n < NT; ++n) rv[e][r][n] = 0.f;
          } else if constexpr (NT == 4) {
            rv[e][r] = __builtin_bit_cast(pix_t, __builtin_amdgcn_raw_buffer_load_b128(rr, v, so, 0));
          } else {
            rv[e][r] = __builtin_bit_cast(pix_t, __builtin_amdgcn_raw_buffer_load_b64(rr, v, so, 0));
          }
        }
      }
#pragma unroll
      for (int e = 0; e < EP; ++e) {
        const int qt = qc + e;
        const int row0 = mbeg + qt * 16 + g * 4;
        if (!ok || row0 >= M) continue;   // (a branch, not an out-of-range offset: stores must not depend on the range check)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pix_t v;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const float t = acc[qt][n][r] + bv[e][r] + rv[e][r][n];
            v[n] = relu ? fmaxf(t, 0.f) : t;
          }
          // The row offset goes into the VECTOR offset of the store, not into its scalar offset: a store of more than
          // 8 bytes reads its data registers for a few cycles after issue, and LLVM only pads the next VALU write to them
          // when the store has no SGPR offset (it takes the hazard not to exist otherwise).  On MI355X it does exist:
          // with `so` in the scalar field, dword 0 of lanes 12-15 of row r was stored with row r + 1's value.
          const unsigned so = (unsigned)(qt * 16 + r) * chan_bytes;
          if constexpr (NT == 4)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dvis_v4u, v), ro, vo + so, 0, 0);
          else
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(unsigned __attribute__((ext_vector_type(2))), v), ro, vo + so, 0, 0);
        }
      }
    }
  }
}

template <int QT, int NT>
int launch(const float *w, const float *x, const float *bias, const float *res, float *out, int N, int K, int M, long long HW,
           int relu, hipStream_t st) {
  const int CQ = ((K + 3) / 4 + 7) / 8 * 8;
  const size_t lds = (size_t)4 * QT * 16 * (CQ + kPad) * sizeof(float);
  static DvisLdsOptIn opted;   // per kernel instantiation, per device
  if (const int rc = dvis_lds_opt_in(reinterpret_cast<const void *>(&conv1x1_kernel<QT, NT>), lds, &opted, "conv1x1"))
    return rc;
  const int PG = 16 * NT;
  const int gpf = (int)((HW + PG - 1) / PG);
  const int spf = (gpf + 7) / 8;
  const long long total = (long long)N * spf;
  const int per_cu = lds > 80 * 1024 ? 1 : 2;
  const int nwg = (int)(total < 256 * per_cu ? total : 256 * per_cu);
  for (int mbeg = 0; mbeg < M; mbeg += QT * 16)
    hipLaunchKernelGGL((conv1x1_kernel<QT, NT>), dim3(nwg), dim3(512), lds, st, w, x, bias, res, out, M, mbeg, K, CQ, HW, relu,
                       gpf, spf, (int)total);
  return dvis_check_launch("conv1x1_kernel");
}

}  // namespace

// 1 if dvis_conv1x1_bias_act serves this shape (the caller keeps the library path otherwise)
DVIS_EXPORT int dvis_conv1x1_supported(int K, int M, int64_t HW) {
  if (K <= 0 || M <= 0 || HW <= 0 || K % 4 != 0 || HW % 4 != 0 || M % 4 != 0) return 0;
  if ((long long)((K + 31) / 32 * 32) * HW * 4 >= (1ll << 31) || (long long)M * HW * 4 >= (1ll << 31)) return 0;
  if (M <= 64) return K <= 512;
  if (M <= 128) return K <= 256;
  return K <= 128;   // passes of 256 rows
}

DVIS_EXPORT int dvis_conv1x1_bias_act(const float *x, const float *w, const float *bias, const float *res, float *out,
                                      int N, int K, int M, int64_t HW, int relu, void *stream) {
  DVIS_REQUIRE(N >= 0 && K > 0 && M > 0 && HW > 0, "conv1x1: bad sizes");
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(x && w && out, "conv1x1: null pointer");
  DVIS_REQUIRE(dvis_conv1x1_supported(K, M, HW), "conv1x1: unsupported shape K=%d M=%d HW=%lld (dvis_conv1x1_supported)", K, M,
               (long long)HW);
  DVIS_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)out | (uintptr_t)res) & 15) == 0, "conv1x1: 16-byte aligned tensors");
  hipStream_t st = (hipStream_t)stream;
  if (M <= 64) return launch<4, 4>(w, x, bias, res, out, N, K, M, HW, relu, st);
  if (M <= 128) return launch<8, 2>(w, x, bias, res, out, N, K, M, HW, relu, st);   // (8 x 4 accumulator tiles spill)
  return launch<16, 2>(w, x, bias, res, out, N, K, M, HW, relu, st);
}
