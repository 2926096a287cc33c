// Multi-scale deformable attention, backward — gfx950.
//
// Replaces ms_deformable_col2im_cuda and its 7 kernel variants
// (ops/src/cuda/ms_deform_im2col_cuda.cuh:306-925, host :961-1331; wrapper ms_deform_attn_cuda.cu:88-158).
// The reference picks a kernel by D (blocksize-aware / dynamic-smem / multi-block / global-atomic) because it
// assigns one thread per channel and reduces grad_loc / grad_w through shared memory.  Here a GROUP of 32 or
// 64 lanes (half / full wavefront) owns one (n, q, m) pair for every D: lanes stride over channels, the three
// scalar gradients of a sample are reduced with wavefront shuffles (no LDS, no __syncthreads), and grad_value
// gets lane-contiguous atomics (one 128-byte line per corner for D = 32).
//   d out / d w      = sum_c go[c] * bilinear(v)[c]
//   d out / d loc_x  = W * w * sum_c go[c] * (hh (v2 - v1) + lh (v4 - v3))      (corners outside the map = 0)
//   d out / d loc_y  = H * w * sum_c go[c] * (hw (v3 - v1) + lw (v4 - v2))
//   d out / d v_k[c] = go[c] * w * corner_weight_k
// Measured (round 4, tools/msda_bwd_time.py, profiles/r04_msda_bwd_time.txt): 695 us per 720p frame-layer (S = Lq = 19 320,
// M = 8, D = 32) = 0.019 of the HBM roofline on 103.9 MB of algorithmic traffic, 935 us at the ViT-L extractor shape; the forward
// takes 41 / 56 us.  The 237 M scalar adds per frame-layer are 7.4 M 128-byte-line atomic transactions (a 32-lane group hits one
// line per instruction) and that is the bound: ~10.6 G line-atomics/s.  A lane-owned 16-byte form (8 lanes x 4 channels per
// pair, as in the forward) was built and is 4x SLOWER (2808 us): every atomic instruction then touches a quarter of each
// line, i.e. four times as many L2 transactions.  What would help is fewer transactions — grad_value accumulated on chip
// per query tile before one atomic per line — not wider lanes.
#include "dvis_common.h"

#include <algorithm>

namespace {

template <typename A, int GS>
__device__ __forceinline__ A group_sum(A v) {
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, GS);
  return v;
}

// DET (fp32 only): grad_value is accumulated in 64-bit FIXED POINT — every contribution is rounded to a multiple of 2^-e on its own
// (e from the launch's max |grad_out| x max |weight|: 42 bits below the largest possible contribution, 20 bits of headroom above it for
// colliding contributions; the resolution is ABSOLUTE — 2^-42 of that bound — not relative to a cell's own sum)
// and integer addition is associative, so the sum does not depend on the order the atomics land in: two runs give the same bits,
// which float atomics do not (the reference's backward is not reproducible either: ms_deform_im2col_cuda.cuh atomicAdd).
struct DetArgs {
  unsigned long long *acc;           // N * S * M * D accumulators (zeroed by the host function)
  const unsigned *absmax;            // bit patterns of max |grad_out|, max |attn weight|
};

__device__ __forceinline__ float det_scale(const unsigned *absmax) {
  const float bound = __uint_as_float(absmax[0]) * __uint_as_float(absmax[1]);
  int ex = 0;
  frexpf(bound > 0.f && bound < 3.0e38f ? bound : 1.f, &ex);      // bound < 2^ex
  return ldexpf(1.f, 42 - ex);
}

template <typename T, int GS, bool DET = false>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const T *__restrict__ loc, const T *__restrict__ w, const T *__restrict__ grad_out, size_t npairs, int S, int M,
    int D, int L, int Lq, int P, T *__restrict__ grad_value, T *__restrict__ grad_loc, T *__restrict__ grad_w,
    DetArgs det = {}) {
  constexpr int GPB = 256 / GS;
  float dscale = 0.f;
  if constexpr (DET) dscale = det_scale(det.absmax);
  const int gl = threadIdx.x % GS;
  const size_t pix = (size_t)M * D;
  for (size_t pair = (size_t)blockIdx.x * GPB + threadIdx.x / GS; pair < npairs; pair += (size_t)gridDim.x * GPB) {
    const int m = (int)(pair % M);
    const size_t n = pair / ((size_t)Lq * M);
    const T *go = grad_out + pair * D;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const size_t lbase = ((n * S + (size_t)level_start[l]) * M + m) * D;
      for (int p = 0; p < P; ++p) {
        const size_t s = pair * (size_t)(L * P) + (size_t)l * P + p;
        const T x = loc[2 * s], y = loc[2 * s + 1], aw = w[s];
        const T h_im = y * (T)H - (T)0.5, w_im = x * (T)W - (T)0.5;
        T gw = 0, gx = 0, gy = 0;
        if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {   // group-uniform branch
          const T hf = floor(h_im), wf = floor(w_im);
          const int h0 = (int)hf, w0 = (int)wf;
          const T lh = h_im - hf, lw = w_im - wf, hh = (T)1 - lh, hw = (T)1 - lw;
          const bool ok1 = h0 >= 0 && w0 >= 0, ok2 = h0 >= 0 && w0 + 1 <= W - 1;
          const bool ok3 = h0 + 1 <= H - 1 && w0 >= 0, ok4 = h0 + 1 <= H - 1 && w0 + 1 <= W - 1;
          const long long i1 = (long long)lbase + ((long long)h0 * W + w0) * (long long)pix;
          const long long i2 = i1 + (long long)pix, i3 = i1 + (long long)W * (long long)pix, i4 = i3 + (long long)pix;
          for (int c = gl; c < D; c += GS) {
            const T a = ok1 ? value[i1 + c] : (T)0, b = ok2 ? value[i2 + c] : (T)0;
            const T e = ok3 ? value[i3 + c] : (T)0, f = ok4 ? value[i4 + c] : (T)0;
            const T g = go[c];
            gw += g * (hh * hw * a + hh * lw * b + lh * hw * e + lh * lw * f);
            gx += g * (hh * (b - a) + lh * (f - e));
            gy += g * (hw * (e - a) + lw * (f - b));
            const T ga = g * aw;
            if constexpr (DET) {
              auto add = [&](long long idx, float v) {
                atomicAdd(det.acc + idx, (unsigned long long)__float2ll_rn(v * dscale));      // two's complement: signed sums wrap correctly
              };
              if (ok1) add(i1 + c, (float)(ga * hh * hw));
              if (ok2) add(i2 + c, (float)(ga * hh * lw));
              if (ok3) add(i3 + c, (float)(ga * lh * hw));
              if (ok4) add(i4 + c, (float)(ga * lh * lw));
            } else {
              if (ok1) atomicAdd(grad_value + i1 + c, ga * hh * hw);
              if (ok2) atomicAdd(grad_value + i2 + c, ga * hh * lw);
              if (ok3) atomicAdd(grad_value + i3 + c, ga * lh * hw);
              if (ok4) atomicAdd(grad_value + i4 + c, ga * lh * lw);
            }
          }
        }
        gw = group_sum<T, GS>(gw);
        gx = group_sum<T, GS>(gx);
        gy = group_sum<T, GS>(gy);
        if (gl == 0) {
          grad_w[s] = gw;
          grad_loc[2 * s] = gx * aw * (T)W;
          grad_loc[2 * s + 1] = gy * aw * (T)H;
        }
      }
    }
  }
}

template <typename T>
int launch_bwd(const void *value, const int64_t *shapes, const int64_t *ls, const void *loc, const void *w,
               const void *go, int N, int S, int M, int D, int L, int Lq, int P, void *gv, void *gl, void *gw,
               hipStream_t st) {
  const size_t npairs = (size_t)N * Lq * M;
  if (D <= 32) {
    const size_t blocks = (npairs + 7) / 8;
    hipLaunchKernelGGL((msda_bwd_kernel<T, 32>), dim3((unsigned)(blocks > 262144 ? 262144 : blocks)), dim3(256), 0, st,
                       (const T *)value, shapes, ls, (const T *)loc, (const T *)w, (const T *)go, npairs, S, M, D, L,
                       Lq, P, (T *)gv, (T *)gl, (T *)gw);
  } else {
    const size_t blocks = (npairs + 3) / 4;
    hipLaunchKernelGGL((msda_bwd_kernel<T, 64>), dim3((unsigned)(blocks > 262144 ? 262144 : blocks)), dim3(256), 0, st,
                       (const T *)value, shapes, ls, (const T *)loc, (const T *)w, (const T *)go, npairs, S, M, D, L,
                       Lq, P, (T *)gv, (T *)gl, (T *)gw);
  }
  return dvis_check_launch("msda_bwd_kernel");
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, size_t n, unsigned *__restrict__ out) {
  // fmaxf DROPS a NaN operand; the fixed-point form must see it (det_finish_kernel hands back NaN then): a NaN-keeping maximum
  auto nanmax = [](float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); };
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = nanmax(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = nanmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns; NaN -> 0x7fc..: the largest
}

__global__ __launch_bounds__(256) void det_finish_kernel(const unsigned long long *__restrict__ acc, const unsigned *__restrict__ absmax,
                                                         float *__restrict__ grad_value, size_t n) {
  const float inv = 1.f / det_scale(absmax);
  // A NaN / Inf in grad_output or the weights (absmax keeps the largest BIT PATTERN: NaN > Inf > finite) has no fixed-point image:
  // the float-atomic form and the reference would hand back non-finite gradients, so does this one — a diverged training run must
  // not be hidden by torch.use_deterministic_algorithms(True).  Headroom: a cell's sum may reach 2^20 x the bound
  // (max |grad_out| x max |weight|) before the 64-bit accumulator wraps — 19 320 x 8 x 12 x 4 = 7.4 M < 2^23 colliding corner
  // contributions of weight <= 1 each would need every sample of a frame-layer to hit ONE pixel with full weight to get there.
  const float bound = __uint_as_float(absmax[0]) * __uint_as_float(absmax[1]);
  const bool finite = bound < 3.0e38f;      // false for Inf and NaN
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    grad_value[i] = finite ? (float)((double)(long long)acc[i] * (double)inv) : __builtin_nanf("");
}

}  // namespace

DVIS_EXPORT int64_t dvis_msda_backward_det_ws_bytes(int N, int S, int M, int D) {
  if (N < 0 || S <= 0 || M <= 0 || D <= 0) return -1;
  return (int64_t)N * S * M * D * 8 + 16;
}

// fp32 backward with a run-to-run REPRODUCIBLE grad_value (see DetArgs); grad_loc / grad_w are per-sample reductions in a fixed
// order in both forms.  ws: dvis_msda_backward_det_ws_bytes bytes, 16-byte aligned.
DVIS_EXPORT int dvis_msda_backward_det(const float *value, const int64_t *shapes, const int64_t *level_start, const float *loc,
                                       const float *w, const float *grad_out, int N, int S, int M, int D, int L, int Lq, int P,
                                       float *grad_value, float *grad_loc, float *grad_w, void *ws, void *stream) {
  DVIS_REQUIRE(N >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_backward_det: bad sizes");
  DVIS_REQUIRE(value && shapes && level_start && loc && w && grad_out && grad_value && grad_loc && grad_w && ws &&
                   (uintptr_t)ws % 16 == 0, "msda_backward_det: null / misaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  const size_t nval = (size_t)N * S * M * D;
  if (N == 0) return DVIS_OK;
  if (Lq == 0) return dvis_zero_words(grad_value, nval, st, "msda_backward_det: zero grad_value");
  unsigned *absmax = (unsigned *)ws;
  unsigned long long *acc = (unsigned long long *)((char *)ws + 16);
  // (a kernel, not hipMemsetAsync: a memset node of a hipGraph fills with garbage from its second replay on, dvis_common.h)
  if (const int rc = dvis_zero_words(ws, 4 + nval * 2, st, "msda_backward_det: zero workspace")) return rc;
  const size_t ngo = (size_t)N * Lq * M * D, nw = (size_t)N * Lq * M * L * P;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>((ngo + 255) / 256, 2048)), dim3(256), 0, st, grad_out, ngo, absmax);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<size_t>((nw + 255) / 256, 2048)), dim3(256), 0, st, w, nw, absmax + 1);
  const size_t npairs = (size_t)N * Lq * M;
  const DetArgs det = {acc, absmax};
  if (D <= 32) {
    const size_t blocks = (npairs + 7) / 8;
    hipLaunchKernelGGL((msda_bwd_kernel<float, 32, true>), dim3((unsigned)(blocks > 262144 ? 262144 : blocks)), dim3(256), 0, st, value,
                       shapes, level_start, loc, w, grad_out, npairs, S, M, D, L, Lq, P, grad_value, grad_loc, grad_w, det);
  } else {
    const size_t blocks = (npairs + 3) / 4;
    hipLaunchKernelGGL((msda_bwd_kernel<float, 64, true>), dim3((unsigned)(blocks > 262144 ? 262144 : blocks)), dim3(256), 0, st, value,
                       shapes, level_start, loc, w, grad_out, npairs, S, M, D, L, Lq, P, grad_value, grad_loc, grad_w, det);
  }
  if (const int rc = dvis_check_launch("msda_bwd_kernel (deterministic)")) return rc;
  hipLaunchKernelGGL(det_finish_kernel, dim3((unsigned)std::min<size_t>((nval + 255) / 256, 65536)), dim3(256), 0, st, acc, absmax, grad_value,
                     nval);
  return dvis_check_launch("det_finish_kernel");
}

DVIS_EXPORT int dvis_msda_backward(int dtype, const void *value, const int64_t *shapes, const int64_t *level_start,
                                   const void *loc, const void *w, const void *grad_out, int N, int S, int M, int D,
                                   int L, int Lq, int P, void *grad_value, void *grad_loc, void *grad_w,
                                   void *stream) {
  DVIS_REQUIRE(N >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_backward: bad sizes");
  if (N == 0 || Lq == 0) return DVIS_OK;
  DVIS_REQUIRE(value && shapes && level_start && loc && w && grad_out && grad_value && grad_loc && grad_w,
               "msda_backward: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DVIS_F32)
    return launch_bwd<float>(value, shapes, level_start, loc, w, grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                             grad_w, st);
  if (dtype == DVIS_F64)
    return launch_bwd<double>(value, shapes, level_start, loc, w, grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                              grad_w, st);
  dvis_set_error("msda_backward: only fp32 / fp64 gradients are supported (got dtype %d)", dtype);
  return DVIS_E_UNSUPPORTED;
}
