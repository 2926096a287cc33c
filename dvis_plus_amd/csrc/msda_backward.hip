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
#include "dvis_common.h"

namespace {

template <typename A, int GS>
__device__ __forceinline__ A group_sum(A v) {
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, GS);
  return v;
}

template <typename T, int GS>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const T *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const T *__restrict__ loc, const T *__restrict__ w, const T *__restrict__ grad_out, size_t npairs, int S, int M,
    int D, int L, int Lq, int P, T *__restrict__ grad_value, T *__restrict__ grad_loc, T *__restrict__ grad_w) {
  constexpr int GPB = 256 / GS;
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
            if (ok1) atomicAdd(grad_value + i1 + c, ga * hh * hw);
            if (ok2) atomicAdd(grad_value + i2 + c, ga * hh * lw);
            if (ok3) atomicAdd(grad_value + i3 + c, ga * lh * hw);
            if (ok4) atomicAdd(grad_value + i4 + c, ga * lh * lw);
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

// fp32, D a multiple of 4 with D / 4 a power of two <= 16 (D = 32: the pixel decoder, D = 64: the ViT-Adapter extractors
// at 16 heads x 64): a group of G = D / 4 lanes owns one (n, q, m) pair and each lane FOUR channels — 16-byte loads of the
// four corners and of grad_out as in the forward kernel (a corner of a head is one 128-byte line at D = 32, read by 8 lanes
// with one request each instead of 32 scalar ones), the three scalar gradients reduced over the G lanes with shuffles, and
// the loc / weight scalars of the pair's L * P samples fetched ONCE per group as 16-byte pieces and broadcast with shuffles.
// grad_value keeps hardware fp32 atomics (global_atomic_add_f32), four per lane and corner: neighbouring queries hit the
// same lines, so most of them meet in the L2.
template <int G>
__global__ __launch_bounds__(256) void msda_bwd_vec_kernel(
    const float *__restrict__ value, const int64_t *__restrict__ shapes, const int64_t *__restrict__ level_start,
    const float *__restrict__ loc, const float *__restrict__ w, const float *__restrict__ grad_out, size_t npairs, int S,
    int M, int L, int Lq, int P, float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_w) {
  constexpr int D = 4 * G, GPB = 256 / G;
  const int gl = threadIdx.x % G;
  const size_t pix = (size_t)M * D;
  const int LP = L * P;
  for (size_t pair = (size_t)blockIdx.x * GPB + threadIdx.x / G; pair < npairs; pair += (size_t)gridDim.x * GPB) {
    const int m = (int)(pair % M);
    const size_t n = pair / ((size_t)Lq * M);
    const float4 g4 = *reinterpret_cast<const float4 *>(grad_out + pair * D + 4 * gl);
    const float *lp = loc + pair * (size_t)(2 * LP), *wp = w + pair * (size_t)LP;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const size_t lbase = ((n * S + (size_t)level_start[l]) * M + m) * D + 4 * gl;
      for (int p = 0; p < P; ++p) {
        const int sidx = l * P + p;
        const float x = lp[2 * sidx], y = lp[2 * sidx + 1], aw = wp[sidx];   // (same address in the G lanes: one request)
        const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
        float gw = 0.f, gx = 0.f, gy = 0.f;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {   // group-uniform branch
          const float hf = floorf(h_im), wf = floorf(w_im);
          const int h0 = (int)hf, w0 = (int)wf;
          const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
          const bool ok1 = h0 >= 0 && w0 >= 0, ok2 = h0 >= 0 && w0 + 1 <= W - 1;
          const bool ok3 = h0 + 1 <= H - 1 && w0 >= 0, ok4 = h0 + 1 <= H - 1 && w0 + 1 <= W - 1;
          const long long i1 = (long long)lbase + ((long long)h0 * W + w0) * (long long)pix;
          const long long i2 = i1 + (long long)pix, i3 = i1 + (long long)W * (long long)pix, i4 = i3 + (long long)pix;
          const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 a = ok1 ? *reinterpret_cast<const float4 *>(value + i1) : z;
          const float4 b = ok2 ? *reinterpret_cast<const float4 *>(value + i2) : z;
          const float4 e = ok3 ? *reinterpret_cast<const float4 *>(value + i3) : z;
          const float4 f = ok4 ? *reinterpret_cast<const float4 *>(value + i4) : z;
          const float c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;
          const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
          const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
          const float ev[4] = {e.x, e.y, e.z, e.w}, fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float g = gv[c];
            gw += g * (c1 * av[c] + c2 * bv[c] + c3 * ev[c] + c4 * fv[c]);
            gx += g * (hh * (bv[c] - av[c]) + lh * (fv[c] - ev[c]));
            gy += g * (hw * (ev[c] - av[c]) + lw * (fv[c] - bv[c]));
            const float ga = g * aw;
            if (ok1) atomicAdd(grad_value + i1 + c, ga * c1);
            if (ok2) atomicAdd(grad_value + i2 + c, ga * c2);
            if (ok3) atomicAdd(grad_value + i3 + c, ga * c3);
            if (ok4) atomicAdd(grad_value + i4 + c, ga * c4);
          }
        }
        gw = group_sum<float, G>(gw);
        gx = group_sum<float, G>(gx);
        gy = group_sum<float, G>(gy);
        if (gl == 0) {
          const size_t s = pair * (size_t)LP + sidx;
          grad_w[s] = gw;
          grad_loc[2 * s] = gx * aw * (float)W;
          grad_loc[2 * s + 1] = gy * aw * (float)H;
        }
      }
    }
  }
}

template <int G>
int launch_bwd_vec(const void *value, const int64_t *shapes, const int64_t *ls, const void *loc, const void *w,
                   const void *go, int N, int S, int M, int L, int Lq, int P, void *gv, void *gl, void *gw, hipStream_t st) {
  const size_t npairs = (size_t)N * Lq * M;
  const size_t blocks = (npairs + 256 / G - 1) / (256 / G);
  hipLaunchKernelGGL((msda_bwd_vec_kernel<G>), dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(256), 0, st,
                     (const float *)value, shapes, ls, (const float *)loc, (const float *)w, (const float *)go, npairs, S, M, L,
                     Lq, P, (float *)gv, (float *)gl, (float *)gw);
  return dvis_check_launch("msda_bwd_vec_kernel");
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

}  // namespace

DVIS_EXPORT int dvis_msda_backward(int dtype, const void *value, const int64_t *shapes, const int64_t *level_start,
                                   const void *loc, const void *w, const void *grad_out, int N, int S, int M, int D,
                                   int L, int Lq, int P, void *grad_value, void *grad_loc, void *grad_w,
                                   void *stream) {
  DVIS_REQUIRE(N >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_backward: bad sizes");
  if (N == 0 || Lq == 0) return DVIS_OK;
  DVIS_REQUIRE(value && shapes && level_start && loc && w && grad_out && grad_value && grad_loc && grad_w,
               "msda_backward: null pointer");
  hipStream_t st = (hipStream_t)stream;
  static const bool vec_on = []() { const char *e = getenv("DVIS_MSDA_BWD_VEC"); return !(e && e[0] == '0'); }();
  const bool aligned = (((uintptr_t)value | (uintptr_t)grad_out | (uintptr_t)grad_value) & 15) == 0;
  if (dtype == DVIS_F32 && vec_on && aligned && (D == 16 || D == 32 || D == 64)) {
    if (D == 16) return launch_bwd_vec<4>(value, shapes, level_start, loc, w, grad_out, N, S, M, L, Lq, P, grad_value, grad_loc, grad_w, st);
    if (D == 32) return launch_bwd_vec<8>(value, shapes, level_start, loc, w, grad_out, N, S, M, L, Lq, P, grad_value, grad_loc, grad_w, st);
    return launch_bwd_vec<16>(value, shapes, level_start, loc, w, grad_out, N, S, M, L, Lq, P, grad_value, grad_loc, grad_w, st);
  }
  if (dtype == DVIS_F32)
    return launch_bwd<float>(value, shapes, level_start, loc, w, grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                             grad_w, st);
  if (dtype == DVIS_F64)
    return launch_bwd<double>(value, shapes, level_start, loc, w, grad_out, N, S, M, D, L, Lq, P, grad_value, grad_loc,
                              grad_w, st);
  dvis_set_error("msda_backward: only fp32 / fp64 gradients are supported (got dtype %d)", dtype);
  return DVIS_E_UNSUPPORTED;
}
