/*
 * ORACLE — test infrastructure, not product code.
 *
 * Plain-C CPU restatement of the reference's multi-scale deformable attention
 * (forward + backward).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it; the product path never does.
 *
 * Follows, per function (paths under /root/reference/DVIS_Plus/mask2former/modeling/pixel_decoder/ops/):
 *   sample validity + pixel mapping   src/cuda/ms_deform_im2col_cuda.cuh:286-293
 *   bilinear read, zero padding       src/cuda/ms_deform_im2col_cuda.cuh:38-89
 *   forward accumulation              src/cuda/ms_deform_im2col_cuda.cuh:242-304
 *   backward (grad_value, grad_loc, grad_w)   src/cuda/ms_deform_im2col_cuda.cuh:92-164, 306-408
 *   == F.grid_sample(2*loc-1, bilinear, zeros, align_corners=False) weighted sum,
 *      functions/ms_deform_attn_func.py:52-72 (the CPU path BASELINE.json names as parity target).
 * Pinned by tests/test_oracle.py against tests/golden/g1_msda_*.npz (generated from the imported
 * reference by tests/golden/gen_golden.py).
 *
 * Layouts: value (N,S,M,D), shapes (L,2)=[H,W] int64, level_start (L,) int64,
 * loc (N,Lq,M,L,P,2) normalised (x,y), w (N,Lq,M,L,P), out (N,Lq,M*D).
 * Parallelised over (n,q) with OpenMP when compiled with -fopenmp (cpu_baseline uses that).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_MSDA(T, SUFFIX)                                                                                    \
  void msda_oracle_forward_##SUFFIX(const T *value, const int64_t *shapes, const int64_t *level_start,            \
                                    const T *loc, const T *w, int N, int S, int M, int D, int L, int Lq, int P,   \
                                    T *out) {                                                                     \
    const int64_t pix = (int64_t)M * D;                                                                           \
    _Pragma("omp parallel for collapse(2) schedule(static)") for (int n = 0; n < N; ++n) for (int q = 0; q < Lq;  \
                                                                                               ++q) {            \
      for (int m = 0; m < M; ++m) {                                                                               \
        const int64_t pair = ((int64_t)n * Lq + q) * M + m;                                                       \
        T *o = out + pair * D;                                                                                    \
        for (int c = 0; c < D; ++c) o[c] = 0;                                                                     \
        for (int l = 0; l < L; ++l) {                                                                             \
          const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                                           \
          const T *vb = value + (((int64_t)n * S + level_start[l]) * M + m) * D;                                  \
          for (int p = 0; p < P; ++p) {                                                                           \
            const int64_t s = pair * L * P + (int64_t)l * P + p;                                                  \
            const T x = loc[2 * s], y = loc[2 * s + 1], aw = w[s];                                                \
            const T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;                                                 \
            if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                                      \
            const T hf = floor(h_im), wf = floor(w_im);                                                           \
            const int h0 = (int)hf, w0 = (int)wf;                                                                 \
            const T lh = h_im - hf, lw = w_im - wf, hh = 1 - lh, hw = 1 - lw;                                     \
            const T *v1 = (h0 >= 0 && w0 >= 0) ? vb + ((int64_t)h0 * W + w0) * pix : 0;                           \
            const T *v2 = (h0 >= 0 && w0 + 1 <= W - 1) ? vb + ((int64_t)h0 * W + w0 + 1) * pix : 0;              \
            const T *v3 = (h0 + 1 <= H - 1 && w0 >= 0) ? vb + ((int64_t)(h0 + 1) * W + w0) * pix : 0;             \
            const T *v4 = (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) ? vb + ((int64_t)(h0 + 1) * W + w0 + 1) * pix : 0; \
            const T c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;                                       \
            for (int c = 0; c < D; ++c) {                                                                         \
              const T a = v1 ? v1[c] : 0, b = v2 ? v2[c] : 0, e = v3 ? v3[c] : 0, f = v4 ? v4[c] : 0;             \
              o[c] += (c1 * a + c2 * b + c3 * e + c4 * f) * aw;                                                   \
            }                                                                                                     \
          }                                                                                                       \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }                                                                                                               \
                                                                                                                  \
  /* grad_value must be zero-filled by the caller; serial (deterministic summation order). */                   \
  void msda_oracle_backward_##SUFFIX(const T *value, const int64_t *shapes, const int64_t *level_start,           \
                                     const T *loc, const T *w, const T *grad_out, int N, int S, int M, int D,     \
                                     int L, int Lq, int P, T *grad_value, T *grad_loc, T *grad_w) {               \
    const int64_t pix = (int64_t)M * D;                                                                           \
    for (int n = 0; n < N; ++n)                                                                                   \
      for (int q = 0; q < Lq; ++q)                                                                                \
        for (int m = 0; m < M; ++m) {                                                                             \
          const int64_t pair = ((int64_t)n * Lq + q) * M + m;                                                     \
          const T *go = grad_out + pair * D;                                                                      \
          for (int l = 0; l < L; ++l) {                                                                           \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                                         \
            const int64_t lbase = (((int64_t)n * S + level_start[l]) * M + m) * D;                                \
            for (int p = 0; p < P; ++p) {                                                                         \
              const int64_t s = pair * L * P + (int64_t)l * P + p;                                                \
              grad_loc[2 * s] = 0;                                                                                \
              grad_loc[2 * s + 1] = 0;                                                                            \
              grad_w[s] = 0;                                                                                      \
              const T x = loc[2 * s], y = loc[2 * s + 1], aw = w[s];                                              \
              const T h_im = y * H - (T)0.5, w_im = x * W - (T)0.5;                                               \
              if (!(h_im > -1 && w_im > -1 && h_im < H && w_im < W)) continue;                                    \
              const T hf = floor(h_im), wf = floor(w_im);                                                         \
              const int h0 = (int)hf, w0 = (int)wf;                                                               \
              const T lh = h_im - hf, lw = w_im - wf, hh = 1 - lh, hw = 1 - lw;                                   \
              const int ok1 = (h0 >= 0 && w0 >= 0), ok2 = (h0 >= 0 && w0 + 1 <= W - 1);                           \
              const int ok3 = (h0 + 1 <= H - 1 && w0 >= 0), ok4 = (h0 + 1 <= H - 1 && w0 + 1 <= W - 1);           \
              const int64_t i1 = lbase + ((int64_t)h0 * W + w0) * pix, i2 = i1 + pix;                             \
              const int64_t i3 = i1 + (int64_t)W * pix, i4 = i3 + pix;                                            \
              T gw = 0, gx = 0, gy = 0;                                                                           \
              for (int c = 0; c < D; ++c) {                                                                       \
                const T a = ok1 ? value[i1 + c] : 0, b = ok2 ? value[i2 + c] : 0;                                 \
                const T e = ok3 ? value[i3 + c] : 0, f = ok4 ? value[i4 + c] : 0;                                 \
                const T g = go[c];                                                                                \
                gw += g * (hh * hw * a + hh * lw * b + lh * hw * e + lh * lw * f);                                \
                gx += g * (hh * (b - a) + lh * (f - e));                                                          \
                gy += g * (hw * (e - a) + lw * (f - b));                                                          \
                const T ga = g * aw;                                                                              \
                if (ok1) grad_value[i1 + c] += ga * hh * hw;                                                      \
                if (ok2) grad_value[i2 + c] += ga * hh * lw;                                                      \
                if (ok3) grad_value[i3 + c] += ga * lh * hw;                                                      \
                if (ok4) grad_value[i4 + c] += ga * lh * lw;                                                      \
              }                                                                                                   \
              grad_w[s] = gw;                                                                                     \
              grad_loc[2 * s] = gx * aw * W;                                                                      \
              grad_loc[2 * s + 1] = gy * aw * H;                                                                  \
            }                                                                                                     \
          }                                                                                                       \
        }                                                                                                         \
  }

DEFINE_MSDA(float, f32)
DEFINE_MSDA(double, f64)
