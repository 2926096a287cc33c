// Panoptic post-processing of a clip in one pass — gfx950.
//
// Replaces the tensor-by-tensor sequence of inference_video_vps (dvis_Plus/meta_architecture.py:890-912):
//   cur_masks = F.interpolate(cur_masks, first_resize_size, bilinear)          # stride-4 logits -> padded input size
//   cur_masks = cur_masks[:, :, :img_h, :img_w].sigmoid()
//   cur_masks = F.interpolate(cur_masks, (out_h, out_w), bilinear)
//   cur_mask_ids = (cur_scores.view(-1,1,1,1) * cur_masks).argmax(0)
// and the three per-segment areas of the bookkeeping loop (:915-925: (ids == k).sum(), (masks[k] >= .5).sum(),
// ((ids == k) & (masks[k] >= .5)).sum(); one .item() host sync each in the reference).
// The reference materialises K' x T x H x W floats four times (2.2 GB each at K'=20, T=30, 720p) on the CPU; here one
// thread owns one output pixel, evaluates both bilinear stages per candidate on the fly from the stride-4 logits
// (141 MB, cache resident), keeps a running arg-max, and the areas are reduced per block in LDS then with one atomic
// per (block, segment).  HBM traffic: read 141 MB, write 5 bytes per output pixel.
//
// Every float that feeds an integer decision (arg-max, >= 0.5, > 0) is evaluated in the operation order of torch's CPU
// kernels — the device the reference post-processes on — see torch_cpu_math.h: the integer outputs then equal the
// reference's wherever its own result does not depend on its thread count.
#include <math.h>

#include "dvis_common.h"
#include "torch_cpu_math.h"

namespace {

constexpr int kMaxK = 256;
using tcpu::Geometry;
using tcpu::Tap;

__global__ __launch_bounds__(256) void vps_argmax_kernel(
    const float *__restrict__ logits, int64_t stride_k, int64_t stride_t, const float *__restrict__ scores, int K, int T,
    int h, int w, int first_h, int first_w, int img_h, int img_w, int out_h, int out_w, int *__restrict__ ids,
    uint8_t *__restrict__ conf, int *__restrict__ areas /* (3, K): mask_area, original_area, intersection */) {
  __shared__ int s_area[3 * kMaxK];
  for (int i = threadIdx.x; i < 3 * K; i += 256) s_area[i] = 0;
  __syncthreads();

  const size_t npix = (size_t)T * out_h * out_w;
  const Geometry g(h, w, first_h, first_w, img_h, img_w, out_h, out_w);
  const int lane = threadIdx.x & 63;
  for (size_t base = (size_t)blockIdx.x * 256; base < npix; base += (size_t)gridDim.x * 256) {   // wave-uniform trip count
    const size_t p = base + threadIdx.x;
    const bool valid = p < npix;
    const size_t pc = valid ? p : npix - 1;
    const int X = (int)(pc % out_w);
    const size_t r = pc / out_w;
    const int Y = (int)(r % out_h);
    const int t = (int)(r / out_h);
    const Tap ty = tcpu::make_tap(Y, g.s2y, img_h, out_h), tx = tcpu::make_tap(X, g.s2x, img_w, out_w);
    const int kind1 = tcpu::bilinear_kind(first_h, first_w, t, T), kind2 = tcpu::bilinear_kind(out_h, out_w, t, T);
    float best = -INFINITY, best_prob = 0.f;
    int best_k = 0;
    for (int k = 0; k < K; ++k) {
      const float *lg = logits + (size_t)k * stride_k + (size_t)t * stride_t;
      const float prob = tcpu::two_stage<true>(lg, g, Y, X, ty, tx, kind1, kind2);
      // original_area[k]: one LDS atomic per wave instead of one per pixel
      const unsigned long long over = __ballot(valid && prob >= 0.5f);
      if (lane == 0 && over) atomicAdd(&s_area[K + k], __popcll(over));
      const float sc = scores[k] * prob;
      if (sc > best) {     // strict: the first maximum wins, like argmax
        best = sc;
        best_k = k;
        best_prob = prob;
      }
    }
    if (valid) {
      const bool cf = best_prob >= 0.5f;
      ids[p] = best_k;
      conf[p] = cf ? 1 : 0;
      atomicAdd(&s_area[best_k], 1);
      if (cf) atomicAdd(&s_area[2 * K + best_k], 1);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * K; i += 256)
    if (s_area[i]) atomicAdd(&areas[i], s_area[i]);
}

// Semantic arg-max of a clip in one pass (inference_video_vss, dvis_Plus/meta_architecture.py:954-979):
//   prob_q = resize2(sigmoid(resize1(logits_q)[:img_h, :img_w]));  sem[c] = sum_q cls[q][c] * prob_q;  out = argmax_c sem
// The reference materialises (Q, T, H, W) twice and (C, T, H, W) once (14 GB at Q=100, C=124, T=30, 720p).  One thread
// owns one output pixel and keeps the C class sums in registers (C <= 4*K4); the class scores of query q are the same
// for every lane, so they are wave-uniform loads (scalar cache -> SGPR operands of the FMAs): no LDS traffic and no
// vector registers spent on them.  `cls` is (Q, 4*K4), zero padded by the caller.
template <int K4>
__global__ __launch_bounds__(256) void vss_argmax_kernel(
    const float *__restrict__ logits, int64_t stride_q, int64_t stride_t, const float *__restrict__ cls, int Q, int C, int T,
    int h, int w, int first_h, int first_w, int img_h, int img_w, int out_h, int out_w, int64_t *__restrict__ out) {
  const size_t npix = (size_t)T * out_h * out_w;
  const Geometry g(h, w, first_h, first_w, img_h, img_w, out_h, out_w);
  const bool identity2 = img_h == out_h && img_w == out_w;
  for (size_t base = (size_t)blockIdx.x * 256; base < npix; base += (size_t)gridDim.x * 256) {
    const size_t p = base + threadIdx.x;
    const bool valid = p < npix;
    const size_t pc = valid ? p : npix - 1;
    const int X = (int)(pc % out_w);
    const size_t r = pc / out_w;
    const int Y = (int)(r % out_h);
    const int t = (int)(r / out_h);
    const Tap ty = tcpu::make_tap(Y, g.s2y, img_h, out_h), tx = tcpu::make_tap(X, g.s2x, img_w, out_w);
    const int kind1 = tcpu::bilinear_kind(first_h, first_w, t, T), kind2 = tcpu::bilinear_kind(out_h, out_w, t, T);
    float4 acc[K4];
#pragma unroll
    for (int k = 0; k < K4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto accumulate = [&](int q, float prob) {
      const float4 *row = reinterpret_cast<const float4 *>(cls + (size_t)q * 4 * K4);    // uniform address
#pragma unroll
      for (int k = 0; k < K4; ++k) {
        const float4 cv = row[k];
        acc[k].x += cv.x * prob; acc[k].y += cv.y * prob; acc[k].z += cv.z * prob; acc[k].w += cv.w * prob;
      }
    };
    if (identity2) {
      // same size: the second resize is the identity.  The 4 first-stage taps do not depend on q: indices once, and the
      // 4 logits of query q+1 are loaded while query q's class sums are accumulated (the loop is latency-bound otherwise:
      // 128 accumulators leave 2 waves per SIMD).
      const Tap sy = tcpu::make_tap(Y, g.s1y, h, first_h), sx = tcpu::make_tap(X, g.s1x, w, first_w);
      const int o00 = sy.i0 * w + sx.i0, o01 = sy.i0 * w + sx.i1, o10 = sy.i1 * w + sx.i0, o11 = sy.i1 * w + sx.i1;
      const float *lg = logits + (size_t)t * stride_t;
      const bool tail = X >= g.tail_x;
      float a = lg[o00], b = lg[o01], c = lg[o10], d = lg[o11];
#pragma unroll 1
      for (int q = 0; q < Q; ++q) {
        const float v = tcpu::bilinear(a, b, c, d, sy, sx, kind1);
        if (q + 1 < Q) {
          const float *nx = lg + (size_t)(q + 1) * stride_q;
          a = nx[o00]; b = nx[o01]; c = nx[o10]; d = nx[o11];
        }
        accumulate(q, tcpu::sigmoid(v, tail));
      }
    } else {
#pragma unroll 1
      for (int q = 0; q < Q; ++q) {
        const float *lg = logits + (size_t)q * stride_q + (size_t)t * stride_t;
        accumulate(q, tcpu::two_stage<true>(lg, g, Y, X, ty, tx, kind1, kind2));
      }
    }
    float best = -INFINITY;
    int best_c = 0;
#pragma unroll
    for (int k = 0; k < K4; ++k) {      // ascending, strict: the first maximum wins, like torch.max(0)
      if (4 * k < C && acc[k].x > best) { best = acc[k].x; best_c = 4 * k; }
      if (4 * k + 1 < C && acc[k].y > best) { best = acc[k].y; best_c = 4 * k + 1; }
      if (4 * k + 2 < C && acc[k].z > best) { best = acc[k].z; best_c = 4 * k + 2; }
      if (4 * k + 3 < C && acc[k].w > best) { best = acc[k].w; best_c = 4 * k + 3; }
    }
    if (valid) out[p] = best_c;
  }
}

// MFMA form of the semantic arg-max for the common case (second resize = identity, C <= 128): the class sums are a
// GEMM  sem[c][px] = sum_q cls[q][c] * prob[q][px]  (C x Q) x (Q x pixels).  A wave owns 64 consecutive output pixels
// (4 tiles of 16) x 128 classes (8 tiles of 16) = 32 accumulator tiles of v_mfma_f32_16x16x4_f32; per k-step of 4
// queries every lane evaluates the 4 probabilities it contributes as B operand (pixel = lane & 15 of each tile, query =
// 4*ks + (lane >> 4)) and reads its A operand (class = lane & 15, same query) from the zero-padded class matrix in LDS.
// 124 classes x 100 queries at 720p: 0.69 TFLOP per clip = 4.4 ms at the fp32-MFMA peak, vs 12 ms of VALU work for the
// one-pixel-per-thread form above (measured 16.7 ms).
__global__ __launch_bounds__(256) void vss_argmax_mfma_kernel(
    const float *__restrict__ logits, int64_t stride_q, int64_t stride_t, const float *__restrict__ cls /* (Qp, 128) */,
    int Q, int Qp, int C, int T, int h, int w, int first_h, int first_w, int out_h, int out_w, int64_t *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_cls[];      // [Qp][128 + 4]: +4 spreads the 4 query rows of an A read over banks
  constexpr int LSC = 132;
  constexpr int NT = 2;                                              // 16-pixel tiles per wave: 8 x NT accumulator tiles
  for (int i = threadIdx.x; i < Qp * 128; i += 256) s_cls[(i >> 7) * LSC + (i & 127)] = cls[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, gq = lane >> 4;
  const size_t npix = (size_t)T * out_h * out_w;
  const Geometry g(h, w, first_h, first_w, out_h, out_w, out_h, out_w);
  const size_t nchunks = (npix + 16 * NT - 1) / (16 * NT);
  for (size_t chunk = (size_t)blockIdx.x * 4 + wv; chunk < nchunks; chunk += (size_t)gridDim.x * 4) {
    // this lane's NT pixels (one per 16-pixel tile) and their first-stage taps
    const float *l00[NT], *l01[NT], *l10[NT], *l11[NT];
    Tap ty1[NT], tx1[NT];
    int kind1[NT];
    bool tail[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const size_t p = min(chunk * (16 * NT) + (size_t)n * 16 + j, npix - 1);
      const int X = (int)(p % out_w);
      const size_t r = p / out_w;
      const int Y = (int)(r % out_h);
      const Tap sy = tcpu::make_tap(Y, g.s1y, h, first_h), sx = tcpu::make_tap(X, g.s1x, w, first_w);
      const float *base = logits + (size_t)(r / out_h) * stride_t + (size_t)gq * stride_q;   // query gq of k-step 0
      l00[n] = base + sy.i0 * w + sx.i0; l01[n] = base + sy.i0 * w + sx.i1;
      l10[n] = base + sy.i1 * w + sx.i0; l11[n] = base + sy.i1 * w + sx.i1;
      ty1[n] = sy; tx1[n] = sx;
      kind1[n] = tcpu::bilinear_kind(first_h, first_w, (int)(r / out_h), T);
      tail[n] = X >= g.tail_x;
    }
    dvis_f4 acc[8][NT];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[mt][n] = dvis_f4{0.f, 0.f, 0.f, 0.f};
    // logits of the lane's query of the NEXT k-step are loaded while this k-step's MFMAs run (Qp % 4 == 0, Qp <= Q + 3:
    // the caller guarantees Qp == Q, so every (ks + gq) row exists)
    float va[NT], vb[NT], vc[NT], vd[NT];
    const size_t step = (size_t)4 * stride_q;
#pragma unroll
    for (int n = 0; n < NT; ++n) { va[n] = *l00[n]; vb[n] = *l01[n]; vc[n] = *l10[n]; vd[n] = *l11[n]; }
#pragma unroll 1
    for (int ks = 0; ks < Qp; ks += 4) {
      float pb[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        pb[n] = tcpu::sigmoid(tcpu::bilinear(va[n], vb[n], vc[n], vd[n], ty1[n], tx1[n], kind1[n]), tail[n]);
      }
      if (ks + 4 < Qp) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          l00[n] += step; l01[n] += step; l10[n] += step; l11[n] += step;
          va[n] = *l00[n]; vb[n] = *l01[n]; vc[n] = *l10[n]; vd[n] = *l11[n];
        }
      }
      const float *arow = s_cls + (size_t)(ks + gq) * LSC + j;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const float a = arow[16 * mt];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[mt][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pb[n], acc[mt][n], 0, 0, 0);
      }
    }
    // acc[mt][n][r] = sem[class 16*mt + 4*gq + r][pixel tile n, column j]: arg-max over classes, first maximum wins
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      float best = -INFINITY;
      int best_c = 0x7fffffff;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * mt + 4 * gq + r;
          const float v = acc[mt][n][r];
          if (c < C && v > best) { best = v; best_c = c; }      // ascending within the lane
        }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float ov = __shfl_xor(best, o);
        const int oc = __shfl_xor(best_c, o);
        if (ov > best || (ov == best && oc < best_c)) { best = ov; best_c = oc; }
      }
      const size_t p = chunk * (16 * NT) + (size_t)n * 16 + j;
      if (gq == 0 && p < npix) out[p] = best_c;
    }
  }
}

// Instance masks (inference_video_vis, dvis_Plus/meta_architecture.py:843-853; MinVIS :390-399):
//   masks = resize2(resize1(logits)[:img_h, :img_w]) > 0      — no sigmoid between the stages.
// The reference materialises (K', T, first_h, first_w) and (K', T, H, W) floats; here one thread owns 4 consecutive
// output pixels of a row and writes them as one 32-bit word of the bool tensor.
__global__ __launch_bounds__(256) void resize2_gt0_kernel(
    const float *__restrict__ logits, int64_t stride_k, int64_t stride_t, int K, int T, int h, int w, int first_h,
    int first_w, int img_h, int img_w, int out_h, int out_w, uint8_t *__restrict__ out) {
  const Geometry g(h, w, first_h, first_w, img_h, img_w, out_h, out_w);
  const int wq = (out_w + 3) >> 2;                                   // 4-pixel groups per row
  const size_t ngroups = (size_t)K * T * out_h * wq;
  const bool word_ok = (out_w & 3) == 0;
  for (size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x; gi < ngroups; gi += (size_t)gridDim.x * 256) {
    const int xg = (int)(gi % wq);
    size_t r = gi / wq;
    const int Y = (int)(r % out_h);
    r /= out_h;
    const int t = (int)(r % T);
    const int k = (int)(r / T);
    const float *lg = logits + (size_t)k * stride_k + (size_t)t * stride_t;
    const Tap ty = tcpu::make_tap(Y, g.s2y, img_h, out_h);
    const int kind1 = tcpu::bilinear_kind(first_h, first_w, t, T), kind2 = tcpu::bilinear_kind(out_h, out_w, t, T);
    uint8_t *orow = out + (((size_t)k * T + t) * out_h + Y) * out_w;
    unsigned word = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int X = 4 * xg + i;
      if (X < out_w) {
        const Tap tx = tcpu::make_tap(X, g.s2x, img_w, out_w);
        const bool on = tcpu::two_stage<false>(lg, g, Y, X, ty, tx, kind1, kind2) > 0.f;
        if (word_ok) word |= (on ? 1u : 0u) << (8 * i);
        else orow[X] = on ? 1 : 0;
      }
    }
    if (word_ok) *reinterpret_cast<unsigned *>(orow + 4 * xg) = word;
  }
}

// The float values behind the integer decisions above: out = resize2(maybe_sigmoid(resize1(logits)[:img_h, :img_w])).
// Exists so that the operation order can be pinned bit for bit against torch's CPU ops (tests), and as the fused form of
// the reference's interpolate -> crop -> (sigmoid) -> interpolate sequence for callers that want the values.
template <bool SIGMOID>
__global__ __launch_bounds__(256) void resize2_kernel(
    const float *__restrict__ logits, int64_t stride_k, int64_t stride_t, int K, int T, int h, int w, int first_h,
    int first_w, int img_h, int img_w, int out_h, int out_w, float *__restrict__ out) {
  const Geometry g(h, w, first_h, first_w, img_h, img_w, out_h, out_w);
  const size_t npix = (size_t)K * T * out_h * out_w;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    const int X = (int)(p % out_w);
    size_t r = p / out_w;
    const int Y = (int)(r % out_h);
    r /= out_h;
    const int t = (int)(r % T);
    const int k = (int)(r / T);
    const Tap ty = tcpu::make_tap(Y, g.s2y, img_h, out_h), tx = tcpu::make_tap(X, g.s2x, img_w, out_w);
    const int kind1 = tcpu::bilinear_kind(first_h, first_w, t, T), kind2 = tcpu::bilinear_kind(out_h, out_w, t, T);
    out[p] = tcpu::two_stage<SIGMOID>(logits + (size_t)k * stride_k + (size_t)t * stride_t, g, Y, X, ty, tx, kind1, kind2);
  }
}

}  // namespace

DVIS_EXPORT int dvis_vps_argmax(const float *logits, int64_t stride_k, int64_t stride_t, const float *scores, int K, int T,
                                int h, int w, int first_h, int first_w, int img_h, int img_w, int out_h, int out_w,
                                int32_t *ids, uint8_t *conf, int32_t *areas, void *stream) {
  DVIS_REQUIRE(K > 0 && K <= kMaxK && T >= 0 && h > 0 && w > 0 && first_h > 0 && first_w > 0 && img_h > 0 && img_w > 0 &&
                   out_h > 0 && out_w > 0,
               "vps_argmax: bad sizes (K must be 1..%d)", kMaxK);
  DVIS_REQUIRE(img_h <= first_h && img_w <= first_w, "vps_argmax: image size exceeds the padded size");
  DVIS_REQUIRE((size_t)T * out_h * out_w < 0x7fffffffu, "vps_argmax: more than 2^31 output pixels");
  DVIS_REQUIRE(logits && scores && ids && conf && areas, "vps_argmax: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (const int rc = dvis_zero_words(areas, (size_t)3 * K, st, "vps_argmax: zero areas")) return rc;      // (a kernel, not a memset node)
  if (T == 0) return DVIS_OK;
  const size_t npix = (size_t)T * out_h * out_w;
  size_t blocks = (npix + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;     // grid-stride: bounds the number of global area atomics
  hipLaunchKernelGGL(vps_argmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, logits, stride_k, stride_t, scores, K, T,
                     h, w, first_h, first_w, img_h, img_w, out_h, out_w, ids, conf, areas);
  return dvis_check_launch("vps_argmax_kernel");
}

DVIS_EXPORT int dvis_vss_argmax(const float *logits, int64_t stride_q, int64_t stride_t, const float *cls,
                                int cls_row_stride, int Q, int C, int T, int h, int w, int first_h, int first_w, int img_h, int img_w, int out_h, int out_w,
                                int64_t *out, void *stream) {
  DVIS_REQUIRE(Q > 0 && C > 0 && C <= 128 && T >= 0 && h > 0 && w > 0 && first_h > 0 && first_w > 0 && img_h > 0 &&
                   img_w > 0 && out_h > 0 && out_w > 0,
               "vss_argmax: bad sizes (C must be 1..128)");
  DVIS_REQUIRE(img_h <= first_h && img_w <= first_w, "vss_argmax: image size exceeds the padded size");
  DVIS_REQUIRE(logits && cls && out, "vss_argmax: null pointer");
  if (T == 0) return DVIS_OK;
  DVIS_REQUIRE(cls_row_stride >= C && cls_row_stride % 32 == 0 && cls_row_stride <= 128 && (((uintptr_t)cls) & 15) == 0,
               "vss_argmax: cls rows must be zero padded to a multiple of 32 floats (<= 128), 16-byte aligned");
  const int k4 = cls_row_stride / 4;              // 8, 16, 24 or 32 float4 of class sums per thread
  hipStream_t st = (hipStream_t)stream;
  const size_t npix = (size_t)T * out_h * out_w;
  size_t blocks = (npix + 255) / 256;
  if (img_h == out_h && img_w == out_w && cls_row_stride == 128 && Q % 4 == 0) {
    // GEMM form on the matrix cores (cls is padded to 128 columns; Q % 4 == 0 so that no row past the matrix is read)
    const size_t lds = (size_t)Q * 132 * sizeof(float);
    if (lds <= 64 * 1024) {
      size_t nb = ((npix + 31) / 32 + 3) / 4;
      if (nb > 256 * 9) nb = 256 * 9;
      hipLaunchKernelGGL(vss_argmax_mfma_kernel, dim3((unsigned)nb), dim3(256), lds, st, logits, stride_q, stride_t, cls, Q, Q,
                         C, T, h, w, first_h, first_w, out_h, out_w, out);
      return dvis_check_launch("vss_argmax_mfma_kernel");
    }
  }
  if (blocks > 256 * 16) blocks = 256 * 16;
#define DVIS_VSS(K4_)                                                                                                   \
  hipLaunchKernelGGL((vss_argmax_kernel<K4_>), dim3((unsigned)blocks), dim3(256), 0, st, logits, stride_q, stride_t, cls, \
                     Q, C, T, h, w, first_h, first_w, img_h, img_w, out_h, out_w, out)
  if (k4 == 8) DVIS_VSS(8);
  else if (k4 == 16) DVIS_VSS(16);
  else if (k4 == 24) DVIS_VSS(24);
  else DVIS_VSS(32);
#undef DVIS_VSS
  return dvis_check_launch("vss_argmax_kernel");
}

DVIS_EXPORT int dvis_resize2_gt0(const float *logits, int64_t stride_k, int64_t stride_t, int K, int T, int h, int w,
                                 int first_h, int first_w, int img_h, int img_w, int out_h, int out_w, uint8_t *out,
                                 void *stream) {
  DVIS_REQUIRE(K >= 0 && T >= 0 && h > 0 && w > 0 && first_h > 0 && first_w > 0 && img_h > 0 && img_w > 0 && out_h > 0 &&
                   out_w > 0,
               "resize2_gt0: bad sizes");
  DVIS_REQUIRE(img_h <= first_h && img_w <= first_w, "resize2_gt0: image size exceeds the padded size");
  if (K == 0 || T == 0) return DVIS_OK;
  DVIS_REQUIRE(logits && out, "resize2_gt0: null pointer");
  DVIS_REQUIRE((((uintptr_t)out) & 3) == 0, "resize2_gt0: out must be 4-byte aligned");
  const size_t ngroups = (size_t)K * T * out_h * ((out_w + 3) / 4);
  size_t blocks = (ngroups + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(resize2_gt0_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, stride_k,
                     stride_t, K, T, h, w, first_h, first_w, img_h, img_w, out_h, out_w, out);
  return dvis_check_launch("resize2_gt0_kernel");
}

DVIS_EXPORT int dvis_resize2(const float *logits, int64_t stride_k, int64_t stride_t, int K, int T, int h, int w,
                             int first_h, int first_w, int img_h, int img_w, int out_h, int out_w, int sigmoid, float *out,
                             void *stream) {
  DVIS_REQUIRE(K >= 0 && T >= 0 && h > 0 && w > 0 && first_h > 0 && first_w > 0 && img_h > 0 && img_w > 0 && out_h > 0 &&
                   out_w > 0,
               "resize2: bad sizes");
  DVIS_REQUIRE(img_h <= first_h && img_w <= first_w, "resize2: image size exceeds the padded size");
  if (K == 0 || T == 0) return DVIS_OK;
  DVIS_REQUIRE(logits && out, "resize2: null pointer");
  const size_t npix = (size_t)K * T * out_h * out_w;
  size_t blocks = (npix + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (sigmoid)
    hipLaunchKernelGGL(resize2_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, stride_k,
                       stride_t, K, T, h, w, first_h, first_w, img_h, img_w, out_h, out_w, out);
  else
    hipLaunchKernelGGL(resize2_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, stride_k,
                       stride_t, K, T, h, w, first_h, first_w, img_h, img_w, out_h, out_w, out);
  return dvis_check_launch("resize2_kernel");
}
