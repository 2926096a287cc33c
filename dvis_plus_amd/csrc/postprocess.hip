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
// Bilinear taps follow torch's upsample_bilinear2d (align_corners=False):
//   src = max(scale * (dst + 0.5) - 0.5, 0), i0 = int(src), i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1,
//   val = h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d),  scale = in / out in fp32.
#include <math.h>

#include "dvis_common.h"

namespace {

constexpr int kMaxK = 256;

struct Tap {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Tap make_tap(int dst, float scale, int in) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  Tap t;
  t.i0 = min((int)src, in - 1);
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

// sigmoid(first-stage bilinear) at pixel (yy, xx) of the padded-size image
__device__ __forceinline__ float stage1_sigmoid(const float *__restrict__ lg, int h, int w, float sy, float sx, int yy,
                                                int xx) {
  const Tap ty = make_tap(yy, sy, h), tx = make_tap(xx, sx, w);
  const float a = lg[ty.i0 * w + tx.i0], b = lg[ty.i0 * w + tx.i1];
  const float c = lg[ty.i1 * w + tx.i0], d = lg[ty.i1 * w + tx.i1];
  const float v = ty.l0 * (tx.l0 * a + tx.l1 * b) + ty.l1 * (tx.l0 * c + tx.l1 * d);
  return 1.f / (1.f + expf(-v));
}

__global__ __launch_bounds__(256) void vps_argmax_kernel(
    const float *__restrict__ logits, int64_t stride_k, int64_t stride_t, const float *__restrict__ scores, int K, int T,
    int h, int w, int first_h, int first_w, int img_h, int img_w, int out_h, int out_w, int *__restrict__ ids,
    uint8_t *__restrict__ conf, int *__restrict__ areas /* (3, K): mask_area, original_area, intersection */) {
  __shared__ int s_area[3 * kMaxK];
  for (int i = threadIdx.x; i < 3 * K; i += 256) s_area[i] = 0;
  __syncthreads();

  const size_t npix = (size_t)T * out_h * out_w;
  const float s1y = (float)h / (float)first_h, s1x = (float)w / (float)first_w;
  const float s2y = (float)img_h / (float)out_h, s2x = (float)img_w / (float)out_w;
  const bool identity2 = img_h == out_h && img_w == out_w;
  const int lane = threadIdx.x & 63;
  for (size_t base = (size_t)blockIdx.x * 256; base < npix; base += (size_t)gridDim.x * 256) {   // wave-uniform trip count
    const size_t p = base + threadIdx.x;
    const bool valid = p < npix;
    const size_t pc = valid ? p : npix - 1;
    const int X = (int)(pc % out_w);
    const size_t r = pc / out_w;
    const int Y = (int)(r % out_h);
    const int t = (int)(r / out_h);
    const Tap ty = make_tap(Y, s2y, img_h), tx = make_tap(X, s2x, img_w);
    float best = -INFINITY, best_prob = 0.f;
    int best_k = 0;
    for (int k = 0; k < K; ++k) {
      const float *lg = logits + (size_t)k * stride_k + (size_t)t * stride_t;
      float prob;
      if (identity2) {     // same size: the second resize is the identity (taps (i, i) with weights (1, 0))
        prob = stage1_sigmoid(lg, h, w, s1y, s1x, Y, X);
      } else {
        const float a = stage1_sigmoid(lg, h, w, s1y, s1x, ty.i0, tx.i0), b = stage1_sigmoid(lg, h, w, s1y, s1x, ty.i0, tx.i1);
        const float c = stage1_sigmoid(lg, h, w, s1y, s1x, ty.i1, tx.i0), d = stage1_sigmoid(lg, h, w, s1y, s1x, ty.i1, tx.i1);
        prob = ty.l0 * (tx.l0 * a + tx.l1 * b) + ty.l1 * (tx.l0 * c + tx.l1 * d);
      }
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
  hipError_t e = hipMemsetAsync(areas, 0, (size_t)3 * K * sizeof(int32_t), st);
  if (e != hipSuccess) {
    dvis_set_error("vps_argmax: hipMemsetAsync: %s", hipGetErrorString(e));
    return DVIS_E_LAUNCH;
  }
  if (T == 0) return DVIS_OK;
  const size_t npix = (size_t)T * out_h * out_w;
  size_t blocks = (npix + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;     // grid-stride: bounds the number of global area atomics
  hipLaunchKernelGGL(vps_argmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, logits, stride_k, stride_t, scores, K, T,
                     h, w, first_h, first_w, img_h, img_w, out_h, out_w, ids, conf, areas);
  return dvis_check_launch("vps_argmax_kernel");
}
