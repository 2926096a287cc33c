// Shared device helpers of the MSDeformAttn forward kernels (msda_forward.hip, msda_forward_2d.hip).
#pragma once
#include "dvis_common.h"

namespace dvis_msda {

constexpr unsigned kOOB = 0x80000000u;  // buffer offset beyond every level slice (< 2 GiB, checked on host)

// Bilinear set-up of one sample for one lane: 4 corner byte offsets (kOOB when the corner is outside the map or
// the sample is not counted) and the 4 corner weights.  Pure VALU, recomputed at consume time instead of being
// kept live across the loads (registers are what limits loads in flight here).
struct Tap {
  unsigned o[4];
  float c[4];
};

__device__ __forceinline__ Tap make_tap(float x, float y, int H, int W, bool active, unsigned pix_bytes,
                                        unsigned lane_bytes) {
  Tap t;
  const float h_im = y * (float)H - 0.5f;
  const float w_im = x * (float)W - 0.5f;
  const bool ok = active && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h0 = (int)hf, w0 = (int)wf;
  const float lh = h_im - hf, lw = w_im - wf;
  const float hh = 1.f - lh, hw = 1.f - lw;
  const bool h0ok = ok && h0 >= 0, h1ok = ok && h0 + 1 <= H - 1;
  const bool w0ok = w0 >= 0, w1ok = w0 + 1 <= W - 1;
  const unsigned o00 = (unsigned)(h0 * W + w0) * pix_bytes + lane_bytes;
  t.o[0] = (h0ok && w0ok) ? o00 : kOOB;
  t.o[1] = (h0ok && w1ok) ? o00 + pix_bytes : kOOB;
  t.o[2] = (h1ok && w0ok) ? o00 + (unsigned)W * pix_bytes : kOOB;
  t.o[3] = (h1ok && w1ok) ? o00 + (unsigned)W * pix_bytes + pix_bytes : kOOB;
  t.c[0] = ok ? hh * hw : 0.f;
  t.c[1] = ok ? hh * lw : 0.f;
  t.c[2] = ok ? lh * hw : 0.f;
  t.c[3] = ok ? lh * lw : 0.f;
  return t;
}

// One sample's contribution to one channel, in ONE pinned operation order (every forward kernel uses it, so different
// schedules of the op give the same bits):  acc + ((((c1 v1) + c2 v2) + c3 v3) + c4 v4) * aw  with fused multiply-adds —
// the reference's `val = w1*v1 + w2*v2 + w3*v3 + w4*v4; col += val * weight` (ms_deform_im2col_cuda.cuh:82-88, :296)
// as a compiler contracts it.
__device__ __forceinline__ float accumulate_sample(float acc, float c1, float c2, float c3, float c4, float v1, float v2,
                                                   float v3, float v4, float aw) {
  float t = c1 * v1;
  t = __builtin_fmaf(c2, v2, t);
  t = __builtin_fmaf(c3, v3, t);
  t = __builtin_fmaf(c4, v4, t);
  return __builtin_fmaf(t, aw, acc);
}

}  // namespace dvis_msda
