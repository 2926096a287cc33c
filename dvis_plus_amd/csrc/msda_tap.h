// Shared device helpers of the MSDeformAttn forward kernels (msda_forward.hip, msda_forward_lds.hip).
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

}  // namespace dvis_msda
