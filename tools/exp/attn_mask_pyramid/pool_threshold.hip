// Probe for DESIGN.md section 10 item 7 (NOT product code): the two small kernels a pyramid form of the decoder's attention
// masks needs around the existing contraction (dvis_mask_logits):
//   center_pool3   the four centre pixels of every s x s block of the stride-4 mask features, s = 2, 4, 8, averaged in the
//                  reference's order ((a + b) + (c + d)) * 0.25 — what F.interpolate(bilinear, align_corners=False) by an even
//                  integer factor samples — for all three decoder levels in ONE read of the map;
//   threshold_count  mask = logit < 0 (1 = blocked), allowed = number of pixels with logit >= 0 per (frame, query) row.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC pool_threshold.hip -o libpool_threshold.so   (run.py does it)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// one thread = one 8 x 8 block of one plane (lanes run along x: 32 contiguous bytes per lane and row)
__global__ __launch_bounds__(256) void center_pool3_kernel(const float *__restrict__ f, float *__restrict__ p2,
                                                           float *__restrict__ p4, float *__restrict__ p8, int H, int W,
                                                           long long blocks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= blocks) return;
  const int bw = W >> 3, bh = H >> 3;
  const int bx = (int)(i % bw);
  const long long r = i / bw;
  const int by = (int)(r % bh);
  const long long plane = r / bh;
  const float *src = f + (plane * H + by * 8) * (long long)W + bx * 8;
  float v[8][8];
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const f4 lo = *reinterpret_cast<const f4 *>(src + (long long)y * W);
    const f4 hi = *reinterpret_cast<const f4 *>(src + (long long)y * W + 4);
    v[y][0] = lo.x; v[y][1] = lo.y; v[y][2] = lo.z; v[y][3] = lo.w;
    v[y][4] = hi.x; v[y][5] = hi.y; v[y][6] = hi.z; v[y][7] = hi.w;
  }
  auto avg = [&](int y, int x) { return ((v[y][x] + v[y][x + 1]) + (v[y + 1][x] + v[y + 1][x + 1])) * 0.25f; };
  // s = 2: 4 x 4 outputs
  float *o2 = p2 + (plane * (H >> 1) + by * 4) * (long long)(W >> 1) + bx * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    *reinterpret_cast<f4 *>(o2 + (long long)k * (W >> 1)) = f4{avg(2 * k, 0), avg(2 * k, 2), avg(2 * k, 4), avg(2 * k, 6)};
  // s = 4: 2 x 2 outputs (rows 4i + 1, 4i + 2; columns 4j + 1, 4j + 2)
  float *o4 = p4 + (plane * (H >> 2) + by * 2) * (long long)(W >> 2) + bx * 2;
#pragma unroll
  for (int k = 0; k < 2; ++k) *reinterpret_cast<f2 *>(o4 + (long long)k * (W >> 2)) = f2{avg(4 * k + 1, 1), avg(4 * k + 1, 5)};
  // s = 8: rows 3, 4; columns 3, 4
  p8[(plane * bh + by) * (long long)bw + bx] = avg(3, 3);
}

// one workgroup per (frame, query) row of n logits (n % 4 == 0)
__global__ __launch_bounds__(256) void threshold_count_kernel(const float *__restrict__ logits, uint8_t *__restrict__ mask,
                                                              int *__restrict__ allowed, int n) {
  __shared__ int part[4];
  const long long row = blockIdx.x;
  const float *src = logits + row * n;
  uint8_t *dst = mask + row * n;
  int cnt = 0;
  for (int i = threadIdx.x * 4; i < n; i += 256 * 4) {
    const f4 x = *reinterpret_cast<const f4 *>(src + i);
    const unsigned b0 = x.x < 0.f, b1 = x.y < 0.f, b2 = x.z < 0.f, b3 = x.w < 0.f;
    *reinterpret_cast<unsigned *>(dst + i) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    cnt += 4 - (int)(b0 + b1 + b2 + b3);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) allowed[row] = part[0] + part[1] + part[2] + part[3];
}

extern "C" int probe_center_pool3(const float *f, float *p2, float *p4, float *p8, long long planes, int H, int W, void *stream) {
  if (H % 8 || W % 8) return 1;
  const long long blocks = planes * (H / 8) * (W / 8);
  hipLaunchKernelGGL(center_pool3_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f, p2, p4, p8, H,
                     W, blocks);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int probe_threshold_count(const float *logits, uint8_t *mask, int *allowed, long long rows, int n, void *stream) {
  if (n % 4) return 1;
  hipLaunchKernelGGL(threshold_count_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, mask, allowed, n);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
