// Sustained rate of v_mfma_f32_32x32x16_f16 with nothing else in the loop: what "the f16 matrix peak" is on this box under load.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float *out, int iters) {
  f16v acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  h8 a, b;
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int e = 0; e < 8; ++e) {
    st = st * 1664525u + 1013904223u;
    a[e] = iters < 0 ? (_Float16)(threadIdx.x * 0.001f + e) : (_Float16)(((int)(st >> 8) % 2000 - 1000) * 0.001f);
    st = st * 1664525u + 1013904223u;
    b[e] = iters < 0 ? (_Float16)(e * 0.5f) : (_Float16)(((int)(st >> 8) % 2000 - 1000) * 0.001f);
  }
  if (iters < 0) iters = -iters;
  h8 a2 = b, b2 = a;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (i & 1) ? __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b2, acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float *out;
  hipMalloc(&out, 4096 * 512 * 4);
  for (int blocks : {256}) for (int iters : {-20000, 20000, 100000, -100000}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 8 * (iters < 0 ? -iters : iters) * 8 * 32768.0;
    printf("blocks %d iters %d (negative: constant operands, else random): %.3f ms  %.1f TFLOP/s f16\n", blocks, iters, ms, flops / ms / 1e9);
  }
  return 0;
}
