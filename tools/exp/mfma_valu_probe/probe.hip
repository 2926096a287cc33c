// Does VALU work overlap with fp32 MFMA work on a gfx950 SIMD?  One workgroup of 8 waves per CU (2 waves per SIMD, like
// the attention / mask kernels); each wave loops over 16 independent v_mfma_f32_16x16x4_f32 (32 clk each) plus NV VALU
// operations (v_fma_f32) and NT transcendentals (v_exp_f32), clustered after the MFMAs or interleaved 1 MFMA : k VALU.
// Prints cycles per iteration per SIMD (2 waves), against 2 x 16 x 32 = 1024 MFMA cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV, int NT, bool INTERLEAVE>
__global__ __launch_bounds__(512) void probe(float *out, int iters, long long *cyc) {
  f4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (!INTERLEAVE) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NV; ++i) x[i & 7] = __builtin_fmaf(x[i & 7], b, a);
#pragma unroll
      for (int i = 0; i < NT; ++i) x[i & 7] = __builtin_amdgcn_exp2f(x[i & 7]);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < (NV + 15) / 16; ++k)
          if (i * ((NV + 15) / 16) + k < NV) x[(i + k) & 7] = __builtin_fmaf(x[(i + k) & 7], b, a);
#pragma unroll
        for (int k = 0; k < (NT + 15) / 16; ++k)
          if (i * ((NT + 15) / 16) + k < NT) x[(i + k + 3) & 7] = __builtin_amdgcn_exp2f(x[(i + k + 3) & 7]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, int NT, bool IL>
void run(const char *name, float *out, long long *cyc) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NV, NT, IL>), dim3(256), dim3(512), 0, 0, out, 10, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NV, NT, IL>), dim3(256), dim3(512), 0, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
  printf("%-34s %8.1f us   %7.0f ns/iter   counter %lld ticks/iter\n", name, ms * 1e3, ms * 1e6 / iters, c / iters);
}

int main() {
  float *out; long long *cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<0, 0, false>("16 mfma only", out, cyc);
  run<16, 0, false>("16 mfma + 16 valu clustered", out, cyc);
  run<32, 0, false>("16 mfma + 32 valu clustered", out, cyc);
  run<64, 0, false>("16 mfma + 64 valu clustered", out, cyc);
  run<16, 0, true>("16 mfma + 16 valu interleaved", out, cyc);
  run<32, 0, true>("16 mfma + 32 valu interleaved", out, cyc);
  run<64, 0, true>("16 mfma + 64 valu interleaved", out, cyc);
  run<0, 4, false>("16 mfma + 4 exp clustered", out, cyc);
  run<0, 16, false>("16 mfma + 16 exp clustered", out, cyc);
  run<0, 16, true>("16 mfma + 16 exp interleaved", out, cyc);
  run<24, 4, false>("16 mfma + 24 valu + 4 exp clustered", out, cyc);
  run<24, 4, true>("16 mfma + 24 valu + 4 exp interl.", out, cyc);
  return 0;
}
