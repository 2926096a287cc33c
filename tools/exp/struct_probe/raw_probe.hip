// Is a raw buffer 16-byte load range-checked per dword on gfx950?  (a) voffset = -4, (b) voffset = num_records - 12
//   hipcc --offload-arch=gfx950 -O2 raw_probe.hip -o raw_probe && ./raw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *x, int n, f4 *out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, n * 4, 0x00020000);
  const int t = threadIdx.x;
  const unsigned off = t == 0 ? (unsigned)-4 : (t == 1 ? (unsigned)(n * 4 - 12) : 8u);
  out[t] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  out[4 + t] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 64, 0));   // + soffset 64 bytes
}
int main() {
  const int n = 64;
  float h[2 * n];
  for (int i = 0; i < 2 * n; ++i) h[i] = 100 + i;
  float *d; f4 *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8 * sizeof(f4));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d + 16, n - 16, o);   // descriptor over [d + 16, d + 64): reads below / above are valid memory
  float r[32];
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  const char *what[3] = {"voffset -4      (per dword: 0 116 117 118)", "voffset end-12  (per dword: 161 162 163 0)", "voffset 8       (118 119 120 121)"};
  for (int t = 0; t < 3; ++t) printf("%s: %g %g %g %g   | soffset 64: %g %g %g %g\n", what[t], r[4 * t], r[4 * t + 1], r[4 * t + 2],
                                     r[4 * t + 3], r[16 + 4 * t], r[16 + 4 * t + 1], r[16 + 4 * t + 2], r[16 + 4 * t + 3]);
  return 0;
}
