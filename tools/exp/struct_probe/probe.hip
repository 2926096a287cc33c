// Does a structured buffer (stride = one image row) range-check every dword of a 16-byte load on gfx950?
//   hipcc --offload-arch=gfx950 -O2 probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
// clang has no builtin for the structured form; bind the LLVM intrinsic by name
__device__ f4 struct_load_f4(i4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");
__global__ void k(const float *x, int W, int rows, f4 *out) {
  // word3 0x00020000 = raw dword format as in dvis_make_rsrc; stride in bits 48..61 of the 128-bit descriptor
  const unsigned long long b = (unsigned long long)x;
  i4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) | ((unsigned)(W * 4) << 16)));
  r[2] = __builtin_amdgcn_readfirstlane(rows);
  r[3] = 0x00020000;
  const int t = threadIdx.x;
  // lane 0: row 1, column -1;  lane 1: row 1, column W-3;  lane 2: row = rows (past the end);  lane 3: row 1, column 2
  int row = t == 2 ? rows : 1;
  int col = t == 0 ? -1 : (t == 1 ? W - 3 : 2);
  out[t] = struct_load_f4(r, row, col * 4, 0, 0);
  // same with a scalar offset that selects another "plane" of `rows` rows (soffset is outside the range check?)
  out[4 + t] = struct_load_f4(r, row, col * 4, rows * W * 4, 0);
}
int main() {
  const int W = 8, rows = 4;
  float h[2 * W * rows];
  for (int i = 0; i < 2 * W * rows; ++i) h[i] = 100 + i;
  float *d; f4 *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8 * sizeof(f4));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, W, rows, o);
  float r[32];
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  const char *what[4] = {"row 1, col -1..2   (want 0 108 109 110)", "row 1, col W-3..W  (want 113 114 115 0)",
                         "row = rows         (want 0 0 0 0)", "row 1, col 2..5    (want 110 111 112 113)"};
  for (int t = 0; t < 4; ++t) printf("%s: %g %g %g %g   | +plane: %g %g %g %g\n", what[t], r[4 * t], r[4 * t + 1], r[4 * t + 2],
                                     r[4 * t + 3], r[16 + 4 * t], r[16 + 4 * t + 1], r[16 + 4 * t + 2], r[16 + 4 * t + 3]);
  return 0;
}
