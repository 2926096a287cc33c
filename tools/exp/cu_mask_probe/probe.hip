// Which compute units does bit i of a hipExtStreamCreateWithCUMask mask select on MI355X (8 XCDs x 32 CUs)?
//   hipcc --offload-arch=gfx950 -O2 probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
__global__ void where(unsigned *out) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // spin a little so that workgroups spread over every CU the stream may use
  for (volatile int i = 0; i < 2000; ++i) {}
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
}
int main() {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const int words = (ncu + 31) / 32, nb = 8192;
  unsigned *d;
  hipMalloc(&d, nb * 4);
  std::vector<unsigned> h(nb);
  auto run = [&](const char *name, std::vector<unsigned> mask) {
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, words, mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    for (unsigned v : h) cus.insert(v);
    for (unsigned v : cus) per_xcc[v >> 16]++;
    printf("%-34s distinct (xcc, se, sh, cu): %3zu   per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %2d", per_xcc[x]);
    printf("\n");
    hipStreamDestroy(st);
  };
  printf("%d CUs, %d mask words\n", ncu, words);
  run("all bits", std::vector<unsigned>(words, 0xffffffffu));
  { std::vector<unsigned> m(words, 0); m[0] = 0xffffffffu; run("bits 0..31", m); }
  { std::vector<unsigned> m(words, 0); m[0] = 0xff; run("bits 0..7", m); }
  { std::vector<unsigned> m(words, 0); m[0] = 0x1; run("bit 0", m); }
  { std::vector<unsigned> m(words, 0); m[0] = 0x2; run("bit 1", m); }
  { std::vector<unsigned> m(words, 0); m[0] = 0x100; run("bit 8", m); }
  { std::vector<unsigned> m(words, 0); for (int w = 0; w < words; ++w) m[w] = 0x01010101u; run("every 8th bit", m); }
  { std::vector<unsigned> m(words, 0); for (int w = 0; w < words; ++w) m[w] = 0x0000000fu; run("bits 32w..32w+3", m); }
  { std::vector<unsigned> m(words, 0xffffffffu); m[0] = 0; run("all but bits 0..31", m); }
  { std::vector<unsigned> m(words, 0xffffffffu); for (int w = 0; w < words; ++w) m[w] = 0xfffffff0u; run("all but bits 32w..32w+3", m); }
  return 0;
}
