// Probe: cost and correctness of a software grid barrier across all XCDs of an MI355X (dev experiment, not product).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ bool grid_barrier(unsigned *cnt, unsigned target, unsigned *fail) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1u << 22)) { ok = false; atomicAdd(fail, 1u); break; }   // bounded: never hang the GPU
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void probe(unsigned *cnt, unsigned *fail, unsigned *slots, unsigned *mism, int iters,
                                             int payload) {
  const unsigned G = gridDim.x, wg = blockIdx.x;
  unsigned bad = 0;
  for (int it = 1; it <= iters; ++it) {
    // every thread writes `payload` words, the neighbour (another XCD) reads them after the barrier
    for (int k = threadIdx.x; k < payload; k += 256) slots[(size_t)wg * payload + k] = (unsigned)it * 1000003u + k;
    grid_barrier(cnt, (unsigned)(2 * it - 1) * G, fail);
    const unsigned nb = (wg + 1) % G;
    for (int k = threadIdx.x; k < payload; k += 256)
      if (slots[(size_t)nb * payload + k] != (unsigned)it * 1000003u + k) ++bad;
    grid_barrier(cnt, (unsigned)(2 * it) * G, fail);   // nobody overwrites before everybody has read
  }
  if (bad) atomicAdd(mism, bad);
}

int main(int argc, char **argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, iters = 2000;
  for (int payload : {64, 4096}) {
    unsigned *cnt, *fail, *slots, *mism;
    hipMalloc(&cnt, 4); hipMalloc(&fail, 4); hipMalloc(&mism, 4); hipMalloc(&slots, (size_t)G * payload * 4);
    hipMemset(cnt, 0, 4); hipMemset(fail, 0, 4); hipMemset(mism, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, cnt, fail, slots, mism, iters, payload);
    hipEventRecord(e1);
    hipError_t rc = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned hf = 0, hm = 0; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost);
    printf("G=%d payload=%d words: rc=%d  %.3f us per barrier (2 per iter, incl. payload r/w)  timeouts=%u  stale reads=%u\n",
           G, payload, (int)rc, ms * 1e3 / (2.0 * iters), hf, hm);
  }
  return 0;
}
