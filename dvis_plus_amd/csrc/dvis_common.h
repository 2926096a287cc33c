// Shared helpers for the gfx950 kernels of libdvis_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/dvis_hip.h"

#define DVIS_EXPORT extern "C" __attribute__((visibility("default")))

void dvis_set_error(const char *fmt, ...);

// Check the launch that was just enqueued; report instead of printf-and-continue.
static inline int dvis_check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    dvis_set_error("%s: %s", what, hipGetErrorString(e));
    return DVIS_E_LAUNCH;
  }
  return DVIS_OK;
}

// > 64 KB of dynamic LDS needs hipFuncSetAttribute, and the attribute belongs to the CURRENT DEVICE's function object: a
// process that drives several GPUs must opt in on each.  `opted`: one static table per kernel instantiation (bytes already
// granted per device); racing threads at worst repeat the idempotent call.
struct DvisLdsOptIn {
  size_t bytes[64] = {};
};
static inline int dvis_lds_opt_in(const void *kernel, size_t bytes, DvisLdsOptIn *opted, const char *what) {
  if (bytes <= 64 * 1024) return DVIS_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  if (dev >= 0 && dev < 64 && __atomic_load_n(&opted->bytes[dev], __ATOMIC_RELAXED) >= bytes) return DVIS_OK;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    dvis_set_error("%s: hipFuncSetAttribute(max dynamic LDS = %zu): %s", what, bytes, hipGetErrorString(e));
    return DVIS_E_LAUNCH;
  }
  if (dev >= 0 && dev < 64) __atomic_store_n(&opted->bytes[dev], bytes, __ATOMIC_RELAXED);
  return DVIS_OK;
}

// Zero `nwords` 32-bit words as a KERNEL on the stream — not hipMemsetAsync.  A hipMemsetAsync captured into a hipGraph (ROCm 7.2,
// HIP 7.0.51831) replays correctly ONCE: from the second replay on the node fills the buffer with a garbage word instead of the
// value (minimal reproduction tools/exp/memset_graph_repro.py: 0 -> 0x0A40xxxx for 2000 B, 4 KB and 1 MB alike,
// profiles/r06_memset_graph_repro.txt).  The attention masks' `allowed_count` was zeroed that way: the first replay of a captured
// segmenter was right, every later one added its atomics onto ~4e8 — the round-5 "segmenter graph goes wrong after a tracker
// call" (tools/exp/seg_graph_bisect.py).  A kernel node replays like every other launch.
static __global__ void dvis_zero_words_kernel(uint32_t *__restrict__ p, size_t nwords) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nwords) p[i] = 0u;
}
static inline int dvis_zero_words(void *p, size_t nwords, hipStream_t st, const char *what) {
  if (nwords == 0) return DVIS_OK;
  hipLaunchKernelGGL(dvis_zero_words_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, (uint32_t *)p, nwords);
  return dvis_check_launch(what);
}

#define DVIS_REQUIRE(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      dvis_set_error(__VA_ARGS__);   \
      return DVIS_E_ARG;             \
    }                                \
  } while (0)

// fp32 <-> storage type conversions used by the generic (any-dtype) kernels.
template <typename T> struct dvis_acc { using type = float; };
template <> struct dvis_acc<double> { using type = double; };

template <typename A, typename T> __device__ __forceinline__ A dvis_load(const T *p) { return (A)(*p); }
template <> __device__ __forceinline__ float dvis_load<float, __half>(const __half *p) { return __half2float(*p); }
template <> __device__ __forceinline__ float dvis_load<float, __hip_bfloat16>(const __hip_bfloat16 *p) {
  return __bfloat162float(*p);
}
template <typename T, typename A> __device__ __forceinline__ void dvis_store(T *p, A v) { *p = (T)v; }
template <> __device__ __forceinline__ void dvis_store<__half, float>(__half *p, float v) { *p = __float2half(v); }
template <> __device__ __forceinline__ void dvis_store<__hip_bfloat16, float>(__hip_bfloat16 *p, float v) {
  *p = __float2bfloat16(v);
}

typedef unsigned dvis_v4u __attribute__((ext_vector_type(4)));
typedef float dvis_f4 __attribute__((ext_vector_type(4)));
typedef float dvis_f16v __attribute__((ext_vector_type(16)));

// Wave-uniform buffer descriptor over [base, base + bytes): loads past the end return 0 in hardware.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dvis_make_rsrc(const void *base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

// Same, with the (already uniform) inputs passed through readfirstlane so the descriptor is PROVABLY wave-uniform:
// otherwise hipcc wraps every buffer load in a waterfall loop (v_readfirstlane x4 + s_and_saveexec + branch).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dvis_make_rsrc_uniform(const void *base, unsigned bytes) {
  const uintptr_t b = (uintptr_t)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return dvis_make_rsrc((const void *)(((uintptr_t)hi << 32) | lo), __builtin_amdgcn_readfirstlane(bytes));
}
