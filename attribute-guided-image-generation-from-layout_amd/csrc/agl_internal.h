// Internal helpers shared by the HIP translation units of libagl.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define AGL_OK 0
#define AGL_ERR_ARG 1
#define AGL_ERR_LAUNCH 2
#define AGL_ERR_WORKSPACE 3

void agl_set_error(const char* fmt, ...);

#define AGL_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      agl_set_error(__VA_ARGS__);         \
      return AGL_ERR_ARG;                 \
    }                                     \
  } while (0)

#define AGL_CHECK_LAUNCH(name)                                             \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      agl_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return AGL_ERR_LAUNCH;                                               \
    }                                                                      \
  } while (0)

static inline int agl_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// out[i] (+)= sum over `splits` slabs of n floats (conv.hip); deterministic for a given slab count
int agl_launch_slab_reduce(const float* slabs, float* out, long n, int splits, int accumulate, hipStream_t st, const char* name,
                           const float* slabs_b = nullptr, float* out_b = nullptr, long nb = 0, int accumulate_b = 0);
// out[o] = epilogue(sum over slabs) with bias[(o / HW) % C], ReLU mask, accumulate, ReLU (conv.hip)
// (out_div: optional device scalar, the slab sum is divided by it first)
int agl_launch_splitk_epilogue(const float* slabs, float* out, long n, int splits, int HW, int C, const float* bias, const float* pos_mask,
                               int accumulate, int relu, hipStream_t st, const char* name, const float* out_div = nullptr);

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Block-wide sum for 256-thread blocks (4 waves); result valid in every thread.
template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T* scratch4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch4[w] = v;
  __syncthreads();
  return scratch4[0] + scratch4[1] + scratch4[2] + scratch4[3];
}
