// Shared device/host helpers for the Zero-TIG gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZT_OK 0
#define ZT_EINVAL 1001

// Every launcher returns 0 or a hipError_t / ZT_E* code; kernels never allocate or synchronise.
#define ZT_LAUNCH_CHECK()                  \
  do {                                     \
    hipError_t e__ = hipGetLastError();    \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)

#define ZT_REQUIRE(cond) \
  do {                   \
    if (!(cond)) return ZT_EINVAL; \
  } while (0)

static inline int zt_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long zt_cdivl(long long a, long long b) { return (a + b - 1) / b; }

__device__ __forceinline__ float zt_clampf(float v, float lo, float hi) {
  // torch.clamp semantics: min(max(v, lo), hi); NaN propagates through fmaxf/fminf differently but inputs are finite
  return fminf(fmaxf(v, lo), hi);
}

__device__ __forceinline__ float zt_wave_sum(float v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

// Block-wide sum of NV values per thread (blockDim.x*blockDim.y threads, multiple of 64, <= 1024).
// Result valid in thread 0.  `red` must hold NV*16 floats.
template <int NV>
__device__ __forceinline__ void zt_block_sum(float (&v)[NV], float* red) {
  int lin = threadIdx.y * blockDim.x + threadIdx.x;
  int lane = lin & 63, wave = lin >> 6;
  int nw = (blockDim.x * blockDim.y + 63) >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = zt_wave_sum(v[i]);
    if (lane == 0) red[i * 16 + wave] = s;
  }
  __syncthreads();
  if (lin == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += red[i * 16 + w];
      v[i] = s;
    }
  }
}

// reflect (no edge repeat) index for |overshoot| < n
__device__ __forceinline__ int zt_reflect(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}
