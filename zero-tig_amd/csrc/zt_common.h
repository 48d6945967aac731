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

// ---- bf16 storage helpers (activations / weights of the throughput mode are bf16 in HBM, math is fp32) -------------
typedef unsigned short zt_bf16;                                        // raw bits
typedef short zt_s16x4 __attribute__((ext_vector_type(4)));
typedef short zt_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 zt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float zt_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float zt_bf2f(zt_bf16 h) {
  unsigned u = ((unsigned)h) << 16;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
__device__ __forceinline__ float zt_u2f(unsigned u) {
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
__device__ __forceinline__ zt_bf16 zt_f2bf(float f) {                   // round to nearest even; hipcc emits v_cvt_pk_bf16_f32
  __bf16 h = (__bf16)f;
  zt_bf16 r;
  __builtin_memcpy(&r, &h, 2);
  return r;
}
// two floats -> packed bf16x2 in one dword (low half = a)
__device__ __forceinline__ unsigned zt_f2bf2(float a, float b) {
  typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  f2_t v = {a, b};
  bf2_t h = __builtin_convertvector(v, bf2_t);
  unsigned r;
  __builtin_memcpy(&r, &h, 4);
  return r;
}

// element access that is generic over the NHWC storage type
template <typename T> struct ZtIO;
template <> struct ZtIO<float> {
  static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ZtIO<zt_bf16> {
  static __device__ __forceinline__ float4 ld4(const zt_bf16* p) {
    uint2 r = *reinterpret_cast<const uint2*>(p);
    float4 v;
    unsigned a = r.x << 16, b = r.x & 0xFFFF0000u, c = r.y << 16, d = r.y & 0xFFFF0000u;
    __builtin_memcpy(&v.x, &a, 4); __builtin_memcpy(&v.y, &b, 4); __builtin_memcpy(&v.z, &c, 4); __builtin_memcpy(&v.w, &d, 4);
    return v;
  }
  static __device__ __forceinline__ void st4(zt_bf16* p, float4 v) {
    uint2 r;
    r.x = zt_f2bf2(v.x, v.y);
    r.y = zt_f2bf2(v.z, v.w);
    *reinterpret_cast<uint2*>(p) = r;
  }
  static __device__ __forceinline__ float ld(const zt_bf16* p) { return zt_bf2f(*p); }
  static __device__ __forceinline__ void st(zt_bf16* p, float v) { *p = zt_f2bf(v); }
};

// eight consecutive bf16 channels <-> eight floats (one 16-byte access per lane: the width the HBM path wants)
__device__ __forceinline__ void zt_ld8(const zt_bf16* p, float (&f)[8]) {
  const uint4 r = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = zt_u2f(w[j] << 16);
    f[2 * j + 1] = zt_u2f(w[j] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ void zt_st8(zt_bf16* p, const float (&f)[8]) {
  uint4 r;
  r.x = zt_f2bf2(f[0], f[1]);
  r.y = zt_f2bf2(f[2], f[3]);
  r.z = zt_f2bf2(f[4], f[5]);
  r.w = zt_f2bf2(f[6], f[7]);
  *reinterpret_cast<uint4*>(p) = r;
}

// D = A(16x32 bf16) * B(32x16 bf16) + C: lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], j = 0..7
__device__ __forceinline__ zt_f32x4 zt_mfma_bf16(zt_s16x8 a, zt_s16x8 b, zt_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(zt_bf16x8, a), __builtin_bit_cast(zt_bf16x8, b), c, 0, 0, 0);
}

// gfx950 transposing LDS read: per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4x16 block of
// 16-bit elements; lane i receives column i of the 4 rows (row q in element q).  EXEC must be all ones.
__device__ __forceinline__ zt_s16x4 zt_lds_read_tr16(const zt_bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) zt_s16x4*)p);
}

// gfx950 LDS-DMA: every lane copies 16 bytes from its own global address to lds_wave_base + 16 * lane (no VGPR destination; the
// destination is lane-linear, so a swizzled LDS image is produced by permuting the SOURCE addresses).  Completion is tracked
// by vmcnt; __syncthreads() drains it.
__device__ __forceinline__ void zt_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same DMA, issued from inline asm so that hipcc does not know about it.  Why: with the builtin form hipcc (ROCm 7.2) treats
// every pending LDS-DMA as a pending LDS write that any later LDS access it cannot disambiguate may alias -- and it cannot
// disambiguate the transposing read intrinsic (ds_read_b64_tr_b16): it puts `s_waitcnt vmcnt(0)` in front of the first such read
// after the DMA issue, i.e. the "prefetch" of the next tile is waited for before the current tile's MFMA loop starts (seen in the
// .s of the first build of wgrad64_dma_bf16_kernel).  Hidden from the compiler, the DMA is ours to order: ZT_WAIT_HIDDEN_DMA()
// (s_waitcnt vmcnt(0)) by every issuing wave, then a workgroup barrier, then the reads (cdna_hip_programming.md 5.7 item 1).
// M0 carries the wave-uniform LDS destination; it is compiler-reserved, so it is saved, set and restored inside ONE statement.
// The host-side test emulator pre-defines both macros (immediate copy / no-op).
#ifndef ZT_GLDS16_HIDDEN
__device__ __forceinline__ void zt_glds16_hidden_(const void* gsrc, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}
#define ZT_GLDS16_HIDDEN(gsrc, lds_wave_base) zt_glds16_hidden_((gsrc), (lds_wave_base))
#define ZT_WAIT_HIDDEN_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// s_waitcnt vmcnt(0) only (gfx9 encoding: vmcnt = simm16[15:14|3:0], expcnt [6:4] and lgkmcnt [11:8] left at their maxima)
__device__ __forceinline__ void zt_wait_vmcnt0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// s_waitcnt vmcnt(N), N < 16: all but the N most recently issued vector-memory operations have completed (in-order counter)
template <int N>
__device__ __forceinline__ void zt_wait_vmcnt() {
  static_assert(N >= 0 && N < 16, "vmcnt immediate");
  __builtin_amdgcn_s_waitcnt(0x0F70 | N);
}

// max(a, b) as ONE v_max_f32: fmaxf() under -fno-fast-math first canonicalises operands that may be signalling NaNs (an extra
// `v_max_f32 x, x, x` for every MFMA output).  Inputs here are finite.  The host-side test emulator pre-defines it as fmaxf.
#ifndef ZT_VMAX
__device__ __forceinline__ float zt_vmax_(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#define ZT_VMAX(a, b) zt_vmax_((a), (b))
#endif

// 16-byte global loads whose completion the KERNEL waits for, not the compiler: hipcc's wait insertion is conservative at control-
// flow joins (a load issued before a conditionally executed block of LDS-DMAs gets `vmcnt(0)` at its first use, i.e. it also waits
// for every DMA issued after it).  ZT_HIDDEN_LD16 issues the load from inline asm (the compiler believes dst is ready);
// ZT_HIDDEN_WAIT4<N> is `s_waitcnt vmcnt(N)` tied to the four destinations, so no use can be scheduled above it.  Between the
// two the destinations must not be read, copied or spilled -- check the .s after touching such code.  The host-side test
// emulator pre-defines both (plain load / no-op).
typedef unsigned zt_u32x4 __attribute__((ext_vector_type(4)));
#ifndef ZT_HIDDEN_LD16
#define ZT_HIDDEN_LD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
#define ZT_HIDDEN_WAIT4(N, a, b, c, d) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N))
#endif

// make a value opaque to the optimiser (keeps per-iteration address arithmetic from being hoisted out of a persistent loop and
// spilled).  The host-side test emulator pre-defines it as a no-op.
#ifndef ZT_OPAQUE
#define ZT_OPAQUE(x) asm volatile("" : "+v"(x))
#endif

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain vmcnt, so LDS-DMA / global loads
// issued before it stay in flight across it.  The host-side test emulator pre-defines it as __syncthreads().
#ifndef ZT_LDS_BARRIER
#define ZT_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// compile-time counted loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int I>
struct ZtIdx {
  static constexpr int value = I;
};
template <int B, int E, typename F>
__device__ __forceinline__ void zt_static_for(F&& f) {
  if constexpr (B < E) {
    f(ZtIdx<B>{});
    zt_static_for<B + 1, E>(f);
  }
}
