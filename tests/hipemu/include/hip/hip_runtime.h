// TEST INFRASTRUCTURE ONLY (tests/hipemu): a minimal host emulation of the HIP device model so the product's
// single-source .hip kernels can be compiled with the host compiler and exercised on a machine without a GPU.
// It shadows <hip/hip_runtime.h> when tests/hipemu/include is first on the include path.  The product never
// builds or loads this; see tests/hipemu/README.md.
//
// Model: one workgroup at a time; every work-item is a fiber (hand-rolled x86-64 context switch); __syncthreads()
// and wave-level operations (shuffles, MFMA) are rendezvous points between fibers.  A wave is 64 consecutive
// linear thread ids.  MFMA lane layouts follow the gfx950 maps in /opt/skills/guides/cdna_hip_programming.md section 3
// and are additionally checked on the real GPU by tests/test_gpu_mfma_layout.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <algorithm>
using std::min;
using std::max;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __constant__ static const
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
template <class F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }

namespace emu {

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  uint3_ tid{};
  int lin = 0;
};

struct Wave {
  int alive = 0, arrived = 0, gen = 0;
  uint32_t buf[64][24];
};

struct Block {
  int nthreads = 0, alive = 0, arrived = 0, gen = 0;
  std::vector<Wave> waves;
};

extern "C" void zt_emu_switch(void** save_sp, void* new_sp);

struct State {
  Fiber* cur = nullptr;
  void* main_sp = nullptr;
  Block blk;
  uint3_ bid{}, bdim{}, gdim{};
  std::function<void()> body;
  std::vector<Fiber> fibers;
  std::vector<char*> stacks;
};
State& S();

static constexpr size_t kStack = 96 * 1024;

inline void yield() {
  State& s = S();
  zt_emu_switch(&s.cur->sp, s.main_sp);
}

inline void block_release_if_complete(Block& b) {
  if (b.alive > 0 && b.arrived == b.alive) { b.arrived = 0; b.gen++; }
}
inline void wave_release_if_complete(Wave& w) {
  if (w.alive > 0 && w.arrived == w.alive) { w.arrived = 0; w.gen++; }
}

inline void syncthreads() {
  State& s = S();
  Block& b = s.blk;
  int gen = b.gen;
  b.arrived++;
  block_release_if_complete(b);
  while (b.gen == gen) yield();
}

inline Wave& my_wave() { State& s = S(); return s.blk.waves[s.cur->lin >> 6]; }
inline int my_lane() { return S().cur->lin & 63; }

inline void wave_sync() {
  Wave& w = my_wave();
  int gen = w.gen;
  w.arrived++;
  wave_release_if_complete(w);
  while (w.gen == gen) yield();
}

void fiber_entry();
void launch(dim3 grid, dim3 block, std::function<void()> body);

template <class T> inline T shfl_from(T v, int src_lane) {
  static_assert(sizeof(T) == 4, "4-byte shuffles only");
  Wave& w = my_wave();
  int l = my_lane();
  memcpy(&w.buf[l][0], &v, 4);
  wave_sync();
  T r;
  memcpy(&r, &w.buf[src_lane & 63][0], 4);
  wave_sync();
  return r;
}

}  // namespace emu

#define threadIdx (emu::S().cur->tid)
#define blockIdx (emu::S().bid)
#define blockDim (emu::S().bdim)
#define gridDim (emu::S().gdim)
#define warpSize 64

static inline void __syncthreads() { emu::syncthreads(); }
static inline void __threadfence() {}                               // one workgroup runs at a time: stores are already visible
template <class T> static inline T __shfl_xor(T v, int m, int = 64) { return emu::shfl_from(v, emu::my_lane() ^ m); }
template <class T> static inline T __shfl_down(T v, int d, int = 64) { int l = emu::my_lane(); return emu::shfl_from(v, l + d < 64 ? l + d : l); }
template <class T> static inline T __shfl(T v, int src, int = 64) { return emu::shfl_from(v, src); }
template <class T> static inline T __builtin_amdgcn_readfirstlane(T v) { return v; }
static inline void __builtin_amdgcn_wave_barrier() { emu::wave_sync(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_sleep(int) {}

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { auto o = *p; *p = o + v; return o; }

static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
#define __expf(x) expf(x)
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }

// ---- MFMA (gfx950 lane maps) -------------------------------------------------------------------------------
typedef float zt_emu_f32x4 __attribute__((ext_vector_type(4)));
typedef float zt_emu_f32x16 __attribute__((ext_vector_type(16)));
typedef short zt_emu_s16x8 __attribute__((ext_vector_type(8)));

// v_mfma_f32_16x16x4_f32: A[row=l&15][k=l>>4], B[k=l>>4][col=l&15]; D: col=l&15,row=4*(l>>4)+reg; k-ordered fmaf chain
static inline zt_emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, zt_emu_f32x4 c, int, int, int) {
  emu::Wave& w = emu::my_wave();
  int l = emu::my_lane();
  memcpy(&w.buf[l][0], &a, 4);
  memcpy(&w.buf[l][1], &b, 4);
  emu::wave_sync();
  zt_emu_f32x4 d = c;
  int col = l & 15;
  for (int j = 0; j < 4; ++j) {
    int row = 4 * (l >> 4) + j;
    float acc = c[j];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, &w.buf[k * 16 + row][0], 4);
      memcpy(&bv, &w.buf[k * 16 + col][1], 4);
      acc = fmaf(av, bv, acc);
    }
    d[j] = acc;
  }
  emu::wave_sync();
  return d;
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31,row=(reg&3)+8*(reg>>2)+4*(l>>5)
static inline zt_emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, zt_emu_f32x16 c, int, int, int) {
  emu::Wave& w = emu::my_wave();
  int l = emu::my_lane();
  memcpy(&w.buf[l][0], &a, 4);
  memcpy(&w.buf[l][1], &b, 4);
  emu::wave_sync();
  zt_emu_f32x16 d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, &w.buf[k * 32 + row][0], 4);
      memcpy(&bv, &w.buf[k * 32 + col][1], 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  emu::wave_sync();
  return d;
}

static inline float zt_emu_bf16_to_f32(short s) { uint32_t u = ((uint32_t)(uint16_t)s) << 16; float f; memcpy(&f, &u, 4); return f; }

typedef __bf16 zt_emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef short zt_emu_s16x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x32_bf16: A[row=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][col=l&15]; D as 16x16x4
static inline zt_emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(zt_emu_bf16x8 a_, zt_emu_bf16x8 b_, zt_emu_f32x4 c, int, int, int) {
  zt_emu_s16x8 a = __builtin_bit_cast(zt_emu_s16x8, a_), b = __builtin_bit_cast(zt_emu_s16x8, b_);
  emu::Wave& w = emu::my_wave();
  int l = emu::my_lane();
  for (int j = 0; j < 8; ++j) {
    float fa = zt_emu_bf16_to_f32(a[j]), fb = zt_emu_bf16_to_f32(b[j]);
    memcpy(&w.buf[l][j], &fa, 4);
    memcpy(&w.buf[l][8 + j], &fb, 4);
  }
  emu::wave_sync();
  zt_emu_f32x4 d = c;
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    double acc = c[r];
    for (int k = 0; k < 32; ++k) {
      float av, bv;
      memcpy(&av, &w.buf[(k >> 3) * 16 + row][k & 7], 4);
      memcpy(&bv, &w.buf[(k >> 3) * 16 + col][8 + (k & 7)], 4);
      acc += (double)av * (double)bv;
    }
    d[r] = (float)acc;
  }
  emu::wave_sync();
  return d;
}

// v_mfma_f32_16x16x16_bf16: A[row=l&15][k=4*(l>>4)+j], B[k=4*(l>>4)+j][col=l&15], j = 0..3; D as 16x16x4
static inline zt_emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(zt_emu_s16x4 a, zt_emu_s16x4 b, zt_emu_f32x4 c, int, int, int) {
  emu::Wave& w = emu::my_wave();
  int l = emu::my_lane();
  for (int j = 0; j < 4; ++j) {
    float fa = zt_emu_bf16_to_f32(a[j]), fb = zt_emu_bf16_to_f32(b[j]);
    memcpy(&w.buf[l][j], &fa, 4);
    memcpy(&w.buf[l][8 + j], &fb, 4);
  }
  emu::wave_sync();
  zt_emu_f32x4 d = c;
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    double acc = c[r];
    for (int k = 0; k < 16; ++k) {
      float av, bv;
      memcpy(&av, &w.buf[(k >> 2) * 16 + row][k & 3], 4);
      memcpy(&bv, &w.buf[(k >> 2) * 16 + col][8 + (k & 3)], 4);
      acc += (double)av * (double)bv;
    }
    d[r] = (float)acc;
  }
  emu::wave_sync();
  return d;
}

// global_load_lds_dwordx4 & co: lane copies `size` bytes from its global address to (wave-uniform LDS base) + size * lane
static inline void zt_emu_glds(const void* g, void* lds_base, int size) {
  memcpy((char*)lds_base + (size_t)size * emu::my_lane(), g, (size_t)size);
}
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define ZT_OPAQUE(x) ((void)0)
#define ZT_LDS_BARRIER() __syncthreads()
#define ZT_GLDS16_HIDDEN(g, l) zt_emu_glds((const void*)(g), (void*)(l), 16)
#define ZT_WAIT_HIDDEN_DMA() ((void)0)
#define ZT_VMAX(a, b) fmaxf((a), (b))
#define ZT_HIDDEN_LD16(dst, ptr) ((dst) = *reinterpret_cast<const zt_u32x4*>(ptr))
#define ZT_HIDDEN_WAIT4(N, a, b, c, d) ((void)0)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) zt_emu_glds((const void*)(g), (void*)(l), (size))

// ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16-lane group, lane 4q+p supplies the address of row q, columns
// 4p..4p+3; lane i receives column i of the 4 rows (row q in element q).
static inline zt_emu_s16x4 zt_emu_ds_read_tr16(const void* p) {
  emu::Wave& w = emu::my_wave();
  int l = emu::my_lane();
  uint64_t addr = (uint64_t)(uintptr_t)p;
  memcpy(&w.buf[l][0], &addr, 8);
  emu::wave_sync();
  zt_emu_s16x4 r;
  int g = l & ~15, i = l & 15;
  for (int q = 0; q < 4; ++q) {
    uint64_t a;
    memcpy(&a, &w.buf[g + 4 * q + (i >> 2)][0], 8);
    r[q] = ((const short*)(uintptr_t)a)[i & 3];
  }
  emu::wave_sync();
  return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) zt_emu_ds_read_tr16((const void*)(p))

// v_mfma_f32_32x32x16_bf16: A[row=l&31][k=8*(l>>5)+j], B[k][col=l&31]; D as 32x32x2
static inline zt_emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(zt_emu_bf16x8 a_, zt_emu_bf16x8 b_, zt_emu_f32x16 c, int, int, int) {
  zt_emu_s16x8 a = __builtin_bit_cast(zt_emu_s16x8, a_), b = __builtin_bit_cast(zt_emu_s16x8, b_);
  emu::Wave& w = emu::my_wave();
  int l = emu::my_lane();
  for (int j = 0; j < 8; ++j) {
    float fa = zt_emu_bf16_to_f32(a[j]), fb = zt_emu_bf16_to_f32(b[j]);
    memcpy(&w.buf[l][j], &fa, 4);
    memcpy(&w.buf[l][8 + j], &fb, 4);
  }
  emu::wave_sync();
  zt_emu_f32x16 d = c;
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double acc = c[r];
    for (int k = 0; k < 16; ++k) {
      float av, bv;
      memcpy(&av, &w.buf[(k >> 3) * 32 + row][k & 7], 4);
      memcpy(&bv, &w.buf[(k >> 3) * 32 + col][8 + (k & 7)], 4);
      acc += (double)av * (double)bv;
    }
    d[r] = (float)acc;
  }
  emu::wave_sync();
  return d;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })
