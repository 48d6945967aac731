// Implicit-GEMM convolutions on the gfx950 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32), for the
// enhancement/denoising nets (reference model/model.py:15-81: conv2d of Denoise_1 / Denoise_2 / Enhancer), their
// data- and weight-gradients, and every RAFT convolution (model/RAFT/extractor.py, update.py) incl. the all-pairs
// correlation volume (corr.py:52-60, a 1x1 "convolution" whose weights are the second feature map).
//
// Layout: activations NHWC fp32 with explicit channel stride; weights [tap][Cin][ldw] (ldw = Cout rounded to 16).
// GEMM view: M = output pixels (16 consecutive pixels of one row per MFMA tile), N = Cout, K = taps x Cin.
// One workgroup = 4 waves = 4 output rows x 32 columns x (NT*16) output channels.  Per 16-channel input chunk the
// halo tile is staged once in LDS as [ci][row][col] planes (plane stride == 16 mod 32 banks => both MFMA operand
// reads are bank-conflict free), weights are staged per kernel row.
#include "zt_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct ConvArgs {
  const float* x;
  const float* x2;
  const float* w;
  const float* bias;
  const float* aux;
  float* y;
  int N, H, W, Cin, ldx, ldx2, csplit;
  int Ho, Wo, Cout, ldy, ldw, ldaux;
  int padH, padW;
  int act, epi, out_planar;
  float alpha;
  int tilesX, tilesY;
  float* y2;                   // epi 4: second destination (r * h), channels [esplit, Cout) go there
  int ldy2, esplit;
};

constexpr int TH = 4, TW = 32, CK = 16;

constexpr int plane_stride(int n) { return (n % 32 <= 16) ? n + (16 - n % 32) : n + (48 - n % 32); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : 0.2f * v;
    case 3: return 1.f / (1.f + expf(-v));
    case 4: return tanhf(v);
    case 5: return fminf(fmaxf(1.f / (1.f + expf(-v)), 0.0001f), 1.f);
    default: return v;
  }
}

template <int KH, int KW, int S, int NT>
__global__ void __launch_bounds__(256) conv_mfma_f32_kernel(ConvArgs a) {
  constexpr int IR = (TH - 1) * S + KH, IC = (TW - 1) * S + KW;
  constexpr int PLANE = plane_stride(IR * IC);
  constexpr int COP = plane_stride(NT * 16);
  __shared__ float xs[CK * PLANE];
  __shared__ float ws[KW * CK * COP];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t = blockIdx.x;
  const int tx = t % a.tilesX;
  t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int n = t / a.tilesY;
  const int co0 = blockIdx.y * (NT * 16);
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int gy0 = oy0 * S - a.padH, gx0 = ox0 * S - a.padW;

  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int l15 = lane & 15, l4 = lane >> 4;

  for (int c0 = 0; c0 < a.Cin; c0 += CK) {
    __syncthreads();
    // ---- stage the input halo tile for channels [c0, c0+16): 16 pixels x 4 channel-quads per 64 lanes
    {
      const bool second = a.x2 != nullptr && c0 >= a.csplit;
      const float* src = second ? a.x2 : a.x;
      const int ld = second ? a.ldx2 : a.ldx;
      const int cbase = second ? c0 - a.csplit : c0;
      const int climit = second ? a.Cin - a.csplit : (a.x2 ? a.csplit : a.Cin);
      constexpr int NGRP = (IR * IC + 15) / 16;
      for (int e = tid; e < NGRP * 64; e += 256) {
        int p = ((e >> 6) << 4) + (e & 15);
        int q = (e >> 4) & 3;
        if (p < IR * IC) {
          int iy = p / IC, ixx = p - iy * IC;
          int gy = gy0 + iy, gx = gx0 + ixx;
          int c = cbase + q * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < climit) {
            const float* g = src + ((size_t)(n * a.H + gy) * a.W + gx) * ld + c;
            if (c + 3 < climit) {
              v = *reinterpret_cast<const float4*>(g);
            } else {
              v.x = g[0];
              if (c + 1 < climit) v.y = g[1];
              if (c + 2 < climit) v.z = g[2];
            }
          }
          float* d = xs + (q * 4) * PLANE + p;
          d[0] = v.x;
          d[PLANE] = v.y;
          d[2 * PLANE] = v.z;
          d[3 * PLANE] = v.w;
        }
      }
    }
#pragma unroll 1
    for (int ky = 0; ky < KH; ++ky) {
      // ---- stage weights of kernel row ky for this channel chunk: ws[kx][ci][co]
      constexpr int NW4 = KW * CK * NT * 4;
      for (int e = tid; e < NW4; e += 256) {
        int co4 = e % (NT * 4);
        int r = e / (NT * 4);
        int ci = r % CK, kx = r / CK;
        int co = co0 + co4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + ci < a.Cin && co < a.ldw)
          v = *reinterpret_cast<const float4*>(a.w + ((size_t)(ky * KW + kx) * a.Cin + c0 + ci) * a.ldw + co);
        *reinterpret_cast<float4*>(ws + (kx * CK + ci) * COP + co4 * 4) = v;
      }
      __syncthreads();
      const float* xrow = xs + (wave * S + ky) * IC;
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const int ci = k4 * 4 + l4;
          float av[2], bv[NT];
#pragma unroll
          for (int m = 0; m < 2; ++m) av[m] = xrow[ci * PLANE + (m * 16 + l15) * S + kx];
#pragma unroll
          for (int q = 0; q < NT; ++q) bv[q] = ws[(kx * CK + ci) * COP + q * 16 + l15];
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[q], acc[m][q], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: D[row = 4*(lane>>4)+j][col = lane&15] -> pixel (oy, ox0 + m*16 + row), channel co0 + q*16 + col
  const int oy = oy0 + wave;
  if (oy >= a.Ho) return;
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int co = co0 + q * 16 + l15;
    if (co >= a.Cout) continue;
    const float b = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = ox0 + m * 16 + l4 * 4 + j;
        if (ox >= a.Wo) continue;
        float v = apply_act(a.alpha * (acc[m][q][j] + b), a.act);
        const size_t pix = (size_t)(n * a.Ho + oy) * a.Wo + ox;
        if (a.epi >= 4) {       // SepConvGRU fusions (update.py:42-58), nhwc only
          if (a.epi == 4) {     // [z | r] = sigmoid(conv): z is stored, r leaves as r * h
            if (co < a.esplit) a.y[pix * a.ldy + co] = v;
            else a.y2[pix * a.ldy2 + co - a.esplit] = v * a.aux[pix * a.ldaux + co - a.esplit];
          } else {              // q = tanh(conv): h = (1 - z) * h + z * q in place (aux = z)
            const float z = a.aux[pix * a.ldaux + co], hv = a.y[pix * a.ldy + co];
            a.y[pix * a.ldy + co] = (1.f - z) * hv + z * v;
          }
          continue;
        }
        if (a.epi) {
          float u = a.aux[pix * a.ldaux + co];
          if (a.epi == 1) v *= (u > 0.f ? 1.f : 0.2f);
          else if (a.epi == 2) v *= (u > 0.f ? 1.f : 0.f);
          else v += u;
        }
        if (a.out_planar) a.y[((size_t)n * a.Cout + co) * a.ldy + (size_t)oy * a.Wo + ox] = v;
        else a.y[pix * a.ldy + co] = v;
      }
    }
  }
}

// bf16 throughput mode: the activation result is rounded to bf16 (3 significant digits) right away, so the hardware
// exp / rcp (1 ulp-ish) replace the ~25-instruction libm expansions -- per element of the issue-bound small-map epilogues
__device__ __forceinline__ float apply_act_fast(float v, int act) {
  switch (act) {
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : 0.2f * v;
    case 3: return __builtin_amdgcn_rcpf(1.f + __expf(-v));
    case 4: return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * v) + 1.f);
    case 5: return fminf(fmaxf(__builtin_amdgcn_rcpf(1.f + __expf(-v)), 0.0001f), 1.f);
    default: return v;
  }
}

template <int KH, int KW, int S>
int launch_conv(const ConvArgs& a, int NT, dim3 grid_base, hipStream_t stream) {
  dim3 block(256);
  int c16 = (a.Cout + 15) / 16;
  dim3 grid(grid_base.x, (c16 + NT - 1) / NT);
  switch (NT) {
    case 1: hipLaunchKernelGGL((conv_mfma_f32_kernel<KH, KW, S, 1>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((conv_mfma_f32_kernel<KH, KW, S, 2>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((conv_mfma_f32_kernel<KH, KW, S, 3>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((conv_mfma_f32_kernel<KH, KW, S, 4>), grid, block, 0, stream, a); break;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[tap][ci][co] = sum_pixels x[p + tap][ci] * dz[p][co]      (stride 1, "same" padding)
// GEMM view: M = ci, N = co, K = pixels.  Each workgroup walks pixel tiles (4 rows x 16 cols) grid-stride and keeps
// its (tap, ci-tile) x co-tile accumulators in registers; partial slabs are summed by wgrad_reduce_kernel
// (deterministic; no float atomics).
struct WgradArgs {
  const float* x;
  const float* dz;
  float* slab;
  int H, W, Cin, ldx, Cout, lddz;
  int tilesX, ntiles;
};

constexpr int WTH = 4, WTW = 16;

template <int KH, int KW, int CT, int NT>
__global__ void __launch_bounds__(256) wgrad_mfma_f32_kernel(WgradArgs a) {
  constexpr int IR = WTH + KH - 1, IC = WTW + KW - 1;
  constexpr int CIP = plane_stride(CT * 16), COP = plane_stride(NT * 16);
  constexpr int NPAIR = KH * KW * CT;
  constexpr int PPW = (NPAIR + 3) / 4;
  __shared__ float xs[IR * IC * CIP];
  __shared__ float zs[WTH * WTW * COP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  constexpr int padH = (KH - 1) / 2, padW = (KW - 1) / 2;

  f32x4 acc[PPW][NT];
#pragma unroll
  for (int p = 0; p < PPW; ++p)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[p][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dz, folded in: thread (co = tid % NT16, part = tid / NT16) sums a strided share of each tile
  constexpr int NPART = 256 / (NT * 16);
  const int bco = tid % (NT * 16), bpart = tid / (NT * 16);
  float bsum = 0.f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int tx = tile % a.tilesX, ty = tile / a.tilesX;
    const int oy0 = ty * WTH, ox0 = tx * WTW;
    __syncthreads();
    // stage x halo tile [pix][ci] (zero beyond the image / beyond Cin)
    for (int e = tid; e < IR * IC * CT * 4; e += 256) {
      int c4 = e % (CT * 4), p = e / (CT * 4);
      int iy = p / IC, ixx = p - iy * IC;
      int gy = oy0 - padH + iy, gx = ox0 - padW + ixx;
      int c = c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.Cin) {
        const float* g = a.x + ((size_t)gy * a.W + gx) * a.ldx + c;
        if (c + 3 < a.Cin) v = *reinterpret_cast<const float4*>(g);
        else {
          v.x = g[0];
          if (c + 1 < a.Cin) v.y = g[1];
          if (c + 2 < a.Cin) v.z = g[2];
        }
      }
      *reinterpret_cast<float4*>(xs + p * CIP + c) = v;
    }
    for (int e = tid; e < WTH * WTW * NT * 4; e += 256) {
      int c4 = e % (NT * 4), p = e / (NT * 4);
      int iy = p / WTW, ixx = p - iy * WTW;
      int gy = oy0 + iy, gx = ox0 + ixx;
      int c = c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy < a.H && gx < a.W && c < a.Cout) {
        const float* g = a.dz + ((size_t)gy * a.W + gx) * a.lddz + c;
        if (c + 3 < a.Cout) v = *reinterpret_cast<const float4*>(g);
        else {
          v.x = g[0];
          if (c + 1 < a.Cout) v.y = g[1];
          if (c + 2 < a.Cout) v.z = g[2];
        }
      }
      *reinterpret_cast<float4*>(zs + p * COP + c) = v;
    }
    __syncthreads();
    if (bpart < NPART)
      for (int p = bpart; p < WTH * WTW; p += NPART) bsum += zs[p * COP + bco];
#pragma unroll 1
    for (int r = 0; r < WTH; ++r) {
#pragma unroll
      for (int k4 = 0; k4 < WTW / 4; ++k4) {
        const int col = k4 * 4 + l4;          // this lane's pixel (K index) within the row
        float bv[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) bv[q] = zs[(r * WTW + col) * COP + q * 16 + l15];
#pragma unroll
        for (int pi = 0; pi < PPW; ++pi) {
          const int pr = wave + 4 * pi;
          if (pr < NPAIR) {                    // wave-uniform
            const int tap = pr / CT, cit = pr - tap * CT;
            const int ky = tap / KW, kx = tap - ky * KW;
            float av = xs[((r + ky) * IC + col + kx) * CIP + cit * 16 + l15];
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[pi][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[q], acc[pi][q], 0, 0, 0);
          }
        }
      }
    }
  }
  // slab[block] = [tap][ci16][co16] weights partial, then [co16] bias partial
  float* out = a.slab + (size_t)blockIdx.x * (KH * KW * CT * 16 * NT * 16 + NT * 16);
#pragma unroll
  for (int pi = 0; pi < PPW; ++pi) {
    const int pr = wave + 4 * pi;
    if (pr < NPAIR) {
      const int tap = pr / CT, cit = pr - tap * CT;
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          out[((size_t)tap * CT * 16 + cit * 16 + l4 * 4 + j) * (NT * 16) + q * 16 + l15] = acc[pi][q][j];
    }
  }
  __syncthreads();
  if (bpart < NPART) zs[bpart * (NT * 16) + bco] = bsum;
  __syncthreads();
  if (tid < NT * 16) {
    float sum = 0.f;
    for (int k = 0; k < NPART; ++k) sum += zs[k * (NT * 16) + tid];
    out[KH * KW * CT * 16 * NT * 16 + tid] = sum;
  }
}

// grad_w[co][ci][ky][kx] (+)= sum_slabs slab[s][tap][ci][co]; grad_b[co] (+)= sum_slabs slab[s][bias tail].  32 slab
// elements (co fastest -> coalesced) x 8 slab groups per workgroup, eight loads in flight per thread; fixed summation order
// (deterministic).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, int ntap, int CT16,
                                                           int NT16, float* __restrict__ grad, int Cout, int Cin,
                                                           int accumulate, float* __restrict__ grad_b) {
  __shared__ float sh[256];
  const int ex = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + ex;
  const int nw = ntap * CT16 * NT16;
  const int total = nw + NT16;
  float s = 0.f;
  if (e < total) {
    const size_t stride = (size_t)total;
    for (int k = sg; k < nslab; k += 64) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (k + 8 * j < nslab) ? slab[(size_t)(k + 8 * j) * stride + e] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (sg == 0 && e < total) {
    s = (((sh[ex] + sh[32 + ex]) + (sh[64 + ex] + sh[96 + ex])) + ((sh[128 + ex] + sh[160 + ex]) + (sh[192 + ex] + sh[224 + ex])));
    if (e < nw) {
      int co = e % NT16;
      int ci = (e / NT16) % CT16;
      int tap = e / (NT16 * CT16);
      if (co < Cout && ci < Cin) {
        size_t o = ((size_t)co * Cin + ci) * ntap + tap;
        grad[o] = accumulate ? grad[o] + s : s;
      }
    } else if (grad_b) {
      int co = e - nw;
      if (co < Cout) grad_b[co] = accumulate ? grad_b[co] + s : s;
    }
  }
}

// One launch for ALL layers of a backward pass: segment s = blockIdx.y sums the slabs that every weight-gradient call of one layer
// appended to that layer's slab region (the three Denoise invocations, the three shared Enhancer blocks) and writes the layer's
// grad_w / grad_b.  Same per-element arithmetic as wgrad_reduce_kernel (fixed order: bit-reproducible); replaces 23 launches of
// ~9 us each per training step.
constexpr int ZT_MAXSEG = 16;
struct ReduceTable {
  const float* slab[ZT_MAXSEG];
  float* gw[ZT_MAXSEG];
  float* gb[ZT_MAXSEG];
  int nslab[ZT_MAXSEG], ntap[ZT_MAXSEG], CT16[ZT_MAXSEG], NT16[ZT_MAXSEG], Cout[ZT_MAXSEG], Cin[ZT_MAXSEG];
  int accumulate;
};

__global__ void __launch_bounds__(256) wgrad_reduce_multi_kernel(ReduceTable t) {
  __shared__ float sh[256];
  const int sgm = blockIdx.y;
  const float* __restrict__ slab = t.slab[sgm];
  const int nslab = t.nslab[sgm], ntap = t.ntap[sgm], CT16 = t.CT16[sgm], NT16 = t.NT16[sgm], Cout = t.Cout[sgm], Cin = t.Cin[sgm];
  const int ex = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + ex;
  const int nw = ntap * CT16 * NT16;
  const int total = nw + NT16;
  if (blockIdx.x * 32 >= total) return;                         // uniform: this segment is shorter than the longest one
  float s = 0.f;
  if (e < total) {
    const size_t stride = (size_t)total;
    for (int k = sg; k < nslab; k += 64) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (k + 8 * j < nslab) ? slab[(size_t)(k + 8 * j) * stride + e] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (sg == 0 && e < total) {
    s = (((sh[ex] + sh[32 + ex]) + (sh[64 + ex] + sh[96 + ex])) + ((sh[128 + ex] + sh[160 + ex]) + (sh[192 + ex] + sh[224 + ex])));
    if (e < nw) {
      const int co = e % NT16, ci = (e / NT16) % CT16, tap = e / (NT16 * CT16);
      if (co < Cout && ci < Cin) {
        const size_t o = ((size_t)co * Cin + ci) * ntap + tap;
        t.gw[sgm][o] = t.accumulate ? t.gw[sgm][o] + s : s;
      }
    } else if (t.gb[sgm]) {
      const int co = e - nw;
      if (co < Cout) t.gb[sgm][co] = t.accumulate ? t.gb[sgm][co] + s : s;
    }
  }
}

// all weight repacks of a step (forward and data-gradient operator of every layer) in one launch: entry = blockIdx.y
constexpr int ZT_MAXREP = 24;
struct RepackTable {
  const float* src[ZT_MAXREP];
  zt_bf16* dst[ZT_MAXREP];
  int Cout[ZT_MAXREP], Cin[ZT_MAXREP], K[ZT_MAXREP], CoutP[ZT_MAXREP], ldk[ZT_MAXREP], tflip[ZT_MAXREP];
};

__global__ void __launch_bounds__(256) repack_w_bf16_multi_kernel(RepackTable t) {
  const int en = blockIdx.y;
  const int Cout = t.Cout[en], Cin = t.Cin[en], KH = t.K[en], KW = t.K[en], CoutP = t.CoutP[en], ldk = t.ldk[en];
  const int total = Cout * Cin * KH * KW;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int kx = idx % KW, ky = (idx / KW) % KH, ci = (idx / (KW * KH)) % Cin, co = idx / (KW * KH * Cin);
    const zt_bf16 v = zt_f2bf(t.src[en][idx]);
    if (!t.tflip[en]) t.dst[en][((size_t)(ky * KW + kx) * CoutP + co) * ldk + ci] = v;
    else t.dst[en][((size_t)((KH - 1 - ky) * KW + (KW - 1 - kx)) * CoutP + ci) * ldk + co] = v;
  }
}

template <int KH, int KW>
int launch_wgrad(const WgradArgs& a, int CT, int NT, int nblk, hipStream_t stream) {
  dim3 grid(nblk), block(256);
#define ZT_WG(ct, nt) hipLaunchKernelGGL((wgrad_mfma_f32_kernel<KH, KW, ct, nt>), grid, block, 0, stream, a); return 0
  if (CT == 1 && NT == 3) { ZT_WG(1, 3); }
  if (CT == 1 && NT == 4) { ZT_WG(1, 4); }
  if (CT == 3 && NT == 3) { ZT_WG(3, 3); }
  if (CT == 3 && NT == 1) { ZT_WG(3, 1); }
  if (CT == 4 && NT == 4) { ZT_WG(4, 4); }
  if (CT == 4 && NT == 1) { ZT_WG(4, 1); }
#undef ZT_WG
  return ZT_EINVAL;
}

// =====================================================================================================================
// bf16 throughput mode: activations and weights are bf16 in HBM, accumulation fp32 (v_mfma_f32_16x16x32_bf16, 16x the
// fp32 matrix rate).  Same tiling as the fp32 kernels; the K step is 32 channels, both operands are read from LDS with
// one ds_read_b128 per fragment ([pixel][40] / [cout][40] bf16 rows: 80-byte pitch -> conflict free).
// Weights: [tap][CoutP16][ldk] with the input channel fastest (ldk = Cin rounded to 8, zero padded).
// =====================================================================================================================
struct ConvArgsH {
  const zt_bf16* x;
  const zt_bf16* x2;
  const zt_bf16* w;
  const float* bias;
  const zt_bf16* aux;
  void* y;
  int N, H, W, Cin, ldx, ldx2, csplit;
  int Ho, Wo, Cout, CoutP, ldk, ldy, ldaux;
  int padH, padW;
  int act, epi, out_mode;      // out_mode: 0 bf16 nhwc, 1 fp32 planar, 2 fp32 nhwc
  int dbg;                     // tuning ablations (tools/bench_conv.py): 1 skip MFMA loop, 2 skip epilogue, 4 skip prefetch
  float alpha;
  int tilesX, tilesY;
  zt_bf16* y2;                 // epi 4: second destination (r * h), channels [esplit, Cout) go there
  int ldy2, esplit;
  float* stats;                // conv_rs STATS: per-workgroup (sum, sum of squares) of the stored outputs, [grid][2][Cout]
  // conv_rs BSTATS (data gradient + residual of an Enhancer block): the BatchNorm backward sums of the PREVIOUS block, whose output
  // gradient this launch produces -- g = out * [bn_scale * zprev + bn_shift > 0]; stats[grid][2][Cout] = (sum g, sum g (zprev - bn_mean))
  const zt_bf16* zprev;
  int ldz;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
};

constexpr int HCK = 32;                 // channel granularity of a two-part (split) input

// ALL: every tap's weights of the current channel chunk fit in LDS next to the input tile -> 2 barriers per chunk;
// otherwise weights are staged per kernel row (7x7).  MT = 16-pixel MFMA tiles per wave along x (1 for small feature maps).
// CH2 = 32-channel MFMA K-steps per staged chunk: 2 (64 channels, 160-byte rows) halves the barrier / staging rounds of the
// latency-bound small-map layers whose Cin is a multiple of 64.
// PD = chunks of global loads in flight (register slots).  The small RAFT maps (45 x 80) are a serial chain of short kernels whose
// MFMA work per chunk (~0.2 us) cannot cover a global latency (~1-2 us): with PD = 3 nearly the whole K range is requested
// before the first MFMA instead of one latency being exposed per chunk.
template <int KH, int KW, int S, int NT, int MT, bool ALL, int CH2, int PD>
__device__ __forceinline__ void conv_mfma_bf16_body(const ConvArgsH& a, const int block_y) {
  static_assert(ALL || PD == 1, "per-row weight groups are staged inside the chunk");
  if (a.dbg & 8) return;                                        // tuning ablation (tools/bench_small.py): launch cost only
  constexpr int KCH = 32 * CH2, KCHP = CH2 == 2 ? 80 : 48, CPP = 4 * CH2;      // channels / LDS pitch / 16-byte chunks per pixel
  constexpr int TWm = 16 * MT;
  constexpr int IR = (TH - 1) * S + KH, IC = (TWm - 1) * S + KW;
  constexpr int TG = ALL ? KH * KW : KW;          // taps staged together
  constexpr int NG = ALL ? 1 : KH;
  constexpr int XS_ELEMS = IR * IC * KCHP, WS_ELEMS = TG * NT * 16 * KCHP;      // XS_ELEMS * 2 bytes is a multiple of 16 (KCHP is)
  __shared__ __attribute__((aligned(16))) zt_bf16 smem[XS_ELEMS + WS_ELEMS];    // pixel tile | weight tile; the fp32 epilogue re-uses both
  zt_bf16* const xs = smem;
  zt_bf16* const ws = smem + XS_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // grid = (tile columns, cout groups, tile rows x images): no integer divisions in the (issue-bound) prologue
  const int tx = blockIdx.x;
  int ty = blockIdx.z, n = 0;
  if (a.N > 1) {
    n = ty / a.tilesY;
    ty -= n * a.tilesY;
  }
  const int co0 = block_y * (NT * 16);
  const int oy0 = ty * TH, ox0 = tx * TWm;
  const int gy0 = oy0 * S - a.padH, gx0 = ox0 * S - a.padW;
  const int l15 = lane & 15, l4 = lane >> 4;

  zt_f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[m][q] = (zt_f32x4){0.f, 0.f, 0.f, 0.f};

  // global -> registers -> LDS, software-pipelined over the 32-channel chunks: all loads of a chunk are issued together
  // (clamped addresses, no branches; borders and ragged channel tails are masked when written) and the NEXT chunk's loads are
  // issued before this chunk's MFMAs, so one global latency is exposed per launch rather than several per chunk.
  constexpr int NWS = (TG * NT * 16 * CPP + 255) / 256;
  constexpr int NXS = (IR * IC * CPP + 255) / 256;
  static_assert(256 % CPP == 0, "a thread's channel octet is the same for all of its staging slots");
  uint4 wv[PD][NWS], xv[PD][NXS];
  // Chunk-invariant slot geometry, computed ONCE: the per-chunk staging code is then a handful of adds per 16-byte slot.  (With
  // the index arithmetic inside the chunk loop these kernels issued ~1200 scalar + vector ALU instructions per 20 MFMAs and
  // were issue-bound on it: every small-map RAFT layer took 11-16 us whatever its FLOP count.)
  const int q8 = (tid % CPP) * 8;                               // this thread's channel octet within a chunk (all slots)
  int x_src1[NXS], x_src2[NXS], x_lds[NXS];
  unsigned x_in[NXS];
#pragma unroll
  for (int i = 0; i < NXS; ++i) {
    const int e = tid + i * 256, p = e / CPP;
    const int gy = gy0 + p / IC, gx = gx0 + p % IC;
    const int gyc = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
    const int pix = (n * a.H + gyc) * a.W + gxc;
    x_src1[i] = pix * a.ldx + q8;
    x_src2[i] = pix * a.ldx2 + q8;
    x_lds[i] = e < IR * IC * CPP ? p * KCHP + q8 : -1;
    x_in[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? ~0u : 0u;
  }
  int w_src[NWS], w_lds[NWS];
  unsigned w_ok[NWS];
#pragma unroll
  for (int i = 0; i < NWS; ++i) {
    const int e = tid + i * 256, r = e / CPP;
    const int co = r % (NT * 16), tl = r / (NT * 16);
    const int tlc = tl < TG ? tl : TG - 1;                      // padding slots (never written) stay inside the weight array
    const int cor = co0 + co < a.CoutP ? co0 + co : a.CoutP - 1;
    w_src[i] = (tlc * a.CoutP + cor) * a.ldk + q8;
    w_lds[i] = e < TG * NT * 16 * CPP ? (tl * NT * 16 + co) * KCHP + q8 : -1;
    w_ok[i] = co0 + co < a.CoutP ? ~0u : 0u;
  }
  const int w_grp_stride = TG * a.CoutP * a.ldk;
  // uniform fast-path flags: a tile whose halo lies inside the image needs no zero fill, a workgroup whose couts all exist no
  // weight mask; full channel chunks need no tail masks.  Slots below the last one are in range for every thread (compile time).
  const bool x_interior = gy0 >= 0 && gy0 + IR <= a.H && gx0 >= 0 && gx0 + IC <= a.W;
  const bool w_all = co0 + NT * 16 <= a.CoutP;
  constexpr bool X_LAST_PARTIAL = (IR * IC * CPP) % 256 != 0, W_LAST_PARTIAL = (TG * NT * 16 * CPP) % 256 != 0;
  auto load_w = [&](auto sl, int c0, int grp) {
    constexpr int d = decltype(sl)::value;
    const int add = c0 + grp * w_grp_stride;
    const bool ragged = c0 + KCH > a.ldk;                       // uniform: only a ragged last chunk needs the channel clamp
#pragma unroll
    for (int i = 0; i < NWS; ++i) {
      int off = w_src[i] + add;
      if (ragged) off = c0 + q8 < a.ldk ? off : off - (c0 + q8);
      wv[d][i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.w) + 2u * (unsigned)off);
    }
  };
  auto write_w = [&](auto sl, int c0) {
    constexpr int d = decltype(sl)::value;
    const bool plain = w_all && c0 + KCH <= a.ldk && c0 < a.Cin;         // uniform
    const unsigned cok = (c0 + q8 < a.ldk && c0 < a.Cin) ? ~0u : 0u;   // beyond the weight row / a padding chunk: zeros
#pragma unroll
    for (int i = 0; i < NWS; ++i) {
      uint4 v = wv[d][i];
      if (!plain) {
        const unsigned m = w_ok[i] & cok;
        v.x &= m; v.y &= m; v.z &= m; v.w &= m;
      }
      if (!(W_LAST_PARTIAL && i == NWS - 1) || w_lds[i] >= 0) *reinterpret_cast<uint4*>(ws + w_lds[i]) = v;
    }
  };
  auto load_x = [&](auto sl, int c0) {
    constexpr int d = decltype(sl)::value;
    const bool second = a.x2 != nullptr && c0 >= a.csplit;
    const zt_bf16* src = second ? a.x2 : a.x;
    const int ld = second ? a.ldx2 : a.ldx;
    const int cbase = second ? c0 - a.csplit : c0;
    const bool ragged = cbase + KCH > ld;
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      int off = (second ? x_src2[i] : x_src1[i]) + cbase;
      if (ragged) off = cbase + q8 < ld ? off : off - (cbase + q8);
      xv[d][i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src) + 2u * (unsigned)off);
    }
  };
  auto write_x = [&](auto sl, int c0) {
    constexpr int d = decltype(sl)::value;
    const bool second = a.x2 != nullptr && c0 >= a.csplit;
    const int cbase = second ? c0 - a.csplit : c0;
    const int climit = second ? a.Cin - a.csplit : (a.x2 ? a.csplit : a.Cin);
    const bool full = cbase + KCH <= climit;                      // uniform: no channel tail in this chunk
    const int nv = climit - (cbase + q8);                         // valid channels of this thread's octet (ragged tail / beyond the input)
    const unsigned m0 = nv >= 2 ? ~0u : (nv == 1 ? 0xFFFFu : 0u), m1 = nv >= 4 ? ~0u : (nv == 3 ? 0xFFFFu : 0u);
    const unsigned m2 = nv >= 6 ? ~0u : (nv == 5 ? 0xFFFFu : 0u), m3 = nv >= 8 ? ~0u : (nv == 7 ? 0xFFFFu : 0u);
#pragma unroll
    for (int i = 0; i < NXS; ++i) {
      uint4 v = xv[d][i];
      if (!(full && x_interior)) {
        if (full) {
          v.x &= x_in[i]; v.y &= x_in[i]; v.z &= x_in[i]; v.w &= x_in[i];
        } else {
          v.x &= x_in[i] & m0;
          v.y &= x_in[i] & m1;
          v.z &= x_in[i] & m2;
          v.w &= x_in[i] & m3;
        }
      }
      if (!(X_LAST_PARTIAL && i == NXS - 1) || x_lds[i] >= 0) *reinterpret_cast<uint4*>(xs + x_lds[i]) = v;
    }
  };

  // Chunk loop.  Every load is issued UNCONDITIONALLY (chunk index clamped to the last one; the channel range is padded to a
  // multiple of PD chunks whose padding chunks are staged as zeros): with `if (more)` around the prefetch the compiler lost
  // count of the outstanding loads and put s_waitcnt vmcnt(0) in front of every LDS write, i.e. one full memory latency per
  // chunk however deep the prefetch (1.6-2.6 us per chunk on the 45 x 80 maps; measured with tools/bench_small.py).
  float bias_q[NT];                                             // requested now: the K loop hides the latency the epilogue used to expose
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int co = co0 + q * 16 + l15;
    bias_q[q] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
  }
  const int nch = (a.dbg & 4) ? 0 : (a.Cin + KCH - 1) / KCH;     // dbg 4: prologue + epilogue only
  const int last_c0 = (nch - 1) * KCH;
  if (nch > 0) {
    zt_static_for<0, PD>([&](auto sl) {
      constexpr int d = decltype(sl)::value;
      const int c0 = d * KCH <= last_c0 ? d * KCH : last_c0;
      load_x(sl, c0);
      load_w(sl, c0, 0);
    });
  }
  // fragment reads run LA (tap, channel-half) steps ahead of the MFMAs that consume them (LDS latency ~100+ clocks against
  // MT*NT*16 clocks of MFMA per step; the compiler's own schedule waited for each step's reads right before its MFMAs)
  constexpr int NSTEP = TG * CH2;
  constexpr int LA = NSTEP > 2 ? 2 : 1;
  const zt_bf16* xfrag = xs + ((wave * S) * IC + l15 * S) * KCHP + 8 * l4;
  const zt_bf16* wfrag = ws + l15 * KCHP + 8 * l4;
  const int ngroups = (nch + PD - 1) / PD;
  const int nrep = (a.dbg & 16) ? 4 : 1;                          // ablation: walk the K range four times (cold-start vs steady-state cost)
#pragma unroll 1
  for (int rep = 0; rep < nrep; ++rep)
#pragma unroll 1
  for (int g = 0; g < ngroups; ++g) {
    zt_static_for<0, PD>([&](auto sl) {
      constexpr int d = decltype(sl)::value;
      const int c0 = (g * PD + d) * KCH;                          // >= Cin: a padding chunk (staged as zeros)
      __syncthreads();
      write_x(sl, c0);
      write_w(sl, c0);
      __syncthreads();
      const int cn = c0 + PD * KCH <= last_c0 ? c0 + PD * KCH : last_c0;
      load_x(sl, cn);
      if (ALL) load_w(sl, cn, 0);                                 // single tap group: its weights are prefetched as well
#pragma unroll 1
      for (int grp = 0; grp < NG; ++grp) {
        if (grp > 0) {                                            // per-kernel-row weight groups (7x7): staged inside the chunk
          __syncthreads();
          load_w(sl, c0, grp);
          write_w(sl, c0);
          __syncthreads();
        }
        zt_s16x8 fa[LA + 1][MT], fb[LA + 1][NT];
        auto loadf = [&](auto bc, auto sc) {
          constexpr int bi = decltype(bc)::value, step = decltype(sc)::value;
          constexpr int tl = step / CH2, kc = step % CH2;
          const int ky = ALL ? tl / KW : grp, kx = ALL ? tl % KW : tl;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            fa[bi][m] = *reinterpret_cast<const zt_s16x8*>(xfrag + (ky * IC + m * 16 * S + kx) * KCHP + kc * 32);
#pragma unroll
          for (int q = 0; q < NT; ++q)
            fb[bi][q] = *reinterpret_cast<const zt_s16x8*>(wfrag + (tl * NT * 16 + q * 16) * KCHP + kc * 32);
        };
        zt_static_for<0, LA>([&](auto sc) { loadf(ZtIdx<decltype(sc)::value % (LA + 1)>{}, sc); });
        zt_static_for<0, NSTEP>([&](auto sc) {
          constexpr int step = decltype(sc)::value;
          constexpr int cur = step % (LA + 1);
          if constexpr (step + LA < NSTEP) loadf(ZtIdx<(step + LA) % (LA + 1)>{}, ZtIdx<step + LA>{});
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[m][q] = zt_mfma_bf16(fa[cur][m], fb[cur][q], acc[m][q]);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      if (!ALL) load_w(sl, cn, 0);
    });
  }

  const int oy = oy0 + wave;
  // fp32 nhwc output without a fused operand (the all-pairs correlation volume, corr.py:52-60: 52 MB at 1080p): the accumulator
  // layout gives every lane 4-byte stores 64 bytes apart; transposed through LDS each lane writes 16 contiguous bytes of a
  // pixel's cout run instead.  Wave-private slice of the (now idle) pixel / weight staging buffers.
  constexpr int SP = NT * 16 + 4;                               // staging row pitch in floats
  constexpr bool CAN_STAGE = 4 * 16 * MT * SP * 4 <= (XS_ELEMS + WS_ELEMS) * 2;
  if constexpr (CAN_STAGE) {
    if (a.out_mode == 2 && a.epi == 0 && a.ldy % 4 == 0 && !(a.dbg & 34)) {      // uniform
      __syncthreads();                                          // every wave is done with the operand tiles
      float* stg = reinterpret_cast<float*>(smem) + wave * (16 * MT * SP);
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            stg[(m * 16 + l4 * 4 + j) * SP + q * 16 + l15] = apply_act_fast(a.alpha * (acc[m][q][j] + bias_q[q]), a.act);
      __builtin_amdgcn_wave_barrier();                          // same wave writes and reads (LDS ops of a wave complete in order)
      if (oy < a.Ho) {
        constexpr int C4 = NT * 4;                              // 16-byte chunks per pixel
        for (int e = lane; e < 16 * MT * C4; e += 64) {
          const int p = e / C4, c4 = e - p * C4;
          const int ox = ox0 + p, co = co0 + c4 * 4;
          if (ox < a.Wo && co < a.Cout) {
            const float4 v = *reinterpret_cast<const float4*>(stg + p * SP + c4 * 4);
            float* dst = (float*)a.y + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.ldy + co;
            if (co + 4 <= a.Cout) *reinterpret_cast<float4*>(dst) = v;
            else {
              const float t[4] = {v.x, v.y, v.z, v.w};
              for (int k = 0; k < 4 && co + k < a.Cout; ++k) dst[k] = t[k];
            }
          }
        }
      }
      return;
    }
  }
  // bf16 nhwc output without a fused operand (most RAFT layers): same transposition, bf16 -- one 16-byte store per lane instead of
  // eight 2-byte stores that each touch four 32-byte pieces of different lines (the epilogue was ~3 us of every small-map launch:
  // tools/bench_small.py "full" vs "no-epilogue")
  constexpr int SPH = NT * 16 + 8;                              // staging row pitch in bf16 elements (16-byte multiple)
  if constexpr (4 * 16 * MT * SPH * 2 <= (XS_ELEMS + WS_ELEMS) * 2) {
    if (a.out_mode == 0 && a.epi == 0 && a.ldy % 8 == 0 && (((uintptr_t)a.y) & 15) == 0 && !(a.dbg & 34)) {      // uniform
      __syncthreads();
      zt_bf16* stg = smem + wave * (16 * MT * SPH);
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            stg[(m * 16 + l4 * 4 + j) * SPH + q * 16 + l15] = zt_f2bf(apply_act_fast(a.alpha * (acc[m][q][j] + bias_q[q]), a.act));
      __builtin_amdgcn_wave_barrier();
      if (oy < a.Ho) {
        constexpr int C8 = NT * 2;                              // 16-byte chunks per pixel
        for (int e = lane; e < 16 * MT * C8; e += 64) {
          const int p = e / C8, c8 = e - p * C8;
          const int ox = ox0 + p, co = co0 + c8 * 8;
          if (ox < a.Wo && co < a.Cout) {
            const uint4 v = *reinterpret_cast<const uint4*>(stg + p * SPH + c8 * 8);
            zt_bf16* dst = (zt_bf16*)a.y + ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.ldy + co;
            if (co + 8 <= a.Cout) *reinterpret_cast<uint4*>(dst) = v;
            else {
              zt_bf16 t[8];
              __builtin_memcpy(t, &v, 16);
              for (int k = 0; k < 8 && co + k < a.Cout; ++k) dst[k] = t[k];
            }
          }
        }
      }
      return;
    }
  }
  // bf16 nhwc output with a fused operand (residual add, ReLU masks, the two GRU fusions): activated values staged in fp32 so the
  // arithmetic is the scalar path's; operands and results move as 16-byte chunks of 8 channels
  if constexpr (CAN_STAGE) {
    const bool al = a.ldy % 8 == 0 && (((uintptr_t)a.y) & 15) == 0 && a.ldaux % 8 == 0 && (((uintptr_t)a.aux) & 15) == 0 &&
                    (a.epi != 4 || (a.ldy2 % 8 == 0 && a.esplit % 8 == 0 && (((uintptr_t)a.y2) & 15) == 0));
    if (a.out_mode == 0 && a.epi != 0 && al && !(a.dbg & 34)) {  // uniform
      __syncthreads();
      float* stg = reinterpret_cast<float*>(smem) + wave * (16 * MT * SP);
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            stg[(m * 16 + l4 * 4 + j) * SP + q * 16 + l15] = apply_act_fast(a.alpha * (acc[m][q][j] + bias_q[q]), a.act);
      __builtin_amdgcn_wave_barrier();
      if (oy < a.Ho) {
        constexpr int C8 = NT * 2;
        for (int e = lane; e < 16 * MT * C8; e += 64) {
          const int p = e / C8, c8 = e - p * C8;
          const int ox = ox0 + p, co = co0 + c8 * 8;
          if (ox >= a.Wo || co >= a.Cout) continue;
          const size_t pix = (size_t)(n * a.Ho + oy) * a.Wo + ox;
          const float4 va = *reinterpret_cast<const float4*>(stg + p * SP + c8 * 8);
          const float4 vb = *reinterpret_cast<const float4*>(stg + p * SP + c8 * 8 + 4);
          const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
          const bool second = a.epi == 4 && co >= a.esplit;     // the r half of [z | r]
          const bool whole = co + 8 <= a.Cout;
          const zt_bf16* up = a.aux + pix * a.ldaux + (second ? co - a.esplit : co);
          zt_bf16* dst = second ? a.y2 + pix * a.ldy2 + (co - a.esplit) : (zt_bf16*)a.y + pix * a.ldy + co;
          const bool need_u = a.epi != 4 || second;
          auto combine = [&](float r, float uf, float hf) {
            if (a.epi == 4) return second ? r * uf : r;
            if (a.epi == 5) return (1.f - uf) * hf + uf * r;
            if (a.epi == 1) return r * (uf > 0.f ? 1.f : 0.2f);
            if (a.epi == 2) return r * (uf > 0.f ? 1.f : 0.f);
            if (a.epi == 6) return fmaxf(r + uf, 0.f);          // ResidualBlock: relu(x + y) (extractor.py:56)
            return r + uf;
          };
          if (whole) {                                          // registers only: no indexed local arrays (they would go to scratch)
            uint4 uq = make_uint4(0, 0, 0, 0), hq = make_uint4(0, 0, 0, 0), oq;
            if (need_u) uq = *reinterpret_cast<const uint4*>(up);
            if (a.epi == 5) hq = *reinterpret_cast<const uint4*>(dst);
            const unsigned uw[4] = {uq.x, uq.y, uq.z, uq.w}, hw[4] = {hq.x, hq.y, hq.z, hq.w};
            unsigned ow[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float lo = combine(v[2 * k], zt_u2f(uw[k] << 16), zt_u2f(hw[k] << 16));
              const float hi = combine(v[2 * k + 1], zt_u2f(uw[k] & 0xffff0000u), zt_u2f(hw[k] & 0xffff0000u));
              ow[k] = zt_f2bf2(lo, hi);
            }
            oq = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            *reinterpret_cast<uint4*>(dst) = oq;
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
              if (co + k < a.Cout)
                dst[k] = zt_f2bf(combine(v[k], need_u ? zt_bf2f(up[k]) : 0.f, a.epi == 5 ? zt_bf2f(dst[k]) : 0.f));
          }
        }
      }
      return;
    }
  }
  if (oy >= a.Ho) return;
  if (a.dbg & 2) {                                              // ablation: no epilogue (accumulators kept live)
    if (acc[0][0][0] == 12345.678f) ((float*)a.y)[0] = acc[MT - 1][NT - 1][3];
    return;
  }
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int co = co0 + q * 16 + l15;
    if (co >= a.Cout) continue;
    const float b = bias_q[q];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = ox0 + m * 16 + l4 * 4 + j;
        if (ox >= a.Wo) continue;
        float v = apply_act_fast(a.alpha * (acc[m][q][j] + b), a.act);
        const size_t pix = (size_t)(n * a.Ho + oy) * a.Wo + ox;
        if (a.epi == 4 || a.epi == 5) {       // SepConvGRU fusions (update.py:42-58), bf16 nhwc only
          zt_bf16* yb = (zt_bf16*)a.y;
          if (a.epi == 4) {     // [z | r] = sigmoid(conv): z is stored, r leaves as r * h
            if (co < a.esplit) yb[pix * a.ldy + co] = zt_f2bf(v);
            else a.y2[pix * a.ldy2 + co - a.esplit] = zt_f2bf(v * zt_bf2f(a.aux[pix * a.ldaux + co - a.esplit]));
          } else {              // q = tanh(conv): h = (1 - z) * h + z * q in place (aux = z)
            const float z = zt_bf2f(a.aux[pix * a.ldaux + co]), hv = zt_bf2f(yb[pix * a.ldy + co]);
            yb[pix * a.ldy + co] = zt_f2bf((1.f - z) * hv + z * v);
          }
          continue;
        }
        if (a.epi) {
          float u = zt_bf2f(a.aux[pix * a.ldaux + co]);
          if (a.epi == 1) v *= (u > 0.f ? 1.f : 0.2f);
          else if (a.epi == 2) v *= (u > 0.f ? 1.f : 0.f);
          else v += u;
          if (a.epi == 6) v = fmaxf(v, 0.f);
        }
        if (a.out_mode == 1) ((float*)a.y)[((size_t)n * a.Cout + co) * a.ldy + (size_t)oy * a.Wo + ox] = v;
        else if (a.out_mode == 2) ((float*)a.y)[pix * a.ldy + co] = v;
        else ((zt_bf16*)a.y)[pix * a.ldy + co] = zt_f2bf(v);
      }
    }
  }
}

template <int KH, int KW, int S, int NT, int MT, bool ALL, int CH2, int PD>
__global__ void __launch_bounds__(256) conv_mfma_bf16_kernel(ConvArgsH a) {
  conv_mfma_bf16_body<KH, KW, S, NT, MT, ALL, CH2, PD>(a, (int)blockIdx.y);
}

// TWO independent convolutions of the same kernel instantiation and the same map in ONE launch: cout groups [0, ysplit) of the
// grid's y axis run problem a0, the rest a1 (RAFT's motion encoder: convc2 || convf2, update.py:91-94 -- the small-map layers are
// bound by their fixed launch + prologue + epilogue cost, and neither of the two fills the chip on its own)
template <int KH, int KW, int S, int NT, int MT, bool ALL, int CH2, int PD>
__global__ void __launch_bounds__(256) conv_mfma_bf16_pair_kernel(ConvArgsH a0, ConvArgsH a1, int ysplit) {
  const bool first = (int)blockIdx.y < ysplit;                    // uniform
  conv_mfma_bf16_body<KH, KW, S, NT, MT, ALL, CH2, PD>(first ? a0 : a1, first ? (int)blockIdx.y : (int)blockIdx.y - ysplit);
}

// ... and of two DIFFERENT instantiations (same map, 16-pixel tiles): convc1 (1x1, 324 -> 256) || convf1 (7x7, 2 -> 128), the two
// heads of the motion encoder (update.py:89, 91).  Each body has its own static LDS tile; a workgroup uses one of them.
template <int KH1, int KW1, int NT1, bool ALL1, int CH21, int PD1, int KH2, int KW2, int NT2, bool ALL2, int CH22, int PD2>
__global__ void __launch_bounds__(256) conv_mfma_bf16_pair2_kernel(ConvArgsH a0, ConvArgsH a1, int ysplit) {
  if ((int)blockIdx.y < ysplit) conv_mfma_bf16_body<KH1, KW1, 1, NT1, 1, ALL1, CH21, PD1>(a0, (int)blockIdx.y);
  else conv_mfma_bf16_body<KH2, KW2, 1, NT2, 1, ALL2, CH22, PD2>(a1, (int)blockIdx.y - ysplit);
}

template <int KH, int KW, int S, int MT>
int launch_conv_h(const ConvArgsH& a, int NT, unsigned gx, hipStream_t stream) {
  dim3 block(256);
  int c16 = (a.Cout + 15) / 16;
  (void)gx;
  if ((long long)a.tilesY * a.N > 65535) return ZT_EINVAL;
  dim3 grid(a.tilesX, (c16 + NT - 1) / NT, a.tilesY * a.N);
  constexpr int IRc = (TH - 1) * S + KH, ICc = (16 * MT - 1) * S + KW;
  // 64-channel chunks where every chunk is full: Cin (and the split point of a two-part input) multiples of 64
  const bool wide = a.Cin % 64 == 0 && (!a.x2 || a.csplit % 64 == 0);
  // latency-bound launches (about two workgroups per CU or fewer, several channel chunks): two chunks of loads in flight
  const bool deep = MT == 1 && (long long)grid.x * grid.y * grid.z <= 1024 && a.Cin > 64;
#define ZT_CH(nt)                                                                                             \
  {                                                                                                           \
    constexpr bool all1 = (KH * KW * nt * 16 + IRc * ICc) * 48 * 2 <= 72 * 1024;                              \
    constexpr bool all2 = (KH * KW * nt * 16 + IRc * ICc) * 80 * 2 <= 64 * 1024;                              \
    constexpr bool fits2 = all2;              /* only while >= 2 workgroups still fit a CU: larger tiles lose more than they gain */ \
    if constexpr (fits2) {                                                                                    \
      if (wide) {                                                                                             \
        if constexpr (MT == 1) {                                                                              \
          if (deep) {                                                                                         \
            hipLaunchKernelGGL((conv_mfma_bf16_kernel<KH, KW, S, nt, MT, all2, 2, 2>), grid, block, 0, stream, a); \
            break;                                                                                            \
          }                                                                                                   \
        }                                                                                                     \
        hipLaunchKernelGGL((conv_mfma_bf16_kernel<KH, KW, S, nt, MT, all2, 2, 1>), grid, block, 0, stream, a); \
        break;                                                                                                \
      }                                                                                                       \
    }                                                                                                         \
    if constexpr (MT == 1 && all1) {                                                                          \
      if (deep) {                                                                                             \
        hipLaunchKernelGGL((conv_mfma_bf16_kernel<KH, KW, S, nt, MT, all1, 1, 2>), grid, block, 0, stream, a); \
        break;                                                                                                \
      }                                                                                                       \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_mfma_bf16_kernel<KH, KW, S, nt, MT, all1, 1, 1>), grid, block, 0, stream, a);    \
  }
  switch (NT) {
    case 1: ZT_CH(1) break;
    case 2: ZT_CH(2) break;
    case 3: ZT_CH(3) break;
    default: ZT_CH(4) break;
  }
#undef ZT_CH
  return 0;
}

// ---- persistent, weight-stationary variant for the full-resolution enhancement / denoising layers (stride 1, K in {1,3},
// Cin <= 64): each workgroup (8 waves = 8 output rows x 32 columns) loads ALL its weights into LDS once and then walks
// pixel tiles grid-stride; the next tile's halo is prefetched into registers while the MFMAs of the current one run
// (two barriers per tile).  LDS rows are [pixel | cout][CCH*32 + 8] bf16 (144 B or 80 B pitch: conflict-free b128 reads).
template <int K, int NT, int CCH, int PTH>
__global__ void __launch_bounds__(64 * PTH) conv_ws_bf16_kernel(ConvArgsH a, int ntiles) {
  constexpr int NTHR = 64 * PTH;
  constexpr int CP = CCH == 2 ? 80 : 48;       // 160 B / 96 B row pitch: conflict-free ds_read_b128 (brute-forced over lane groups)
  constexpr int IR = PTH + K - 1, IC = TW + K - 1;
  constexpr int NPF = (IR * IC * CCH * 4 + NTHR - 1) / NTHR;     // 16-byte prefetch registers per thread
  constexpr int XS_HALO = IR * IC * CP, XS_STAGE = PTH * TW * (NT * 16 + 8);      // halo tile / output staging share xs
  __shared__ __attribute__((aligned(16))) zt_bf16 ws[K * K * NT * 16 * CP];
  __shared__ __attribute__((aligned(16))) zt_bf16 xs[XS_HALO > XS_STAGE ? XS_HALO : XS_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int co0 = blockIdx.y * (NT * 16);
  constexpr int pad = (K - 1) / 2;

  for (int e = tid; e < K * K * NT * 16 * CCH * 4; e += NTHR) {
    int q = e % (CCH * 4);
    int r = e / (CCH * 4);
    int co = r % (NT * 16), tap = r / (NT * 16);
    int c = q * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (c < a.ldk && co0 + co < a.CoutP) v = *reinterpret_cast<const uint4*>(a.w + ((size_t)tap * a.CoutP + co0 + co) * a.ldk + c);
    *reinterpret_cast<uint4*>(ws + (tap * NT * 16 + co) * CP + c) = v;
  }

  // the halo element a thread fetches is the same for every tile: precompute its (row, col, channel) once
  uint4 pf[NPF];
  int pf_iy[NPF], pf_ix[NPF], pf_c[NPF];
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    int e = tid + i * NTHR;
    int q = e % (CCH * 4), p = e / (CCH * 4);
    pf_iy[i] = e < IR * IC * CCH * 4 ? p / IC : -100000;       // out-of-range slots never pass the bounds test
    pf_ix[i] = p % IC;
    pf_c[i] = q * 8;
  }
  auto prefetch = [&](int tile) {
    const int tx = tile % a.tilesX, ty = tile / a.tilesX;
    const int gy0 = ty * PTH - pad, gx0 = tx * TW - pad;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      const int gy = gy0 + pf_iy[i], gx = gx0 + pf_ix[i], c = pf_c[i];
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.Cin) {
        v = *reinterpret_cast<const uint4*>(a.x + ((size_t)gy * a.W + gx) * a.ldx + c);
        if (c + 8 > a.Cin) {
          zt_bf16 tmp[8];
          __builtin_memcpy(tmp, &v, 16);
          for (int j = 0; j < 8; ++j)
            if (c + j >= a.Cin) tmp[j] = 0;
          __builtin_memcpy(&v, tmp, 16);
        }
      }
      pf[i] = v;
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  // de-phase neighbouring workgroups by ~half a tile so that HBM reads, MFMA work and HBM writes of different CUs interleave
  // instead of the whole chip moving through the same phase in lock-step (speed only; no correctness dependence)
  if ((a.dbg & 8) == 0 && (blockIdx.x & 1)) {
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
  }
  for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      int e = tid + i * NTHR;
      if (e < IR * IC * CCH * 4) *reinterpret_cast<uint4*>(xs + (e / (CCH * 4)) * CP + (e % (CCH * 4)) * 8) = pf[i];
    }
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < ntiles && !(a.dbg & 4)) prefetch(next);

    zt_f32x4 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[m][q] = (zt_f32x4){0.f, 0.f, 0.f, 0.f};
    if (!(a.dbg & 1)) {
      // software-pipelined over the K*K*CCH (tap, channel-half) steps: the fragments of step i+1 are in flight while the
      // MFMAs of step i issue.  Weights are the A operand, pixels the B operand: D[row = cout 4*(lane>>4)+j][col = pixel
      // lane&15], i.e. every lane ends up with 4 CONSECUTIVE output channels of one pixel (8-byte staging writes below).
      constexpr int NSTEP = K * K * CCH;
      zt_s16x8 av[2][2], bv[2][NT];
      const zt_bf16* xb = xs + (wave * IC + l15) * CP + 8 * l4;
      const zt_bf16* wb = ws + l15 * CP + 8 * l4;
#define ZT_LOADF(buf, step)                                                                                         \
  {                                                                                                                 \
    constexpr int tap_ = (step) / CCH, kc_ = (step) % CCH, ky_ = tap_ / K, kx_ = tap_ % K;                          \
    _Pragma("unroll") for (int m = 0; m < 2; ++m) av[buf][m] =                                                      \
        *reinterpret_cast<const zt_s16x8*>(xb + (ky_ * IC + m * 16 + kx_) * CP + kc_ * 32);                         \
    _Pragma("unroll") for (int q = 0; q < NT; ++q) bv[buf][q] =                                                     \
        *reinterpret_cast<const zt_s16x8*>(wb + (tap_ * NT * 16 + q * 16) * CP + kc_ * 32);                         \
  }
      ZT_LOADF(0, 0)
      zt_static_for<0, NSTEP>([&](auto step_c) {
        constexpr int step = decltype(step_c)::value;
        constexpr int cur = step & 1;
        if constexpr (step + 1 < NSTEP) ZT_LOADF(cur ^ 1, step + 1)
        __builtin_amdgcn_sched_barrier(0);      // keep the next step's LDS reads ahead of this step's MFMAs (hipcc re-serialises them otherwise)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q = 0; q < NT; ++q) acc[m][q] = zt_mfma_bf16(bv[cur][q], av[cur][m], acc[m][q]);
        __builtin_amdgcn_sched_barrier(0);
      });
#undef ZT_LOADF
    }

    const int tx = tile % a.tilesX, ty = tile / a.tilesX;
    const int oy = ty * PTH + wave, ox0 = tx * TW;
    if (a.dbg & 2) {
      if (acc[0][0][0] == 12345.678f) ((float*)a.y)[0] = acc[1][NT - 1][3];      // keep the accumulators live
    } else if (a.out_mode == 0) {
      // bf16 nhwc output: transpose the accumulators through LDS (wave-private slice of the halo buffer) so that global
      // traffic is 16 bytes per lane (2-byte stores are store-issue bound: ~15x slower on this layer)
      constexpr int OP = NT * 16 + 8;                      // staging row pitch (elements)
      __syncthreads();                                      // every wave is done reading xs
      zt_bf16* st = xs + wave * (TW * OP);
      // none / ReLU / LeakyReLU(0.2) are max(v, slope*v) with slope 1 / 0 / 0.2: branch-free on the hot path (a runtime switch
      // expanded over the 32 accumulators blew up the code size and the instruction cache); other activations go the slow way.
      const bool simple_act = a.act <= 2;
      const float slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.2f);
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const int cb = co0 + q * 16 + l4 * 4;
        float bj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bj[j] = (a.bias && cb + j < a.Cout) ? a.bias[cb + j] : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = a.alpha * (acc[m][q][j] + bj[j]);
            v[j] = fmaxf(v[j], slope * v[j]);
          }
          if (!simple_act) {
#pragma unroll 1
            for (int j = 0; j < 4; ++j) v[j] = apply_act(a.alpha * (acc[m][q][j] + bj[j]), a.act);
          }
          uint2 pk;
          pk.x = zt_f2bf2(v[0], v[1]);
          pk.y = zt_f2bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(st + (m * 16 + l15) * OP + q * 16 + l4 * 4) = pk;
        }
      }
      // same wave wrote and reads: LDS ops of one wave complete in order, so no workgroup barrier is needed; the wave barrier
      // only pins the compiler's (and the test emulator's) ordering of the two phases
      __builtin_amdgcn_wave_barrier();
      if (oy < a.Ho) {
        for (int e = lane; e < TW * NT * 2; e += 64) {
          const int p = e / (NT * 2), c8 = (e % (NT * 2)) * 8;
          const int ox = ox0 + p, co = co0 + c8;
          if (ox < a.Wo && co < a.Cout) {
            uint4 v = *reinterpret_cast<const uint4*>(st + p * OP + c8);
            const size_t pix = (size_t)oy * a.Wo + ox;
            if (a.epi) {
              uint4 u = *reinterpret_cast<const uint4*>(a.aux + pix * a.ldaux + co);
              zt_bf16 tv[8], tu[8];
              __builtin_memcpy(tv, &v, 16);
              __builtin_memcpy(tu, &u, 16);
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                float fv = zt_bf2f(tv[k]), fu = zt_bf2f(tu[k]);
                if (a.epi == 1) fv *= (fu > 0.f ? 1.f : 0.2f);
                else if (a.epi == 2) fv *= (fu > 0.f ? 1.f : 0.f);
                else fv += fu;
                tv[k] = zt_f2bf(fv);
              }
              __builtin_memcpy(&v, tv, 16);
            }
            zt_bf16* dst = (zt_bf16*)a.y + pix * a.ldy + co;
            if (co + 8 <= a.Cout) *reinterpret_cast<uint4*>(dst) = v;
            else {
              zt_bf16 tv[8];
              __builtin_memcpy(tv, &v, 16);
              for (int k = 0; k < 8 && co + k < a.Cout; ++k) dst[k] = tv[k];
            }
          }
        }
      }
    } else if (oy < a.Ho) {
#pragma unroll
      for (int q = 0; q < NT; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co0 + q * 16 + l4 * 4 + j;
          if (co < a.Cout) {
            const float b = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              const int ox = ox0 + m * 16 + l15;
              if (ox < a.Wo) {
                float v = apply_act(a.alpha * (acc[m][q][j] + b), a.act);
                const size_t pix = (size_t)oy * a.Wo + ox;
                if (a.epi) {
                  float u = zt_bf2f(a.aux[pix * a.ldaux + co]);
                  if (a.epi == 1) v *= (u > 0.f ? 1.f : 0.2f);
                  else if (a.epi == 2) v *= (u > 0.f ? 1.f : 0.f);
                  else v += u;
                }
                if (a.out_mode == 1) ((float*)a.y)[(size_t)co * a.ldy + pix] = v;
                else ((float*)a.y)[pix * a.ldy + co] = v;
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int K>
int launch_conv_ws(ConvArgsH& a, int NT, int CCH, int pth, hipStream_t stream) {
  int c16 = (a.Cout + 15) / 16;
  a.tilesY = zt_cdiv(a.Ho, pth);
  int ntiles = a.tilesX * a.tilesY;
  // LDS per workgroup decides how many are co-resident per CU (phases of different workgroups overlap HBM reads, MFMA and stores)
  int cp = CCH == 2 ? 80 : 48;
  int lds = 2 * (K * K * NT * 16 * cp + (pth + K - 1) * (TW + K - 1) * cp);
  int per_cu = 160 * 1024 / (lds + 1024);
  per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
  int gx = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
  dim3 grid(gx, (c16 + NT - 1) / NT), block(64 * pth);
#define ZT_WS(nt, cch)                                                                               \
  hipLaunchKernelGGL((conv_ws_bf16_kernel<K, nt, cch, 8>), grid, block, 0, stream, a, ntiles);      \
  return 0
  if (CCH == 1) {
    if (NT == 1) { ZT_WS(1, 1); }
    if (NT == 2) { ZT_WS(2, 1); }
    if (NT == 3) { ZT_WS(3, 1); }
    ZT_WS(4, 1);
  }
  if (NT == 1) { ZT_WS(1, 2); }
  if (NT == 2) { ZT_WS(2, 2); }
  if (NT == 3) { ZT_WS(3, 2); }
  ZT_WS(4, 2);
#undef ZT_WS
}

// ---- 1x1 convolution with a thin input (Cin <= 8: the data gradient of Denoise_1/2's 48 -> 3 / 48 -> 6 output layers).
// 2 * Cin FLOP per output element: a pure streaming kernel, no MFMA.  Thread = one cout octet x 4 pixels (weights for its 8
// couts live in registers); load j of a wave covers 64 / (Cout/8) consecutive pixels; 16-byte loads and stores throughout.
// PLAIN: the path's only use (data gradient of Denoise_1/2's 1x1 output layer: no bias, no activation, alpha 1, LeakyReLU-mask
// epilogue) with 48 couts -- compile-time octet count (the 64-bit i % Q8, i / Q8 and the per-element runtime epilogue selection
// made the generic form issue-bound: ~1600 instructions per thread for 32 outputs, 3.1 TB/s)
template <bool PLAIN>
__global__ void __launch_bounds__(256) conv1x1_thin_bf16_kernel(ConvArgsH a, int npg) {
  const int Q8 = PLAIN ? 6 : (a.Cout >> 3);
  int o, pg;
  if constexpr (PLAIN) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    o = (int)(i % 6u);
    pg = (int)(i / 6u);
  } else {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    o = (int)(i % Q8);
    pg = (int)(i / Q8);
  }
  if (pg >= npg) return;
  const int HW = a.Ho * a.Wo;
  float w[8][8], b[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    zt_ld8(a.w + (size_t)(o * 8 + c) * a.ldk, w[c]);            // [CoutP][ldk = 8], zero beyond Cin
    b[c] = (!PLAIN && a.bias) ? a.bias[o * 8 + c] : 0.f;
  }
  const float slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.2f);
  const float neg = a.epi == 1 ? 0.2f : 0.f;
  float x[4][8], u[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = min(pg + j * npg, HW - 1);
    zt_ld8(a.x + (size_t)p * a.ldx, x[j]);
    if (PLAIN || a.epi) zt_ld8(a.aux + (size_t)p * a.ldaux + o * 8, u[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)                                   // the buffer's padding lanes are not trusted (NaN * 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) x[j][k] = k < a.Cin ? x[j][k] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = pg + j * npg;
    float r[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s = fmaf(w[c][k], x[j][k], s);
      if constexpr (PLAIN) {
        s *= (u[j][c] > 0.f ? 1.f : 0.2f);
      } else {
        s = a.alpha * (s + b[c]);
        s = fmaxf(s, slope * s);
        if (a.epi == 3) s += u[j][c];
        else if (a.epi) s *= (u[j][c] > 0.f ? 1.f : neg);
      }
      r[c] = s;
    }
    if (p < HW) zt_st8((zt_bf16*)a.y + (size_t)p * a.ldy + o * 8, r);
  }
}

// ---- Backward of Denoise_1/2's 1x1 output layer (model.py:27, 43: conv3, 48 -> 3 / 6) in ONE pass over its two operands.
// The data gradient dz2 = (W3^T dr) * LeakyReLU'(a2) and the weight / bias gradients dW3[co][ci] = sum_p dr[p][co] a2[p][ci],
// db3[co] = sum_p dr[p][co] read the same a2 (48 ch) and dr (8 ch) pixels: two launches (conv1x1_thin 43 us + wgrad<1,1,3,1,4> 34 us
// per call, six calls per step) each streamed a2 once; here it is streamed once for both.  Same thread layout as conv1x1_thin
// (thread = 4 pixels x one 8-channel octet of a2 / dz2, 16-byte accesses), as a grid-stride loop so that the 6 x 8 products per
// pixel accumulate in registers; a workgroup = 42 pixel groups x 6 octets (252 of 256 threads); at the end the 42 partial sets of
// an octet are summed in a FIXED order through LDS (bit-reproducible: no atomics) into one slab of the layout the batched slab
// reduction expects ([ci 48][co 16] + [co 16]).  dz2 is bit-identical to conv1x1_thin_bf16_kernel<true>'s.
struct ThinBwdArgs {
  const zt_bf16* dr;       // [HW][8]   gradient of the 1x1 output (Cdr = 3 or 6 valid channels)
  const zt_bf16* wT;       // [48][8]   data-gradient operator: row = a2 channel, column = output channel (zero beyond Cdr)
  const zt_bf16* a2;       // [HW][lda] the layer's input activation (LeakyReLU output)
  zt_bf16* dz;             // [HW][lddz] data gradient w.r.t. the pre-activation of a2
  float* slab;             // [grid][48 * 16 + 16]
  int HW, npg, lda, lddz, Cdr;
};

__global__ void __launch_bounds__(256) thin1x1_bwd_bf16_kernel(ThinBwdArgs a) {
  __shared__ float red[252 * 49];                                // 48 products + pad: per-thread partial sets, then the bias sets
  const int tid = threadIdx.x;
  const int o = tid % 6, gl = tid / 6;                           // octet of a2 / dz2, pixel group inside the workgroup (0..41; 42: idle)
  const bool active = tid < 252;
  float w[8][8];
#pragma unroll
  for (int c = 0; c < 8; ++c) zt_ld8(a.wT + (size_t)(o * 8 + c) * 8, w[c]);
  float acc[6][8], bsum[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    bsum[k] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[k][c] = 0.f;
  }
  for (int pg = blockIdx.x * 42 + gl; active && pg < a.npg; pg += gridDim.x * 42) {
    float x[4][8], u[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = min(pg + j * a.npg, a.HW - 1);
      zt_ld8(a.dr + (size_t)p * 8, x[j]);
      zt_ld8(a.a2 + (size_t)p * a.lda + o * 8, u[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)                                 // the buffer's padding lanes are not trusted (NaN * 0)
#pragma unroll
      for (int k = 0; k < 8; ++k) x[j][k] = k < a.Cdr ? x[j][k] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = pg + j * a.npg;
      float r[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum = fmaf(w[c][k], x[j][k], sum);
        r[c] = sum * (u[j][c] > 0.f ? 1.f : 0.2f);
      }
      if (p < a.HW) {
        zt_st8(a.dz + (size_t)p * a.lddz + o * 8, r);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          bsum[k] += x[j][k];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[k][c] = fmaf(x[j][k], u[j][c], acc[k][c]);
        }
      }
    }
  }
  // deterministic workgroup reduction: every thread publishes its 48 partial products, then (ci, co) is summed over the 42 pixel
  // groups of its octet in index order; the bias sums (identical in the 6 octet threads of a pixel group) go through the same buffer
  float* out = a.slab + (size_t)blockIdx.x * (48 * 16 + 16);
  if (active) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int c = 0; c < 8; ++c) red[tid * 49 + k * 8 + c] = acc[k][c];
  }
  __syncthreads();
  for (int e = tid; e < 48 * 16; e += 256) {
    const int ci = e >> 4, co = e & 15;
    float sum = 0.f;
    if (co < 6) {
      const int oo = ci >> 3, c = ci & 7;
      for (int g = 0; g < 42; ++g) sum += red[(g * 6 + oo) * 49 + co * 8 + c];
    }
    out[e] = sum;
  }
  __syncthreads();
  if (active && o == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[gl * 8 + k] = bsum[k];
  }
  __syncthreads();
  if (tid < 16) {
    float sum = 0.f;
    if (tid < 6)
      for (int g = 0; g < 42; ++g) sum += red[g * 8 + tid];
    out[48 * 16 + tid] = sum;
  }
}

// ---- 1x1 convolution with a thin fp32 planar output (Cout <= 8: the 48 -> 3 / 48 -> 6 output layers of Denoise_1/2).  Streaming:
// thread = one pixel, reads its Cin/8 16-byte chunks (a wave reads one contiguous span), weights are wave-uniform (scalar
// loads), each output plane is written coalesced.
template <int CO>
__global__ void __launch_bounds__(256) conv1x1_thinout_bf16_kernel(ConvArgsH a) {
  const int HW = a.Ho * a.Wo;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  uint4 xv[8];                                                  // Cin <= 64: all of the pixel's loads in flight together
#pragma unroll
  for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const uint4*>(a.x + (size_t)p * a.ldx + (i * 8 < a.Cin ? i * 8 : 0));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i * 8 < a.Cin) {                                        // uniform
      const unsigned xw[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
      float x[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[2 * j] = zt_u2f(xw[j] << 16);
        x[2 * j + 1] = zt_u2f(xw[j] & 0xFFFF0000u);
      }
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        float w[8];
        zt_ld8(a.w + (size_t)c * a.ldk + i * 8, w);             // uniform address: scalar loads
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[c] = fmaf(w[k], x[k], acc[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CO; ++c)
    if (c < a.Cout) ((float*)a.y)[(size_t)c * a.ldy + p] = apply_act(a.alpha * (acc[c] + (a.bias ? a.bias[c] : 0.f)), a.act);
}

// ---- register-stationary persistent kernel for the full-resolution 3x3 layers (stride 1, bf16 nhwc output, 48 or 64 couts).
// The LDS-fed kernels above are LDS-bandwidth bound (0.75 fragment reads per MFMA against the 0.5 that 128 B/clk sustains), so
// here the WEIGHTS LIVE IN REGISTERS for the whole launch (one persistent workgroup per CU, <= 162 VGPRs of A fragments per
// wave) and LDS only carries pixels: a wave owns two adjacent output rows, so every pixel fragment it reads from the 4 halo
// rows feeds both rows (ky and ky-1) -- 0.17-0.33 reads per MFMA.  What LDS capacity that frees goes to double-buffering the
// halo (next tile's global loads fly during this tile's MFMAs and are written to the other buffer at its end) and the output
// staging (tile k-1's 16-byte global stores, with the fused mask / residual epilogue, are issued inside tile k's MFMA loop).
// One barrier per tile.  128-byte pixel rows are XOR-swizzled by (halo column & 7): conflict-free ds_read_b128 for every tap.
// Channel tails use the K=16 MFMA (48 = 32 + 16, and the thin 3/9/12-channel inputs are a single K=16 chunk).
__device__ const uint4 zt_zero_chunk = {0u, 0u, 0u, 0u};        // LDS-DMA source of the halo's out-of-image pixels

__device__ __forceinline__ zt_f32x4 zt_mfma_bf16_k16(zt_s16x4 a, zt_s16x4 b, zt_f32x4 c) {
  // D = A(16x16) * B(16x16) + C: lane l holds A[row l&15][k = 4(l>>4)+j], B[k = 4(l>>4)+j][col l&15], j = 0..3
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// wave w of 8: row pair w >> 1; COSPLIT: couts [NQ*16*(w&1), +NQ*16) of 2*NQ*16, both 16-pixel halves (NM == 2)
//                               else   : all NQ*16 couts, 16-pixel half (w & 1) (NM == 1)
// RT = tile rows = waves per workgroup: 8 (one workgroup per CU, double-buffered staging) or 4 (two independent workgroups per
// CU, single staging buffer: one workgroup's epilogue / barrier / DMA issue overlaps the other's MFMA loop; not for EPI, whose
// aux operand needs the second staging buffer)
// STATS: the train-mode BatchNorm that follows the layer (model.py:62) needs the per-channel sum and sum of squares of the output
// over all pixels: they are accumulated from the staged (bf16-rounded, i.e. exactly the stored) values in the store phase --
// per-thread over its chunks, butterfly over the 8 lanes of a wave that own the same channel octet, then into a per-wave LDS
// table owned lane by lane (no atomics: fixed summation order, bit-reproducible) -- and written once per workgroup at the end;
// zt_norm_finalize_f32 reduces the [grid][2][Cout] partials.  Replaces a separate 265 MB read pass per Enhancer block.
template <int NQ, int NM, bool COSPLIT, int C32, int C16, bool EPI, int RT, bool STATS = false>
__global__ void __launch_bounds__(64 * RT, RT == 4 ? 2 : 1) conv_rs_bf16_kernel(ConvArgsH a, int ntiles) {
  static_assert(!STATS || RT == 4, "statistics are fused into the 4-row variants");
  static_assert(!STATS || !EPI || COSPLIT, "backward statistics ride in the 64-cout data-gradient variant (aux fetched by the store phase)");
  constexpr bool BSTATS = STATS && EPI;                         // BatchNorm-backward sums instead of forward statistics
  constexpr int RTH = RT, NTHR = 64 * RT, NST = RT == 8 ? 2 : 1;
  // fused aux operand (activation mask / residual): the 8-row form DMAs its tile into the idle staging buffer (AUXL); the 4-row
  // form has no second staging buffer and reads it in accumulator layout (8 bytes per lane and 16x16 block) half a loop ahead
  // (AUXD: 48 couts); with 64 couts (144 VGPRs of weights) that spills, so there the aux chunk is fetched by the store phase at
  // the start of the next tile, where no accumulator is live (AUXS; the other workgroup of the CU covers the exposed latency)
  constexpr bool AUXL = EPI && RT == 8, AUXD = EPI && RT == 4 && !COSPLIT, AUXS = EPI && RT == 4 && COSPLIT;
  constexpr int IR = RTH + 2, IC = TW + 2;
  constexpr int KC = C32 * 32 + C16 * 16;                       // input channels staged per pixel
  constexpr int PE = KC > 32 ? 64 : (KC > 16 ? 32 : 16);        // LDS elements per pixel; only the 128-byte rows need the swizzle
  constexpr bool SWZ = PE == 64;
  constexpr int NCHK = PE / 8;                                  // 16-byte chunks per pixel
  constexpr bool GLDS = PE == 64;                               // full 128-byte rows go global -> LDS by DMA: no staging registers
  constexpr int NPF = GLDS ? 1 : (IR * IC * NCHK + NTHR - 1) / NTHR;
  constexpr int NGL = (IR * IC * 8 + NTHR - 1) / NTHR;                // LDS-DMA wave-instructions per wave and tile
  constexpr int CW = (COSPLIT ? 2 : 1) * NQ * 16;               // couts of the layer (== a.Cout)
  constexpr int CH8 = CW / 8;
  constexpr bool SWZO = CW == 64;
  constexpr int NOUT = RTH * TW * CH8 / NTHR;
  static_assert(RTH * TW * CH8 % NTHR == 0 && NTHR % NCHK == 0, "tile geometry");
  __shared__ __attribute__((aligned(16))) zt_bf16 xs[2][IR * IC * PE];
  __shared__ __attribute__((aligned(16))) zt_bf16 st[NST][RTH * TW * CW];
  __shared__ float bias_s[CW];
  __shared__ __attribute__((aligned(16))) float stat_s[STATS ? RT * 2 * CW : 4];      // [wave][octet][sum 8 | sumsq 8]
  __shared__ __attribute__((aligned(16))) float bn_s[BSTATS ? 3 * CW : 4];             // BSTATS: [scale | shift | mean][channel]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int rp = wave >> 1, sel = wave & 1;
  if constexpr (STATS) {
    for (int e = tid; e < RT * 2 * CW; e += NTHR) stat_s[e] = 0.f;
  }
  if constexpr (BSTATS) {
    for (int e = tid; e < 3 * CW; e += NTHR) bn_s[e] = e < CW ? a.bn_scale[e] : (e < 2 * CW ? a.bn_shift[e - CW] : a.bn_mean[e - 2 * CW]);
  }
  const int q0 = COSPLIT ? sel * NQ : 0, m0 = COSPLIT ? 0 : sel;

  if (tid < CW) bias_s[tid] = a.bias ? a.bias[tid] : 0.f;

  // A fragments: weights [tap][CoutP][ldk], this wave's couts, all taps and channel chunks -- resident for the whole launch
  zt_s16x8 w32[9][C32 > 0 ? C32 : 1][NQ];
  zt_s16x4 w16[9][NQ];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const zt_bf16* wr = a.w + ((size_t)tap * a.CoutP + (q0 + q) * 16 + l15) * a.ldk;
#pragma unroll
      for (int c = 0; c < C32; ++c) {
        const int kk = c * 32 + l4 * 8;
        w32[tap][c][q] = kk < a.ldk ? *reinterpret_cast<const zt_s16x8*>(wr + kk) : (zt_s16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
      if (C16) {
        const int kk = C32 * 32 + l4 * 4;
        w16[tap][q] = kk < a.ldk ? *reinterpret_cast<const zt_s16x4*>(wr + kk) : (zt_s16x4){0, 0, 0, 0};
      }
    }

  // XCD-aware tile order: workgroup b runs on XCD b % 8, so within every round of gridDim tiles XCD x takes the x-th run of
  // gridDim / 8 consecutive indices, and indices walk the image in bands of 4 tile rows, column-major inside a band: an XCD's
  // 32 tiles of a round form an 8 x 4 block whose interior halos are shared in that XCD's L2 instead of re-fetched from HBM.
  const int G = gridDim.x;
  const int pb = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int n_my = pb < ntiles ? (ntiles - 1 - pb) / G + 1 : 0;
  // the (ty, tx) of this workgroup's k-th tile: computed once into a small LDS table (the divisions are ~40 instructions and
  // three phases per tile need the coordinates), recomputed only beyond the table
  constexpr int TTAB = 256;
  __shared__ int tile_s[TTAB];
  auto tile_calc = [&](int k, int& ty, int& tx) {
    const int idx = pb + k * G;
    const int band = idx / (4 * a.tilesX), r = idx - band * 4 * a.tilesX;
    const int rows = a.tilesY - band * 4 < 4 ? a.tilesY - band * 4 : 4;
    tx = r / rows;
    ty = band * 4 + r - tx * rows;
  };
  for (int k = tid; k < n_my && k < TTAB; k += NTHR) {
    int ty, tx;
    tile_calc(k, ty, tx);
    tile_s[k] = (ty << 16) | tx;
  }
  auto tile_xy = [&](int k, int& ty, int& tx) {
    if (k < TTAB) {
      const int v = __builtin_amdgcn_readfirstlane(tile_s[k]);
      ty = v >> 16;
      tx = v & 0xFFFF;
    } else {
      tile_calc(k, ty, tx);
    }
  };

  // halo slot e = tid + 512 i -> pixel e / NCHK (row-major in the IR x IC halo), chunk e % NCHK == tid % NCHK for every i
  uint4 pf[NPF];
  const int hq = tid & (NCHK - 1);
  const int hq8 = hq * 8 + 8 <= a.ldx ? hq * 8 : a.ldx - 8;     // never read past the pixel's channels; masked below
  // LDS-DMA form: wave-instruction (8 i + wave) fills positions [64 (8 i + wave), +64) of the linear image; position e holds
  // pixel e / 8, logical chunk (e % 8) ^ (column & 7) -- the swizzle is applied to the source address.  Needs Cin % 8 == 0.
  // Interior tiles (the halo lies inside the image: ~95 % of them) take a precomputed per-slot offset relative to the tile
  // origin -- one add per DMA; border tiles recompute the clamped / zero-filled addresses.  The offsets cost NGL registers,
  // which the EPI instantiations do not have: they always take the general path.
  constexpr bool FASTSLOT = GLDS && !EPI;
  int soff[FASTSLOT ? NGL : 1];
  if constexpr (FASTSLOT) {
#pragma unroll
    for (int i = 0; i < NGL; ++i) {
      const int e = (i * RT + wave) * 64 + lane;
      const int p = e >> 3, col = p % IC;
      const int cj = (e & 7) ^ (col & 7);
      soff[i] = cj * 8 < a.Cin ? ((p / IC) * a.W + col) * a.ldx + cj * 8 : -1;       // -1: channel chunk beyond Cin -> zeros
    }
  }
  auto glds_halo = [&](int k) {
    int ty, tx;
    tile_xy(k, ty, tx);
    const int gy0 = ty * RTH - 1, gx0 = tx * TW - 1;
    zt_bf16* xb = xs[k & 1];
    if constexpr (FASTSLOT) {
      if (gy0 >= 0 && gy0 + IR <= a.H && gx0 >= 0 && gx0 + IC <= a.W) {
        const zt_bf16* base = a.x + (unsigned)((gy0 * a.W + gx0) * a.ldx);
#pragma unroll
        for (int i = 0; i < NGL; ++i) {
          const void* src = soff[i] >= 0 ? (const void*)(base + soff[i]) : (const void*)&zt_zero_chunk;
          if (i * NTHR + NTHR - 1 < IR * IC * 8 || (i * RT + wave) * 64 + lane < IR * IC * 8) zt_glds16(src, xb + (i * RT + wave) * 512);
        }
        return;
      }
    }
    int ln = lane;
    ZT_OPAQUE(ln);                                              // recompute the slot geometry per tile instead of keeping it in registers
    if constexpr (!FASTSLOT && !BSTATS) {                       // (BSTATS: the second code path costs it two spilled registers)
      // EPI instantiations: interior tiles without the precomputed offsets -- the same address as the general path minus its
      // clamps, bounds tests and selects (35 -> ~12 vector instructions per DMA; the kernel is issue-co-limited, section 5)
      if (gy0 >= 0 && gy0 + IR <= a.H && gx0 >= 0 && gx0 + IC <= a.W) {
        const zt_bf16* base = a.x + (unsigned)((gy0 * a.W + gx0) * a.ldx);
#pragma unroll
        for (int i = 0; i < NGL; ++i) {
          const int e = (i * RT + wave) * 64 + ln;
          const int p = e >> 3, row = p / IC, col = p - row * IC;
          const int cj = (e & 7) ^ (col & 7);
          const void* src = cj * 8 < a.Cin ? (const void*)(base + (unsigned)((row * a.W + col) * a.ldx + cj * 8)) : (const void*)&zt_zero_chunk;
          if (i * NTHR + NTHR - 1 < IR * IC * 8 || e < IR * IC * 8) zt_glds16(src, xb + (i * RT + wave) * 512);
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < NGL; ++i) {
      const int e = (i * RT + wave) * 64 + ln;
      const int p = e >> 3, col = p % IC;
      const int cj = (e & 7) ^ (col & 7);
      const int gy = gy0 + p / IC, gx = gx0 + col;
      const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && cj * 8 < a.Cin;
      const int gyc = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
      const int cc = cj * 8 + 8 <= a.ldx ? cj * 8 : 0;
      const zt_bf16* s1 = a.x + (unsigned)((gyc * a.W + gxc) * a.ldx + cc);
      const void* src = in ? (const void*)s1 : (const void*)&zt_zero_chunk;
      if (i * NTHR + NTHR - 1 < IR * IC * 8 || e < IR * IC * 8) zt_glds16(src, xb + (i * RT + wave) * 512);
    }
  };
  auto load_halo = [&](int k) {
    if constexpr (GLDS) {
      glds_halo(k);
      return;
    }
    int ty, tx;
    tile_xy(k, ty, tx);
    const int gy0 = ty * RTH - 1, gx0 = tx * TW - 1;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int p = (tid + i * NTHR) / NCHK;
      int gy = gy0 + p / IC, gx = gx0 + p % IC;                 // out-of-image slots read a clamped address, zeroed when written
      gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
      gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
      pf[i] = *reinterpret_cast<const uint4*>(a.x + (unsigned)((gy * a.W + gx) * a.ldx + hq8));
    }
  };
  auto write_halo = [&](int k) {
    if constexpr (GLDS) return;
    int ty, tx;
    tile_xy(k, ty, tx);
    const int gy0 = ty * RTH - 1, gx0 = tx * TW - 1;
    const int nv = a.Cin - hq * 8;                              // valid channels of this thread's chunk: padding lanes are not trusted
    const unsigned k0 = nv >= 2 ? ~0u : (nv == 1 ? 0xFFFFu : 0u), k1 = nv >= 4 ? ~0u : (nv == 3 ? 0xFFFFu : 0u);
    const unsigned k2 = nv >= 6 ? ~0u : (nv == 5 ? 0xFFFFu : 0u), k3 = nv >= 8 ? ~0u : (nv == 7 ? 0xFFFFu : 0u);
    zt_bf16* xb = xs[k & 1];
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int e = tid + i * NTHR, p = e / NCHK, col = p % IC;
      const int gy = gy0 + p / IC, gx = gx0 + col;
      const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      uint4 v = pf[i];
      v.x = in ? (v.x & k0) : 0u;
      v.y = in ? (v.y & k1) : 0u;
      v.z = in ? (v.z & k2) : 0u;
      v.w = in ? (v.w & k3) : 0u;
      if (e < IR * IC * NCHK) *reinterpret_cast<uint4*>(xb + p * PE + ((SWZ ? (hq ^ (col & 7)) : hq) * 8)) = v;
    }
  };

  // staged outputs of tile k -> global: chunk e = tid + 512 i -> pixel e / CH8 of the 8 x 32 tile, couts 8 (e % CH8)..+8.
  // EPI: the aux tile (activation mask / residual) is DMA'd into the staging buffer that is idle during this tile's MFMA loop;
  // chunk e is fetched by the very lane that consumes it (e = 64 (8 i + wave) + lane), so only that lane's vmcnt matters.
  auto glds_aux = [&](int k, int buf) {
    int ty, tx;
    tile_xy(k, ty, tx);
    const int oy0 = ty * RTH, ox0 = tx * TW;
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const int e = tid + i * NTHR, pl = e / CH8, ch = e % CH8;
      int oy = oy0 + pl / TW, ox = ox0 + pl % TW;
      oy = oy >= a.Ho ? a.Ho - 1 : oy;
      ox = ox >= a.Wo ? a.Wo - 1 : ox;
      zt_glds16(a.aux + (unsigned)((oy * a.Wo + ox) * a.ldaux + ch * 8), st[buf] + (i * RT + wave) * 512);
    }
  };
  // AUXS: the aux chunks (and the BatchNorm pre-activations of BSTATS) of tile k straight from global, 16 bytes per lane, all in
  // flight -- issued BEFORE the next halo's DMAs: vector-memory operations complete in order, so the store phase's wait for these
  // loads would otherwise also wait for the whole halo that was issued in front of them (a full HBM round trip per tile)
  // (BSTATS keeps 32 registers of operands per lane: holding them across the DMA address arithmetic spills, and a scratch reload
  // is itself a vector-memory operation behind the DMAs -- there the loads stay inside the store phase, behind the halo issue)
  constexpr bool AUXE = AUXS && !BSTATS;
  static_assert(!AUXE || NOUT == 4, "hidden aux loads are waited for four at a time");
  zt_u32x4 uxe[AUXE ? NOUT : 1];
  auto aux_fetch = [&](int k) {
    int ty, tx;
    tile_xy(k, ty, tx);
    const int oy0 = ty * RTH, ox0 = tx * TW;
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const int e = tid + i * NTHR, pl = e / CH8, ch = e % CH8;
      int oy = oy0 + pl / TW, ox = ox0 + pl % TW;
      oy = oy >= a.Ho ? a.Ho - 1 : oy;
      ox = ox >= a.Wo ? a.Wo - 1 : ox;
      ZT_HIDDEN_LD16(uxe[AUXE ? i : 0], a.aux + (unsigned)((oy * a.Wo + ox) * a.ldaux + ch * 8));
    }
  };
  auto store_tile = [&](int k, bool halo_in_flight = false) {
    int ty, tx;
    tile_xy(k, ty, tx);
    const int oy0 = ty * RTH, ox0 = tx * TW;
    const zt_bf16* sb = st[k & (NST - 1)];
    const zt_bf16* ab = st[(k + 1) & (NST - 1)];
    const float neg = a.epi == 1 ? 0.2f : 0.f;
    if (AUXL) {                                                 // this lane's aux DMA has landed; the halo DMAs issued after it may still fly
      if (halo_in_flight) zt_wait_vmcnt<GLDS ? NGL : NPF>();
      else zt_wait_vmcnt0();
    }
    uint4 v[NOUT], ux[BSTATS ? NOUT : 1], zx[BSTATS ? NOUT : 1];
    float ssum[STATS ? 8 : 1], ssq[STATS ? 8 : 1];
    if constexpr (STATS) {
#pragma unroll
      for (int c = 0; c < 8; ++c) ssum[c] = ssq[c] = 0.f;
    }
    if constexpr (AUXE) {
      // the aux loads of aux_fetch(k) were issued before the NGL halo DMAs of tile k + 2 (the last of which a wave may skip): all
      // but those may still be in flight
      // (one register-tied wait on every path -- two alternatives would meet in copies of the still pending registers; the drain
      // for the tiles without a following halo is a separate, untied statement in front of it)
      if (!(GLDS && k + 2 < n_my && !(a.dbg & 4)) || (a.dbg & 128)) ZT_WAIT_HIDDEN_DMA();
      ZT_HIDDEN_WAIT4(GLDS ? NGL - 1 : 0, uxe[0], uxe[AUXE ? 1 : 0], uxe[AUXE ? 2 : 0], uxe[AUXE ? 3 : 0]);
    }
    if constexpr (BSTATS) {                                     // aux chunks + pre-activations straight from global (16 bytes per lane), all in flight
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        const int e = tid + i * NTHR, pl = e / CH8, ch = e % CH8;
        int oy = oy0 + pl / TW, ox = ox0 + pl % TW;
        oy = oy >= a.Ho ? a.Ho - 1 : oy;
        ox = ox >= a.Wo ? a.Wo - 1 : ox;
        ux[BSTATS ? i : 0] = *reinterpret_cast<const uint4*>(a.aux + (unsigned)((oy * a.Wo + ox) * a.ldaux + ch * 8));
        zx[BSTATS ? i : 0] = *reinterpret_cast<const uint4*>(a.zprev + (unsigned)((oy * a.Wo + ox) * a.ldz + ch * 8));
      }
    }
    if (!AUXL) {                                                // AUXL runs mid-loop with every accumulator live: one chunk at a time
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        const int e = tid + i * NTHR, pl = e / CH8, ch = e % CH8;
        v[i] = *reinterpret_cast<const uint4*>(sb + pl * CW + ((SWZO ? (ch ^ (pl & 7)) : ch) * 8));
      }
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const int e = tid + i * NTHR, pl = e / CH8, ch = e % CH8;
      const int oy = oy0 + pl / TW, ox = ox0 + pl % TW;
      uint4 o;
      if (AUXL || AUXS) {
        uint4 u;
        if constexpr (AUXL) {
          o = *reinterpret_cast<const uint4*>(sb + pl * CW + ((SWZO ? (ch ^ (pl & 7)) : ch) * 8));
          u = *reinterpret_cast<const uint4*>(ab + e * 8);
        } else {
          o = v[i];
          if constexpr (BSTATS) u = ux[BSTATS ? i : 0];
          else u = make_uint4(uxe[AUXE ? i : 0].x, uxe[AUXE ? i : 0].y, uxe[AUXE ? i : 0].z, uxe[AUXE ? i : 0].w);
        }
        const unsigned vv[4] = {o.x, o.y, o.z, o.w}, uu[4] = {u.x, u.y, u.z, u.w};
        unsigned oo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float f0 = zt_u2f(vv[j] << 16), f1 = zt_u2f(vv[j] & 0xFFFF0000u);
          const float g0 = zt_u2f(uu[j] << 16), g1 = zt_u2f(uu[j] & 0xFFFF0000u);
          if (a.epi == 3) { f0 += g0; f1 += g1; }
          else { f0 *= (g0 > 0.f ? 1.f : neg); f1 *= (g1 > 0.f ? 1.f : neg); }
          oo[j] = zt_f2bf2(f0, f1);
        }
        o = make_uint4(oo[0], oo[1], oo[2], oo[3]);
        if constexpr (AUXL) __builtin_amdgcn_sched_barrier(0);  // keep the chunks sequential (register pressure)
      } else {
        o = v[i];
      }
      if (oy < a.Ho && ox < a.Wo) *reinterpret_cast<uint4*>((zt_bf16*)a.y + (unsigned)((oy * a.Wo + ox) * a.ldy + ch * 8)) = o;
      if constexpr (STATS && !BSTATS) {
        if (oy < a.Ho && ox < a.Wo) {
          const unsigned ow[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float f0 = zt_u2f(ow[j] << 16), f1 = zt_u2f(ow[j] & 0xFFFF0000u);
            ssum[2 * j] += f0;
            ssum[2 * j + 1] += f1;
            ssq[2 * j] += f0 * f0;
            ssq[2 * j + 1] += f1 * f1;
          }
        }
      }
      if constexpr (BSTATS) {
        // the stored (bf16-rounded) gradient of the previous block's output, masked by that block's ReLU, summed plain and against
        // its centred pre-activation: what zt_bn_bwd_reduce computes in a pass of its own over the same two tensors
        if (oy < a.Ho && ox < a.Wo) {
          const unsigned ow[4] = {o.x, o.y, o.z, o.w};
          const uint4 zq = zx[BSTATS ? i : 0];
          const unsigned zw[4] = {zq.x, zq.y, zq.z, zq.w};
          const float4 sa = *reinterpret_cast<const float4*>(bn_s + ch * 8), sb2 = *reinterpret_cast<const float4*>(bn_s + ch * 8 + 4);
          const float4 ha = *reinterpret_cast<const float4*>(bn_s + CW + ch * 8), hb = *reinterpret_cast<const float4*>(bn_s + CW + ch * 8 + 4);
          const float4 ma = *reinterpret_cast<const float4*>(bn_s + 2 * CW + ch * 8), mb = *reinterpret_cast<const float4*>(bn_s + 2 * CW + ch * 8 + 4);
          const float scv[8] = {sa.x, sa.y, sa.z, sa.w, sb2.x, sb2.y, sb2.z, sb2.w}, shv[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
          const float muv[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float f0 = zt_u2f(ow[j] << 16), f1 = zt_u2f(ow[j] & 0xFFFF0000u);
            const float z0 = zt_u2f(zw[j] << 16), z1 = zt_u2f(zw[j] & 0xFFFF0000u);
            const float g0 = z0 * scv[2 * j] + shv[2 * j] > 0.f ? f0 : 0.f, g1 = z1 * scv[2 * j + 1] + shv[2 * j + 1] > 0.f ? f1 : 0.f;
            ssum[2 * j] += g0;
            ssum[2 * j + 1] += g1;
            ssq[2 * j] += g0 * (z0 - muv[2 * j]);
            ssq[2 * j + 1] += g1 * (z1 - muv[2 * j + 1]);
          }
        }
      }
    }
    if constexpr (STATS) {
      // chunk e = tid + NTHR i has channel octet tid % 8 for every i: lanes l, l^8, l^16, l^32 of a wave share it
      // Reduce-scatter over those 8 lanes instead of a full butterfly: every stage hands HALF of the still-live values to the
      // partner and keeps the sums of the other half (8 + 4 + 2 = 14 cross-lane moves instead of 48); each lane ends up owning 2 of
      // the octet's 16 sums -- index 8 (lane>>5 & 1) + 4 (lane>>4 & 1) + 2 (lane>>3 & 1) + {0, 1} -- and adds them to its own two
      // slots of the per-wave table (fixed order: bit-reproducible).
      const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
      float k8[8], k4[4], k2[2];
#pragma unroll
      for (int c = 0; c < 8; ++c) k8[c] = (h5 ? ssq[c] : ssum[c]) + __shfl_xor(h5 ? ssum[c] : ssq[c], 32);
#pragma unroll
      for (int c = 0; c < 4; ++c) k4[c] = (h4 ? k8[c + 4] : k8[c]) + __shfl_xor(h4 ? k8[c] : k8[c + 4], 16);
#pragma unroll
      for (int c = 0; c < 2; ++c) k2[c] = (h3 ? k4[c + 2] : k4[c]) + __shfl_xor(h3 ? k4[c] : k4[c + 2], 8);
      float* t = stat_s + (wave * CH8 + (lane & 7)) * 16 + (h5 ? 8 : 0) + (h4 ? 4 : 0) + (h3 ? 2 : 0);
      t[0] += k2[0];
      t[1] += k2[1];
    }
  };

  __syncthreads();                                              // tile table and bias visible
  if (n_my > 0) {
    load_halo(0);
    write_halo(0);
  }
  __syncthreads();

  const float slope = a.act == 0 ? 1.f : (a.act == 1 ? 0.f : 0.2f);     // none / ReLU / LeakyReLU(0.2) == max(v, slope*v)
  // lane part of the pixel fragment address for kx = 0..2 (halo row and 16-pixel half are immediate offsets)
  int xoff32[3][C32 > 0 ? C32 : 1], xoff16[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int col = l15 + kx;                                   // + 16 m: does not change col & 7
#pragma unroll
    for (int c = 0; c < C32; ++c) xoff32[kx][c] = col * PE + (SWZ ? (((c * 4 + l4) ^ (col & 7)) * 8) : (c * 32 + l4 * 8));
    xoff16[kx] = col * PE + (SWZ ? (((C32 * 4 + (l4 >> 1)) ^ (col & 7)) * 8 + (l4 & 1) * 4) : (C32 * 32 + l4 * 4));
  }

  for (int k = 0; k < n_my; ++k) {
    const zt_bf16* xb = xs[k & 1] + ((2 * rp) * IC + m0 * 16) * PE;
    zt_f32x4 acc[2][NM][NQ];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[r][m][q] = *reinterpret_cast<const zt_f32x4*>(&bias_s[(q0 + q) * 16 + l4 * 4]);   // bias

    if (AUXL) {
      if (k >= 1) glds_aux(k - 1, k & 1);
      if (k + 1 < n_my && !(a.dbg & 4)) load_halo(k + 1);
    } else {
      if constexpr (AUXE) {
        if (k >= 1 && !(a.dbg & 2) && !(a.dbg & 128)) aux_fetch(k - 1);
      }
      if (k + 1 < n_my && !(a.dbg & 4)) load_halo(k + 1);
      if constexpr (AUXE) {                                     // A/B form (ZT_RS_AUX_LATE=1): aux loads behind the halo issue, full drain
        if (k >= 1 && !(a.dbg & 2) && (a.dbg & 128)) aux_fetch(k - 1);
      }
      if (k >= 1 && !(a.dbg & 2)) store_tile(k - 1);
      if constexpr (NST == 1) ZT_LDS_BARRIER();                 // single staging buffer: every wave has read tile k-1 before tile k is staged
    }
    uint2 au[AUXD ? 2 : 1][AUXD ? NM : 1][AUXD ? NQ : 1];       // AUXD: this lane's aux values, in accumulator layout

    // steps: halo row h (0..3) x kx x channel chunk; each step's fragments serve output rows r with ky = h - r in [0, 2]
    constexpr int NCK = C32 + C16;
    constexpr int NSTEP = 4 * 3 * NCK;
    zt_s16x8 xa[2][NM];
    zt_s16x4 xt[2][NM];
#define ZT_LOADX(bufi, step)                                                                                          \
  {                                                                                                                   \
    constexpr int h_ = (step) / (3 * NCK), kx_ = ((step) / NCK) % 3, c_ = (step) % NCK;                               \
    _Pragma("unroll") for (int m = 0; m < NM; ++m) {                                                                  \
      if constexpr (c_ < C32) xa[bufi][m] = *reinterpret_cast<const zt_s16x8*>(xb + (h_ * IC + m * 16) * PE + xoff32[kx_][c_ < C32 ? c_ : 0]); \
      else xt[bufi][m] = *reinterpret_cast<const zt_s16x4*>(xb + (h_ * IC + m * 16) * PE + xoff16[kx_]);              \
    }                                                                                                                 \
  }
    ZT_LOADX(0, 0)
    if (!(a.dbg & 1))
    zt_static_for<0, NSTEP>([&](auto step_c) {
      constexpr int step = decltype(step_c)::value;
      constexpr int cur = step & 1;
      constexpr int h = step / (3 * NCK), kx = (step / NCK) % 3, c = step % NCK;
      if constexpr (step + 1 < NSTEP) ZT_LOADX(cur ^ 1, step + 1)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int ky = h - r;
        if (ky >= 0 && ky <= 2) {
#pragma unroll
          for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
              if constexpr (c < C32) acc[r][m][q] = zt_mfma_bf16(w32[ky * 3 + kx][c < C32 ? c : 0][q], xa[cur][m], acc[r][m][q]);
              else acc[r][m][q] = zt_mfma_bf16_k16(w16[ky * 3 + kx][q], xt[cur][m], acc[r][m][q]);
            }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (AUXD && step == NSTEP / 2 - 1) {            // aux in accumulator layout: half a loop of latency cover
        int ty, tx;
        tile_xy(k, ty, tx);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            int oy = ty * RTH + 2 * rp + r, ox = tx * TW + (m0 + m) * 16 + l15;
            oy = oy >= a.Ho ? a.Ho - 1 : oy;
            ox = ox >= a.Wo ? a.Wo - 1 : ox;
            const zt_bf16* ap = a.aux + (unsigned)((oy * a.Wo + ox) * a.ldaux + q0 * 16 + l4 * 4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) au[AUXD ? r : 0][AUXD ? m : 0][AUXD ? q : 0] = *reinterpret_cast<const uint2*>(ap + q * 16);
          }
      }
      if constexpr (AUXL && step == NSTEP / 2 - 1) {            // aux has had half of the loop to arrive
        if (k >= 1) store_tile(k - 1, k + 1 < n_my);
        ZT_LDS_BARRIER();                                       // aux consumed (LDS reads only: the halo DMAs keep flying): the rest of the loop may end in staging writes to that buffer
      }
    });
#undef ZT_LOADX
    if (k + 1 < n_my) write_halo(k + 1);

    // accumulators -> staging (bias, alpha, activation, bf16): lane holds couts 4 l4 .. +3 of 16-cout block q for pixel l15
    zt_bf16* sb = st[k & (NST - 1)];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int cb = (q0 + q) * 16 + l4 * 4;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[r][m][q][j];
          // alpha == 1 on this path (rs_ok): hipcc had if-converted `if (alpha != 1) v *= alpha` into 2 packed multiplies + 4 selects
          // per 4 values, executed always; and fmaxf() on MFMA outputs costs a canonicalising v_max per operand -- ZT_VMAX is the bare
          // instruction.  The epilogue is the largest share of this VALU-issue co-limited kernel's 2.7 VALU per MFMA (section 5).
          if (a.act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ZT_VMAX(v[j], slope * v[j]);
          }
          if constexpr (AUXD) {                                 // fused epilogue on the fp32 values: one rounding
            const uint2 u = au[AUXD ? r : 0][AUXD ? m : 0][AUXD ? q : 0];
            const float g[4] = {zt_u2f(u.x << 16), zt_u2f(u.x & 0xFFFF0000u), zt_u2f(u.y << 16), zt_u2f(u.y & 0xFFFF0000u)};
            const float neg = a.epi == 1 ? 0.2f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a.epi == 3 ? v[j] + g[j] : v[j] * (g[j] > 0.f ? 1.f : neg);
          }
          uint2 pk;
          pk.x = zt_f2bf2(v[0], v[1]);
          pk.y = zt_f2bf2(v[2], v[3]);
          const int pl = (2 * rp + r) * TW + (m0 + m) * 16 + l15;
          const int ch = cb >> 3;
          *reinterpret_cast<uint2*>(sb + pl * CW + ((SWZO ? (ch ^ (pl & 7)) : ch) * 8) + (cb & 4)) = pk;
        }
    __syncthreads();
  }
  if (n_my > 0) {
    if (AUXL) glds_aux(n_my - 1, n_my & 1);
    if constexpr (AUXE) aux_fetch(n_my - 1);
    store_tile(n_my - 1);
  }
  if constexpr (STATS) {
    __syncthreads();
    if (tid < 2 * CW) {                                         // stats[block][0: sum | 1: sum of squares][channel]
      const int which = tid / CW, c = tid % CW;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < RT; ++w) t += stat_s[(w * CH8 + (c >> 3)) * 16 + which * 8 + (c & 7)];
      a.stats[(size_t)blockIdx.x * 2 * CW + tid] = t;
    }
  }
}

int launch_conv_rs(ConvArgsH& a, hipStream_t stream) {
  // two 4-wave workgroups per CU on 4-row tiles (ZT_CONV_RS4=0: the one-workgroup 8-row form)
  static const int rt4 = getenv("ZT_CONV_RS4") ? atoi(getenv("ZT_CONV_RS4")) : 1;
  const int rt = rt4 ? 4 : 8;
  a.tilesX = zt_cdiv(a.Wo, TW);
  a.tilesY = zt_cdiv(a.Ho, rt);
  const int ntiles = a.tilesX * a.tilesY;
  const int maxg = rt == 4 ? 512 : 256;
  dim3 grid(ntiles < maxg ? ntiles : maxg), block(64 * rt);
  const int kc = a.Cin <= 16 ? 0 : (a.Cin > 48 ? 2 : 1);       // 0: one K=16 chunk, 1: 32 + 16, 2: 32 + 32
  if (a.stats) {                                                // fused BatchNorm statistics: the 64 -> 64 layer, 4-row tiles
    if (!(a.Cout == 64 && kc == 2 && (!a.epi || (a.epi == 3 && a.zprev)))) return ZT_EINVAL;
    a.tilesY = zt_cdiv(a.Ho, 4);
    const int nt4 = a.tilesX * a.tilesY;
    dim3 g4(nt4 < 512 ? nt4 : 512);
    if (a.epi) hipLaunchKernelGGL((conv_rs_bf16_kernel<2, 2, true, 2, 0, true, 4, true>), g4, dim3(256), 0, stream, a, nt4);      // data gradient + residual + BN-backward sums
    else hipLaunchKernelGGL((conv_rs_bf16_kernel<2, 2, true, 2, 0, false, 4, true>), g4, dim3(256), 0, stream, a, nt4);           // forward + BN statistics
    return 0;
  }
#define ZT_RS(nq, nm, cs, c32, c16)                                                                                            \
  {                                                                                                                            \
    if (a.epi && rt == 4) {                                                                                                    \
      hipLaunchKernelGGL((conv_rs_bf16_kernel<nq, nm, cs, c32, c16, true, 4>), grid, block, 0, stream, a, ntiles);              \
      return 0;                                                                                                                \
    }                                                                                                                          \
    if (a.epi) hipLaunchKernelGGL((conv_rs_bf16_kernel<nq, nm, cs, c32, c16, true, 8>), grid, block, 0, stream, a, ntiles); \
    else if (rt == 4) hipLaunchKernelGGL((conv_rs_bf16_kernel<nq, nm, cs, c32, c16, false, 4>), grid, block, 0, stream, a, ntiles); \
    else hipLaunchKernelGGL((conv_rs_bf16_kernel<nq, nm, cs, c32, c16, false, 8>), grid, block, 0, stream, a, ntiles);          \
    return 0;                                                                                                                  \
  }
  if (a.Cout == 64 && kc == 2) ZT_RS(2, 2, true, 2, 0)
  if (a.Cout == 64 && kc == 0) ZT_RS(2, 2, true, 0, 1)
  if (a.Cout == 48 && kc == 1) ZT_RS(3, 1, false, 1, 1)
  if (a.Cout == 48 && kc == 0) ZT_RS(3, 1, false, 0, 1)
#undef ZT_RS
  return ZT_EINVAL;
}

// ---- bf16 weight gradient.  K = pixels: the MFMA needs 8 consecutive PIXELS per lane for one channel, i.e. the
// transpose of the NHWC tile; ds_read_b64_tr_b16 delivers exactly that from a [pixel][channel] LDS image, so staging is a
// plain 16-byte copy and tap shifts are row shifts (alignment preserved).
struct WgradArgsH {
  const zt_bf16* x;
  const zt_bf16* dz;
  float* slab;
  int H, W, Cin, ldx, Cout, lddz;
  int tilesX, ntiles;
  const zt_bf16* mask;         // MASK: dz is taken as dz * [mask > 0] (the ReLU that follows the layer, folded in)
  int ldmask;
};

constexpr int HTW = 32;                 // tile = HTH rows x 32 pixels; HTH = 2 * NW (4 or 8): 8-wave workgroups keep twice the bytes in flight

// NW waves per workgroup share the (tap, ci-tile) pairs; 8 for the 64x64 layer so that accumulators + staging registers stay <= 128
template <int KH, int KW, int CT, int NT, int NW, bool MASK = false>
__global__ void __launch_bounds__(NW * 64, (NW == 4 && CT == 4 && NT == 4) ? 2 : 1) wgrad_mfma_bf16_kernel(WgradArgsH a) {
  constexpr int NTHR = NW * 64, HTH = NW;
  constexpr int IR = HTH + KH - 1, IC = HTW + KW - 1;
  constexpr int CIP = CT * 16 + 8, COP = NT * 16 + 8;
  constexpr int NPAIR = KH * KW * CT;
  constexpr int PPW = (NPAIR + NW - 1) / NW;
  constexpr int padH = (KH - 1) / 2, padW = (KW - 1) / 2;
  __shared__ __attribute__((aligned(16))) zt_bf16 xs[IR * IC * CIP];
  __shared__ __attribute__((aligned(16))) zt_bf16 zs[HTH * HTW * COP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g8 = (lane >> 4) * 8;
  const int trq = l15 >> 2, trp = (l15 & 3) * 4;       // this lane's row / column quad inside a transposing 4x16 block

  zt_f32x4 acc[PPW][NT];
#pragma unroll
  for (int p = 0; p < PPW; ++p)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[p][q] = (zt_f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NPART = NTHR / (NT * 16);
  const int bco = tid % (NT * 16), bpart = tid / (NT * 16);
  float bsum = 0.f;

  // global -> registers -> LDS staging, software-pipelined: the next tile's loads are issued before this tile's MFMAs and land
  // while they run.  Loads are unconditional (clamped addresses); image borders and channel tails are masked when written.
  constexpr int NXL = (IR * IC * CT * 2 + NTHR - 1) / NTHR, NZL = (HTH * HTW * NT * 2 + NTHR - 1) / NTHR;
  uint4 px[NXL], pz[NZL], pm[MASK ? NZL : 1];
  auto relu_keep = [](unsigned g, unsigned m) {                  // two packed bf16: keep g where the activation m is > 0
    const unsigned lo = ((m & 0x8000u) == 0u && (m & 0x7FFFu) != 0u) ? 0xFFFFu : 0u;
    const unsigned hi = ((m & 0x80000000u) == 0u && (m & 0x7FFF0000u) != 0u) ? 0xFFFF0000u : 0u;
    return g & (lo | hi);
  };
  auto chan_mask = [](uint4 v, int nv, bool in) {               // keep the first nv (of 8) bf16 lanes
    const unsigned m0 = nv >= 2 ? ~0u : (nv == 1 ? 0xFFFFu : 0u), m1 = nv >= 4 ? ~0u : (nv == 3 ? 0xFFFFu : 0u);
    const unsigned m2 = nv >= 6 ? ~0u : (nv == 5 ? 0xFFFFu : 0u), m3 = nv >= 8 ? ~0u : (nv == 7 ? 0xFFFFu : 0u);
    v.x = in ? (v.x & m0) : 0u;
    v.y = in ? (v.y & m1) : 0u;
    v.z = in ? (v.z & m2) : 0u;
    v.w = in ? (v.w & m3) : 0u;
    return v;
  };
  // Interior tiles (halo inside the image, full channel octets: ~95 % of the tiles at 1080p) take a uniform fast path without the
  // per-slot clamps, bounds tests and channel masks (no extra registers: the slot's pixel / channel decomposition is recomputed).
  // (thin-input variants, CT == 1, measured 10-20 % slower with the extra path: they keep the general one)
  constexpr bool FASTP = CT >= 3;
  const bool x_plain = FASTP && a.Cin == CT * 16 && a.ldx >= CT * 16, z_plain = FASTP && a.Cout == NT * 16 && a.lddz >= NT * 16 && (!MASK || a.ldmask >= NT * 16);
  auto tile_interior = [&](int oy0, int ox0) {
    return FASTP && oy0 - padH >= 0 && oy0 - padH + IR <= a.H && ox0 - padW >= 0 && ox0 - padW + IC <= a.W && oy0 + HTH <= a.H && ox0 + HTW <= a.W;
  };
  auto load_tile = [&](int tile) {
    const int oy0 = (tile / a.tilesX) * HTH, ox0 = (tile % a.tilesX) * HTW;
    const bool fast = tile_interior(oy0, ox0);                  // uniform
    if (fast && x_plain) {
      const zt_bf16* xb = a.x + (unsigned)(((oy0 - padH) * a.W + ox0 - padW) * a.ldx);
#pragma unroll
      for (int i = 0; i < NXL; ++i) {
        int e = tid + i * NTHR;
        e = e < IR * IC * CT * 2 ? e : 0;                       // slots beyond the tile re-read slot 0 (never written)
        const int c8 = e % (CT * 2), p = e / (CT * 2);
        px[i] = *reinterpret_cast<const uint4*>(xb + (unsigned)(((p / IC) * a.W + p % IC) * a.ldx + c8 * 8));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NXL; ++i) {
        const int e = tid + i * NTHR;
        const int c8 = e % (CT * 2), p = e / (CT * 2);
        int gy = oy0 - padH + p / IC, gx = ox0 - padW + p % IC;
        gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
        gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
        const int c = c8 * 8 + 8 <= a.ldx ? c8 * 8 : 0;
        px[i] = *reinterpret_cast<const uint4*>(a.x + (unsigned)((gy * a.W + gx) * a.ldx + c));
      }
    }
    if (fast && z_plain) {
      const unsigned zo = (unsigned)((oy0 * a.W + ox0) * a.lddz), mo = MASK ? (unsigned)((oy0 * a.W + ox0) * a.ldmask) : 0u;
#pragma unroll
      for (int i = 0; i < NZL; ++i) {
        int e = tid + i * NTHR;
        e = e < HTH * HTW * NT * 2 ? e : 0;
        const int c8 = e % (NT * 2), p = e / (NT * 2);
        const int rel = (p / HTW) * a.W + p % HTW;
        pz[i] = *reinterpret_cast<const uint4*>(a.dz + zo + (unsigned)(rel * a.lddz + c8 * 8));
        if constexpr (MASK) pm[i] = *reinterpret_cast<const uint4*>(a.mask + mo + (unsigned)(rel * a.ldmask + c8 * 8));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NZL; ++i) {
        const int e = tid + i * NTHR;
        const int c8 = e % (NT * 2), p = e / (NT * 2);
        int gy = oy0 + p / HTW, gx = ox0 + p % HTW;
        gy = gy >= a.H ? a.H - 1 : gy;
        gx = gx >= a.W ? a.W - 1 : gx;
        const int c = c8 * 8 + 8 <= a.lddz ? c8 * 8 : 0;
        pz[i] = *reinterpret_cast<const uint4*>(a.dz + (unsigned)((gy * a.W + gx) * a.lddz + c));
        if constexpr (MASK) {
          const int cm = c8 * 8 + 8 <= a.ldmask ? c8 * 8 : 0;
          pm[i] = *reinterpret_cast<const uint4*>(a.mask + (unsigned)((gy * a.W + gx) * a.ldmask + cm));
        }
      }
    }
  };
  auto write_tile = [&](int tile) {
    const int oy0 = (tile / a.tilesX) * HTH, ox0 = (tile % a.tilesX) * HTW;
    const bool fast = tile_interior(oy0, ox0);                  // uniform
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int e = tid + i * NTHR;
      const int c8 = e % (CT * 2), p = e / (CT * 2);
      if (fast && x_plain) {
        if (e < IR * IC * CT * 2) *reinterpret_cast<uint4*>(xs + p * CIP + c8 * 8) = px[i];
      } else {
        const int gy = oy0 - padH + p / IC, gx = ox0 - padW + p % IC;
        const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        if (e < IR * IC * CT * 2) *reinterpret_cast<uint4*>(xs + p * CIP + c8 * 8) = chan_mask(px[i], a.Cin - c8 * 8, in);
      }
    }
#pragma unroll
    for (int i = 0; i < NZL; ++i) {
      const int e = tid + i * NTHR;
      const int c8 = e % (NT * 2), p = e / (NT * 2);
      uint4 g = pz[i];
      if constexpr (MASK) {
        g.x = relu_keep(g.x, pm[i].x);
        g.y = relu_keep(g.y, pm[i].y);
        g.z = relu_keep(g.z, pm[i].z);
        g.w = relu_keep(g.w, pm[i].w);
      }
      if (fast && z_plain) {
        if (e < HTH * HTW * NT * 2) *reinterpret_cast<uint4*>(zs + p * COP + c8 * 8) = g;
      } else {
        const int gy = oy0 + p / HTW, gx = ox0 + p % HTW;
        const bool in = gy < a.H && gx < a.W;
        if (e < HTH * HTW * NT * 2) *reinterpret_cast<uint4*>(zs + p * COP + c8 * 8) = chan_mask(g, a.Cout - c8 * 8, in);
      }
    }
  };

  if ((int)blockIdx.x < a.ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();
    write_tile(tile);
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) load_tile(tile + gridDim.x);
    if (bpart < NPART)
      for (int p = bpart; p < HTH * HTW; p += NPART) bsum += zt_bf2f(zs[p * COP + bco]);
    // per-wave (tap, ci-tile) pairs: branch-free (a wave without a pair in the last round recomputes the final pair into an
    // accumulator that is never written out), A fragments double-buffered and pinned ahead of the previous pair's MFMAs
    int aoff[PPW];
#pragma unroll
    for (int pi = 0; pi < PPW; ++pi) {
      int pr = wave + NW * pi;
      pr = pr < NPAIR ? pr : NPAIR - 1;
      const int tap = pr / CT, cit = pr - tap * CT;
      const int ky = tap / KW, kx = tap - ky * KW;
      aoff[pi] = (ky * IC + kx + g8 + trq) * CIP + cit * 16 + trp;
    }
    // Rows in blocks of four, fully unrolled inside a block: one flat software pipeline over the 4 * PPW (row, pair) steps.  The
    // A fragments (transposed x reads) run LA = 3 steps ahead of the MFMAs that consume them and the B fragments (dz) of a row
    // are requested one row earlier, across the row and block boundaries (indices clamped at the tile's end): with one step of
    // look-ahead inside a row and the B reads at the head of every row the 128+ clocks of LDS latency were exposed five-plus
    // times per row.
    constexpr int RB = 4, NS = RB * PPW, AD = 4, LA = 3;
    static_assert(HTH % RB == 0 && NS % AD == 0, "block geometry");
    zt_s16x4 alo[AD], ahi[AD];
    zt_s16x8 bv[2][NT];
    auto load_a = [&](auto bc, int row, auto pc) {
      constexpr int bi = decltype(bc)::value, pi = decltype(pc)::value;
      const zt_bf16* xr = xs + (row < HTH ? row : HTH - 1) * IC * CIP;
      alo[bi] = zt_lds_read_tr16(xr + aoff[pi]);
      ahi[bi] = zt_lds_read_tr16(xr + aoff[pi] + 4 * CIP);
    };
    auto load_b = [&](auto bc, int row) {
      constexpr int bi = decltype(bc)::value;
      const int rr = row < HTH ? row : HTH - 1;
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        zt_s16x4 lo = zt_lds_read_tr16(zs + (rr * HTW + g8 + trq) * COP + q * 16 + trp);
        zt_s16x4 hi = zt_lds_read_tr16(zs + (rr * HTW + g8 + 4 + trq) * COP + q * 16 + trp);
        bv[bi][q] = (zt_s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    load_b(ZtIdx<0>{}, 0);
    zt_static_for<0, LA>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      load_a(ZtIdx<st % AD>{}, st / PPW, ZtIdx<st % PPW>{});
    });
#pragma unroll 1
    for (int r0 = 0; r0 < HTH; r0 += RB) {
      zt_static_for<0, NS>([&](auto sc) {
        constexpr int st = decltype(sc)::value;
        constexpr int rl = st / PPW, pi = st % PPW, cur = st % AD;
        if constexpr (pi == 0) load_b(ZtIdx<(rl + 1) & 1>{}, r0 + rl + 1);          // next row's dz fragments (RB is even)
        {
          constexpr int nx = st + LA;                                                 // may run into the next block: row r0 + RB + ..
          load_a(ZtIdx<nx % AD>{}, r0 + nx / PPW, ZtIdx<nx % PPW>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        zt_s16x8 av = (zt_s16x8){alo[cur][0], alo[cur][1], alo[cur][2], alo[cur][3], ahi[cur][0], ahi[cur][1], ahi[cur][2], ahi[cur][3]};
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[pi][q] = zt_mfma_bf16(av, bv[rl & 1][q], acc[pi][q]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }
  float* out = a.slab + (size_t)blockIdx.x * (KH * KW * CT * 16 * NT * 16 + NT * 16);
  const int l4 = lane >> 4;
#pragma unroll
  for (int pi = 0; pi < PPW; ++pi) {
    const int pr = wave + NW * pi;
    if (pr < NPAIR) {
      const int tap = pr / CT, cit = pr - tap * CT;
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          out[((size_t)tap * CT * 16 + cit * 16 + l4 * 4 + j) * (NT * 16) + q * 16 + l15] = acc[pi][q][j];
    }
  }
  __syncthreads();
  float* fz = reinterpret_cast<float*>(zs);                 // HTH*HTW*COP bf16 >= NPART*NT16 floats
  if (bpart < NPART) fz[bpart * (NT * 16) + bco] = bsum;
  __syncthreads();
  if (tid < NT * 16) {
    float sum = 0.f;
    for (int k = 0; k < NPART; ++k) sum += fz[k * (NT * 16) + tid];
    out[KH * KW * CT * 16 * NT * 16 + tid] = sum;
  }
}

// ---- 64 -> 64 3x3 weight gradient (Enhancer conv.0: 3 launches per step), LDS-DMA form.
// Same MFMA decomposition as wgrad_mfma_bf16_kernel<3,3,4,4,8> (8 waves share the 36 (tap, ci-tile) pairs of an 8-row x 32-pixel
// tile; K = pixels through ds_read_b64_tr_b16), but
//  * both operand tiles go global -> LDS by DMA (`global_load_lds_dwordx4`): no staging registers, no ds_write pass, nothing of the
//    staging in any wave's instruction stream except the ~10 DMA issues per wave and tile;
//  * TWO tile buffers (2 x (10 x 34 + 8 x 32) pixels x 128 B = 149 KB): tile k+1 lands while tile k's MFMA loop runs, ONE barrier
//    per tile;
//  * pixel rows are exactly 128 B (a DMA destination is lane-linear, so rows cannot be padded) and XOR-swizzled at 32-byte (ci-tile)
//    granularity by s(col) = bit1(col) | bit3(col) << 1 -- applied to the SOURCE address of the DMA and to the read address.  A
//    transposing read's 32-lane half covers pixels {c..c+3, c+8..c+11} x 32 B: unswizzled these are 4-way bank conflicts on 128-B
//    rows (and 41 % of the LDS cycles on the former 144-byte-pitch image); with the swizzle every read is conflict-free (brute force
//    over all kx / ci-tile / row / half: DESIGN section 5);
//  * the bias gradient (column sums of dz) is an MFMA with an all-ones A fragment in the pair slot that wave 4 had idle (36 pairs
//    over 8 waves) instead of 32 two-byte LDS reads + adds per thread and tile;
//  * XCD-aware banded tile order as in conv_rs, so a tile's halo rows / columns are in its XCD's L2.
// Requires Cin == Cout == 64 and channel strides >= 64 (multiples of 8).
// CH = 48 (Denoise_1/2 conv2, six launches per step; round 3): the same kernel on 96-byte pixel rows.  A lane-linear DMA image cannot
// be padded and 6 chunks per pixel cannot be XOR-swizzled, so the transposing reads keep a 2-way conflict ({c..c+3} against
// {c+8..c+11}: every pitch from 96 to 208 bytes gives 2-way, brute force) -- the loop is VALU / MFMA bound, not LDS bound.  27 pairs
// over 8 waves: 4 slots per wave, the bias sums in wave 3's spare one.
constexpr int WG64_IR = 10, WG64_IC = 34;

template <int CH>
__device__ __forceinline__ int wg64_swz(int col) { return CH == 64 ? (((col >> 1) & 1) | (((col >> 3) & 1) << 1)) : 0; }

// CHX != CHZ (round 3): the thin-input first layers of Denoise_1/2 (Cin 3 / 12 in 8- / 16-channel pixels -> 48): one ci-tile, 9 pairs;
// the x image has 16- or 32-byte pixels (a transposing read of an 8-channel pixel takes its upper 8 "channels" from the next pixel:
// rows >= Cin of the product, which the slab reduction ignores, like the buffer's padding lanes).  Purely DMA / HBM bound.
template <int CHX, int CHZ>
__global__ void __launch_bounds__(512, 1) wgrad64_dma_bf16_kernel(WgradArgsH a) {
  static_assert((CHX == 64 && CHZ == 64) || (CHX == 48 && CHZ == 48) || ((CHX == 8 || CHX == 16) && CHZ == 48), "built shapes");
  constexpr int NW = 8, NTHR = 512, HTH = 8, IR = WG64_IR, IC = WG64_IC, CT = CHX >= 16 ? CHX / 16 : 1, NT = CHZ / 16;
  constexpr int CKX = CHX / 8, CKZ = CHZ / 8;
  constexpr int NPAIR = 9 * CT, PPW = (NPAIR + NW - 1) / NW;
  constexpr int WG64_XE = (IR * IC * CHX + 16 + 511) / 512 * 512, WG64_ZE = HTH * HTW * CHZ;       // x image + 32 B of slack, whole 1-KB pieces
  constexpr int NGX = (IR * IC * CKX + NTHR - 1) / NTHR, NGZ = (HTH * HTW * CKZ + NTHR - 1) / NTHR;    // DMA wave-instructions per wave and tile
  static_assert(HTH * HTW * CKZ % 64 == 0, "dz image = whole wave-instructions");
  __shared__ __attribute__((aligned(16))) zt_bf16 smem[2 * (WG64_XE + WG64_ZE)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g8 = (lane >> 4) * 8;
  const int trq = l15 >> 2, trp = (l15 & 3) * 4;

  // XCD-aware tile order (see conv_rs): workgroup b runs on XCD b % 8; an XCD's run of tiles walks bands of 4 tile rows column-major
  const int G = gridDim.x;
  const int pb = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int tilesY = a.ntiles / a.tilesX;
  auto tile_xy = [&](int idx, int& ty, int& tx) {
    const int band = idx / (4 * a.tilesX), r = idx - band * 4 * a.tilesX;
    const int rows = tilesY - band * 4 < 4 ? tilesY - band * 4 : 4;
    tx = r / rows;
    ty = band * 4 + r - tx * rows;
  };

  // DMA slot e = 64 (8 i + wave) + lane -> pixel e >> 3 of the tile image (row-major), PHYSICAL 16-byte chunk e & 7, which receives
  // the logical chunk (e & 7) ^ (s(col) << 1): the swizzle sits on the source address (a DMA destination is lane-linear).  The
  // slot's source offset relative to the tile origin is tile-invariant: computed once (interior tiles: one 64-bit add per DMA;
  // the inner loop is VALU-issue bound -- 2.9 VALU per MFMA in the first build of this kernel -- so per-tile index arithmetic counts)
  int xoff[NGX], zoff[NGZ];
#pragma unroll
  for (int i = 0; i < NGX; ++i) {
    const int e = (i * NW + wave) * 64 + lane;
    const int p = e / CKX, row = p / IC, col = p - row * IC;
    xoff[i] = (row * a.W + col) * a.ldx + (((e - p * CKX) ^ (wg64_swz<CHX>(col) << 1)) * 8);
  }
#pragma unroll
  for (int i = 0; i < NGZ; ++i) {
    const int e = (i * NW + wave) * 64 + lane;
    const int p = e / CKZ, row = p / HTW, col = p - row * HTW;
    zoff[i] = (row * a.W + col) * a.lddz + (((e - p * CKZ) ^ (wg64_swz<CHZ>(col) << 1)) * 8);
  }
  auto dma_tile = [&](int idx, int buf) {
    int ty, tx;
    tile_xy(idx, ty, tx);
    const int oy0 = ty * HTH, ox0 = tx * HTW;
    zt_bf16* xb = smem + buf * (WG64_XE + WG64_ZE);
    zt_bf16* zb = xb + WG64_XE;
    if (oy0 - 1 >= 0 && oy0 - 1 + IR <= a.H && ox0 - 1 >= 0 && ox0 - 1 + IC <= a.W) {      // uniform: interior tile (~95 % at 1080p)
      const zt_bf16* xo = a.x + (unsigned)(((oy0 - 1) * a.W + ox0 - 1) * a.ldx);
      const zt_bf16* zo = a.dz + (unsigned)((oy0 * a.W + ox0) * a.lddz);
#pragma unroll
      for (int i = 0; i < NGX; ++i)
        if ((i * NW + NW) * 64 <= IR * IC * CKX || (i * NW + wave) * 64 + lane < IR * IC * CKX) ZT_GLDS16_HIDDEN(xo + xoff[i], xb + (i * NW + wave) * 512);
#pragma unroll
      for (int i = 0; i < NGZ; ++i)
        if ((i * NW + NW) * 64 <= HTH * HTW * CKZ || (i * NW + wave) * 64 < HTH * HTW * CKZ) ZT_GLDS16_HIDDEN(zo + zoff[i], zb + (i * NW + wave) * 512);
      return;
    }
    int ln = lane;
    ZT_OPAQUE(ln);                                              // border tiles: slot geometry recomputed, out-of-image pixels read zeros
#pragma unroll
    for (int i = 0; i < NGX; ++i) {
      const int e = (i * NW + wave) * 64 + ln;
      const int p = e / CKX, row = p / IC, col = p - row * IC;
      const int cj = (e - p * CKX) ^ (wg64_swz<CHX>(col) << 1);
      const int gy = oy0 - 1 + row, gx = ox0 - 1 + col;
      const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
      const int gyc = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
      const void* src = in ? (const void*)(a.x + (unsigned)((gyc * a.W + gxc) * a.ldx + cj * 8)) : (const void*)&zt_zero_chunk;
      if ((i * NW + NW) * 64 <= IR * IC * CKX || e < IR * IC * CKX) ZT_GLDS16_HIDDEN(src, xb + (i * NW + wave) * 512);
    }
#pragma unroll
    for (int i = 0; i < NGZ; ++i) {
      const int e = (i * NW + wave) * 64 + ln;
      const int p = e / CKZ, row = p / HTW, col = p - row * HTW;
      const int cj = (e - p * CKZ) ^ (wg64_swz<CHZ>(col) << 1);
      const int gy = oy0 + row, gx = ox0 + col;
      const bool in = gy < a.H && gx < a.W;
      const int gyc = gy >= a.H ? a.H - 1 : gy, gxc = gx >= a.W ? a.W - 1 : gx;
      const void* src = in ? (const void*)(a.dz + (unsigned)((gyc * a.W + gxc) * a.lddz + cj * 8)) : (const void*)&zt_zero_chunk;
      if ((i * NW + NW) * 64 <= HTH * HTW * CKZ || (i * NW + wave) * 64 < HTH * HTW * CKZ) ZT_GLDS16_HIDDEN(src, zb + (i * NW + wave) * 512);
    }
  };

  zt_f32x4 acc[PPW][NT];
#pragma unroll
  for (int p = 0; p < PPW; ++p)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[p][q] = (zt_f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane fragment offsets (elements) inside a tile buffer: pair slot pi -> (tap, ci-tile); pairs 36..39 do not exist: wave 4's
  // spare slot carries the bias sums (A = ones), the spare slots of waves 5..7 recompute pair 35 into a discarded accumulator
  int alo[PPW], ahi[PPW];
#pragma unroll
  for (int pi = 0; pi < PPW; ++pi) {
    int pr = wave + NW * pi;
    pr = pr < NPAIR ? pr : NPAIR - 1;
    const int tap = pr / CT, cit = pr - tap * CT;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int c0 = kx + g8 + trq, c1 = c0 + 4;
    alo[pi] = (ky * IC + c0) * CHX + ((cit ^ wg64_swz<CHX>(c0)) * 16) + trp;
    ahi[pi] = (ky * IC + c1) * CHX + ((cit ^ wg64_swz<CHX>(c1)) * 16) + trp;
  }
  int blo[NT], bhi[NT];
  {
    const int c0 = g8 + trq, c1 = c0 + 4;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      blo[q] = WG64_XE + c0 * CHZ + ((q ^ wg64_swz<CHZ>(c0)) * 16) + trp;
      bhi[q] = WG64_XE + c1 * CHZ + ((q ^ wg64_swz<CHZ>(c1)) * 16) + trp;
    }
  }
  const bool ones_slot = wave == NPAIR % NW;                     // uniform: the first wave whose last pair slot is spare = bias column sums
  const zt_s16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};     // bf16 1.0

  const int n_my = pb < a.ntiles ? (a.ntiles - 1 - pb) / G + 1 : 0;
  if (n_my > 0) dma_tile(pb, 0);
  for (int k = 0; k < n_my; ++k) {
    ZT_WAIT_HIDDEN_DMA();             // this wave's pieces of tile k have landed ...
    __syncthreads();                  // ... and so have everyone else's; every wave has left tile k-1's loop (its buffer is free)
    // (a stagger -- waves 4..7 issuing their DMAs a quarter of the MFMA loop later, under their SIMD partner's MFMAs -- measured
    // no gain: 144.1 vs 143.9 us, profiles/r03_wgrad64_*; all eight issue at the head of the tile)
    if (k + 1 < n_my) dma_tile(pb + (k + 1) * G, (k + 1) & 1);
    // this tile's per-lane read addresses, once: everything below them is a compile-time row offset (ds_read immediate)
    const zt_bf16* tb = smem + (k & 1) * (WG64_XE + WG64_ZE);
    const zt_bf16 *pal[PPW], *pah[PPW], *pbl[NT], *pbh[NT];
#pragma unroll
    for (int pi = 0; pi < PPW; ++pi) {
      pal[pi] = tb + alo[pi];
      pah[pi] = tb + ahi[pi];
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      pbl[q] = tb + blo[q];
      pbh[q] = tb + bhi[q];
    }
    // ONE flat, fully unrolled software pipeline over the 8 rows x 5 pair slots: A fragments LA steps ahead of the MFMAs that
    // consume them, a row's B fragments one row ahead; steps past the tile's end re-read the last row (results unused)
    constexpr int NS = HTH * PPW, AD = 4, LA = 3;
    zt_s16x4 fal[AD], fah[AD];
    zt_s16x8 bv[2][NT];
    auto load_a = [&](auto bc, auto rc, auto pc) {
      constexpr int bi = decltype(bc)::value, pi = decltype(pc)::value;
      constexpr int row = decltype(rc)::value < HTH ? decltype(rc)::value : HTH - 1;
      fal[bi] = zt_lds_read_tr16(pal[pi] + row * IC * CHX);
      fah[bi] = zt_lds_read_tr16(pah[pi] + row * IC * CHX);
    };
    auto load_b = [&](auto bc, auto rc) {
      constexpr int bi = decltype(bc)::value;
      constexpr int row = decltype(rc)::value < HTH ? decltype(rc)::value : HTH - 1;
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const zt_s16x4 lo = zt_lds_read_tr16(pbl[q] + row * HTW * CHZ);
        const zt_s16x4 hi = zt_lds_read_tr16(pbh[q] + row * HTW * CHZ);
        bv[bi][q] = (zt_s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    load_b(ZtIdx<0>{}, ZtIdx<0>{});
    zt_static_for<0, LA>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      load_a(ZtIdx<st % AD>{}, ZtIdx<st / PPW>{}, ZtIdx<st % PPW>{});
    });
    zt_static_for<0, NS>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      constexpr int row = st / PPW, pi = st % PPW, cur = st % AD;
      if constexpr (pi == 0) load_b(ZtIdx<(row + 1) & 1>{}, ZtIdx<row + 1>{});
      {
        constexpr int nx = st + LA;
        load_a(ZtIdx<nx % AD>{}, ZtIdx<nx / PPW>{}, ZtIdx<nx % PPW>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      zt_s16x8 av = (zt_s16x8){fal[cur][0], fal[cur][1], fal[cur][2], fal[cur][3], fah[cur][0], fah[cur][1], fah[cur][2], fah[cur][3]};
      if constexpr (pi == PPW - 1) av = ones_slot ? ones : av;
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[pi][q] = zt_mfma_bf16(av, bv[row & 1][q], acc[pi][q]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // slab of this workgroup: [tap][ci CT*16][co CHZ] + [co CHZ] (same layout as wgrad_mfma_bf16_kernel)
  float* out = a.slab + (size_t)blockIdx.x * (9 * CT * 16 * CHZ + CHZ);
  const int l4 = lane >> 4;
#pragma unroll
  for (int pi = 0; pi < PPW; ++pi) {
    const int pr = wave + NW * pi;
    if (pr < NPAIR) {
      const int tap = pr / CT, cit = pr - tap * CT;
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[((size_t)tap * CT * 16 + cit * 16 + l4 * 4 + j) * CHZ + q * 16 + l15] = acc[pi][q][j];
    }
  }
  if (ones_slot && l4 == 0) {          // every row of the ones product holds the column sums: row 0 (lanes 0..15, register 0)
#pragma unroll
    for (int q = 0; q < NT; ++q) out[9 * CT * 16 * CHZ + q * 16 + l15] = acc[PPW - 1][q][0];
  }
}

// the 48 -> 48 3x3 layers take the DMA form too (ZT_WGRAD_DMA=0: the register-staged 4-wave kernel)
static bool wgrad48_dma(int K, int Cin, int Cout, int ldx, int lddz) {
  const int dma = getenv("ZT_WGRAD_DMA") ? atoi(getenv("ZT_WGRAD_DMA")) : 1;
  return dma && K == 3 && Cin == 48 && Cout == 48 && ldx >= 48 && lddz >= 48 && ldx % 8 == 0 && lddz % 8 == 0;
}

// ... and the thin-input 3x3 layers with 48 couts whose pixels are exactly 8 or 16 channels wide (Denoise_1/2 conv1)
static bool wgrad_thin48_dma(int K, int Cin, int Cout, int ldx, int lddz, const void* mask) {
  const int dma = getenv("ZT_WGRAD_DMA") ? atoi(getenv("ZT_WGRAD_DMA")) : 1;
  return dma && !mask && K == 3 && Cout == 48 && (ldx == 8 || ldx == 16) && Cin <= ldx && lddz >= 48 && lddz % 8 == 0;
}

template <int KH, int KW>
int launch_wgrad_h(const WgradArgsH& a, int CT, int NT, int nblk, hipStream_t stream) {
  dim3 grid(nblk);
#define ZT_WG(ct, nt, nw) hipLaunchKernelGGL((wgrad_mfma_bf16_kernel<KH, KW, ct, nt, nw>), grid, dim3(nw * 64), 0, stream, a); return 0
  if (CT == 1 && NT == 3) {
    if (wgrad_thin48_dma(KH, a.Cin, a.Cout, a.ldx, a.lddz, a.mask)) {
      if (a.ldx == 8) hipLaunchKernelGGL((wgrad64_dma_bf16_kernel<8, 48>), grid, dim3(512), 0, stream, a);
      else hipLaunchKernelGGL((wgrad64_dma_bf16_kernel<16, 48>), grid, dim3(512), 0, stream, a);
      return 0;
    }
    ZT_WG(1, 3, 4);
  }
  if (CT == 1 && NT == 4) {
    if (a.mask) {
      hipLaunchKernelGGL((wgrad_mfma_bf16_kernel<KH, KW, 1, 4, 4, true>), grid, dim3(256), 0, stream, a);
      return 0;
    }
    ZT_WG(1, 4, 4);
  }
  if (a.mask) return ZT_EINVAL;                                  // the folded ReLU mask exists for the thin-input 64-cout layer only
  if (CT == 3 && NT == 3) {
    if (wgrad48_dma(KH, a.Cin, a.Cout, a.ldx, a.lddz)) {          // 8-row tiles, one workgroup per CU (ntiles / nblk sized for it by the caller)
      hipLaunchKernelGGL((wgrad64_dma_bf16_kernel<48, 48>), grid, dim3(512), 0, stream, a);
      return 0;
    }
    ZT_WG(3, 3, 4);
  }
  if (CT == 3 && NT == 1) { ZT_WG(3, 1, 4); }
  if (CT == 4 && NT == 4) {
    static const int nw4 = getenv("ZT_WGRAD_NW4") ? atoi(getenv("ZT_WGRAD_NW4")) : 0;      // tuning hook: 4-wave / 4-row form
    if (nw4) { ZT_WG(4, 4, 4); }
    const int dma = getenv("ZT_WGRAD_DMA") ? atoi(getenv("ZT_WGRAD_DMA")) : 1;             // 0: the register-staged form (A/B: tests, tools/bench_wgrad.py)
    if (dma && KH == 3 && KW == 3 && a.Cin == 64 && a.Cout == 64 && a.ldx >= 64 && a.lddz >= 64 && a.ldx % 8 == 0 && a.lddz % 8 == 0 &&
        a.ntiles % a.tilesX == 0) {
      hipLaunchKernelGGL((wgrad64_dma_bf16_kernel<64, 64>), grid, dim3(512), 0, stream, a);
      return 0;
    }
    ZT_WG(4, 4, 8);
  }
  if (CT == 4 && NT == 1) { ZT_WG(4, 1, 4); }
#undef ZT_WG
  return ZT_EINVAL;
}

// torch fp32 [Cout][Cin][KH][KW] -> bf16 [tap][CoutP][ldk] (input channel fastest); transpose_flip: the data-gradient operator
__global__ void __launch_bounds__(256) repack_w_bf16_kernel(const float* __restrict__ src, zt_bf16* __restrict__ dst, int Cout,
                                                            int Cin, int KH, int KW, int CoutP, int ldk, int co_off,
                                                            int transpose_flip, int total) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int kx = idx % KW;
  int ky = (idx / KW) % KH;
  int ci = (idx / (KW * KH)) % Cin;
  int co = idx / (KW * KH * Cin);
  zt_bf16 v = zt_f2bf(src[idx]);
  if (!transpose_flip) dst[((size_t)(ky * KW + kx) * CoutP + co_off + co) * ldk + ci] = v;
  else dst[((size_t)((KH - 1 - ky) * KW + (KW - 1 - kx)) * CoutP + co_off + ci) * ldk + co] = v;
}

// torch [Cout][Cin][KH][KW] -> device [tap][Cin'][ldw] (forward) or the data-gradient form
// [tap'][Cout][ldw] with taps flipped and in/out channels exchanged.
__global__ void __launch_bounds__(256) repack_w_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout,
                                                       int Cin, int KH, int KW, int ldw, int co_off, int transpose_flip,
                                                       int total) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int kx = idx % KW;
  int ky = (idx / KW) % KH;
  int ci = (idx / (KW * KH)) % Cin;
  int co = idx / (KW * KH * Cin);
  float v = src[idx];
  if (!transpose_flip) dst[((size_t)(ky * KW + kx) * Cin + ci) * ldw + co_off + co] = v;
  else dst[((size_t)((KH - 1 - ky) * KW + (KW - 1 - kx)) * Cout + co) * ldw + co_off + ci] = v;
}

}  // namespace

extern "C" int zt_conv2d_nhwc_f32_ex(const float* x, const float* x2, int csplit, int ldx, int ldx2, int N, int H, int W,
                                     int Cin, const float* w, int ldw, const float* bias, float* y, int ldy, int out_planar,
                                     int Cout, int KH, int KW, int stride, int padH, int padW, int act, float alpha,
                                     const float* aux, int ldaux, int epi, float* y2, int ldy2, int esplit, hipStream_t stream) {
  ZT_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
  ZT_REQUIRE(epi >= 0 && epi <= 5 && (epi < 4 || !out_planar) && (epi != 4 || (y2 && esplit > 0 && esplit < Cout)));
  ZT_REQUIRE(ldx % 4 == 0 && ldw % 16 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0);
  ZT_REQUIRE(!x2 || (csplit % CK == 0 && ldx2 % 4 == 0 && ((uintptr_t)x2 & 15) == 0));
  ZT_REQUIRE(epi == 0 || aux);
  ConvArgs a;
  a.x = x; a.x2 = x2; a.w = w; a.bias = bias; a.aux = aux; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.ldx2 = ldx2; a.csplit = csplit;
  a.Ho = (H + 2 * padH - KH) / stride + 1;
  a.Wo = (W + 2 * padW - KW) / stride + 1;
  a.Cout = Cout; a.ldy = ldy; a.ldw = ldw; a.ldaux = ldaux;
  a.padH = padH; a.padW = padW; a.act = act; a.epi = epi; a.out_planar = out_planar; a.alpha = alpha;
  a.y2 = y2; a.ldy2 = ldy2; a.esplit = esplit;
  a.tilesX = zt_cdiv(a.Wo, TW);
  a.tilesY = zt_cdiv(a.Ho, TH);
  ZT_REQUIRE(a.Ho > 0 && a.Wo > 0);
  int c16 = (Cout + 15) / 16;
  int NT = c16 >= 4 ? ((c16 % 4 == 0) ? 4 : (c16 % 3 == 0 ? 3 : 4)) : c16;
  dim3 gb((unsigned)(a.tilesX * a.tilesY * N));
  int rc = ZT_EINVAL;
  if (KH == 3 && KW == 3 && stride == 1) rc = launch_conv<3, 3, 1>(a, NT, gb, stream);
  else if (KH == 3 && KW == 3 && stride == 2) rc = launch_conv<3, 3, 2>(a, NT, gb, stream);
  else if (KH == 1 && KW == 1 && stride == 1) rc = launch_conv<1, 1, 1>(a, NT, gb, stream);
  else if (KH == 1 && KW == 1 && stride == 2) rc = launch_conv<1, 1, 2>(a, NT, gb, stream);
  else if (KH == 1 && KW == 5 && stride == 1) rc = launch_conv<1, 5, 1>(a, NT, gb, stream);
  else if (KH == 5 && KW == 1 && stride == 1) rc = launch_conv<5, 1, 1>(a, NT, gb, stream);
  else if (KH == 7 && KW == 7 && stride == 1) rc = launch_conv<7, 7, 1>(a, NT, gb, stream);
  else if (KH == 7 && KW == 7 && stride == 2) rc = launch_conv<7, 7, 2>(a, NT, gb, stream);
  if (rc) return rc;
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_conv2d_nhwc_f32(const float* x, const float* x2, int csplit, int ldx, int ldx2, int N, int H, int W,
                                  int Cin, const float* w, int ldw, const float* bias, float* y, int ldy, int out_planar,
                                  int Cout, int KH, int KW, int stride, int padH, int padW, int act, float alpha,
                                  const float* aux, int ldaux, int epi, hipStream_t stream) {
  ZT_REQUIRE(epi >= 0 && epi <= 3);
  return zt_conv2d_nhwc_f32_ex(x, x2, csplit, ldx, ldx2, N, H, W, Cin, w, ldw, bias, y, ldy, out_planar, Cout, KH, KW, stride, padH,
                               padW, act, alpha, aux, ldaux, epi, nullptr, 0, 0, stream);
}

extern "C" int zt_conv2d_wgrad_nhwc_f32(const float* x, int ldx, const float* dz, int lddz, int H, int W, int Cin,
                                        int Cout, int KH, int KW, float* slab, size_t slab_bytes, float* grad_w,
                                        float* grad_b, int accumulate, hipStream_t stream) {
  ZT_REQUIRE(x && dz && slab && grad_w && ldx % 4 == 0 && lddz % 4 == 0);
  ZT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dz & 15) == 0);
  int CT = (Cin + 15) / 16, NT = (Cout + 15) / 16;
  WgradArgs a;
  a.x = x; a.dz = dz; a.slab = slab; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.Cout = Cout; a.lddz = lddz;
  a.tilesX = zt_cdiv(W, WTW);
  a.ntiles = a.tilesX * zt_cdiv(H, WTH);
  size_t per = ((size_t)KH * KW * CT * 16 * NT * 16 + NT * 16) * sizeof(float);
  int nblk = a.ntiles < 512 ? a.ntiles : 512;
  if ((size_t)nblk * per > slab_bytes) nblk = (int)(slab_bytes / per);
  ZT_REQUIRE(nblk >= 1);
  int rc = ZT_EINVAL;
  if (KH == 3 && KW == 3) rc = launch_wgrad<3, 3>(a, CT, NT, nblk, stream);
  else if (KH == 1 && KW == 1) rc = launch_wgrad<1, 1>(a, CT, NT, nblk, stream);
  if (rc) return rc;
  int total = KH * KW * CT * 16 * NT * 16 + NT * 16;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(zt_cdiv(total, 32)), dim3(256), 0, stream, (const float*)slab, nblk,
                     KH * KW, CT * 16, NT * 16, grad_w, Cout, Cin, accumulate, grad_b);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_repack_conv_weight_f32(const float* src, float* dst, int Cout, int Cin, int KH, int KW, int ldw,
                                         int co_off, int transpose_flip, hipStream_t stream) {
  ZT_REQUIRE(src && dst && ldw % 16 == 0);
  int total = Cout * Cin * KH * KW;
  hipLaunchKernelGGL(repack_w_kernel, dim3(zt_cdiv(total, 256)), dim3(256), 0, stream, src, dst, Cout, Cin, KH, KW, ldw,
                     co_off, transpose_flip, total);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

// variant: 0 = choose by problem size, 1 = force the persistent weight-stationary kernel, 2 = force the tiled kernel
struct BnBwdFuse {             // zt_conv3x3_dgrad_bn_sums_bf16: the previous block's pre-activation and BatchNorm constants
  const void* zprev;
  int ldz;
  const float *scale, *shift, *mean;
};

static int conv2d_bf16_impl(const void* x, const void* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin,
                            const void* w, int CoutP, int ldk, const float* bias, void* y, int ldy, int out_mode,
                            int Cout, int KH, int KW, int stride, int padH, int padW, int act, float alpha,
                            const void* aux, int ldaux, int epi, int variant, void* y2, int ldy2, int esplit, hipStream_t stream,
                            float* stats = nullptr, const BnBwdFuse* bnb = nullptr) {
  ZT_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && out_mode >= 0 && out_mode <= 2);
  ZT_REQUIRE(epi >= 0 && epi <= 6 && (epi < 4 || (out_mode == 0 && variant == 2)) && (epi != 4 || (y2 && esplit > 0 && esplit < Cout)));
  ZT_REQUIRE(ldx % 8 == 0 && ldk % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0);
  ZT_REQUIRE(!x2 || (csplit % HCK == 0 && ldx2 % 8 == 0 && ((uintptr_t)x2 & 15) == 0));
  ZT_REQUIRE(epi == 0 || aux);
  ConvArgsH a;
  a.x = (const zt_bf16*)x; a.x2 = (const zt_bf16*)x2; a.w = (const zt_bf16*)w; a.bias = bias; a.aux = (const zt_bf16*)aux; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.ldx2 = ldx2; a.csplit = csplit;
  a.Ho = (H + 2 * padH - KH) / stride + 1;
  a.Wo = (W + 2 * padW - KW) / stride + 1;
  a.Cout = Cout; a.CoutP = CoutP; a.ldk = ldk; a.ldy = ldy; a.ldaux = ldaux;
  a.padH = padH; a.padW = padW; a.act = act; a.epi = epi; a.out_mode = out_mode; a.alpha = alpha;
  a.y2 = (zt_bf16*)y2; a.ldy2 = ldy2; a.esplit = esplit; a.stats = stats;
  a.zprev = nullptr; a.ldz = 0; a.bn_scale = a.bn_shift = a.bn_mean = nullptr;
  if (bnb) { a.zprev = (const zt_bf16*)bnb->zprev; a.ldz = bnb->ldz; a.bn_scale = bnb->scale; a.bn_shift = bnb->shift; a.bn_mean = bnb->mean; }
  static const int scalar_epi = getenv("ZT_TILED_SCALAR_EPI") ? atoi(getenv("ZT_TILED_SCALAR_EPI")) : 0;   // A/B knob: per-element epilogue
  static const int aux_late = getenv("ZT_RS_AUX_LATE") ? atoi(getenv("ZT_RS_AUX_LATE")) : 0;              // A/B knob: conv_rs aux loads behind the halo DMAs
  a.dbg = (aux_late ? 128 : 0) | (variant >= 64 ? (variant - 64) : (variant >= 32 ? (variant - 32) : (variant >= 16 ? (variant - 16) : 0))) | (scalar_epi ? 32 : 0);   // tuning ablations, see tools/bench_conv.py / bench_small.py
  if (variant >= 64) variant = 2;
  if (variant >= 32) variant = 3;
  if (variant >= 16) variant = 1;
  ZT_REQUIRE(a.Ho > 0 && a.Wo > 0);
  // 32-bit element offsets in the staging code
  ZT_REQUIRE((long long)N * H * W * (ldx > ldx2 ? ldx : ldx2) < 0x7FFFFFFFll && (long long)KH * KW * CoutP * ldk < 0x7FFFFFFFll);
  a.tilesY = zt_cdiv(a.Ho, TH);
  int c16 = (Cout + 15) / 16;
  int NT = c16 >= 4 ? ((c16 % 4 == 0) ? 4 : (c16 % 3 == 0 ? 3 : 4)) : c16;
  // small feature maps (RAFT at 1/8 resolution): narrower tiles / fewer channels per workgroup so that >= ~2 workgroups per CU exist
  int MT = 2;
  long long wgs = (long long)zt_cdiv(a.Wo, 32) * a.tilesY * N * zt_cdiv(c16, NT);
  if (wgs < 512 || stride == 2) MT = 1;
  if (MT == 1 && NT == 4 && (long long)zt_cdiv(a.Wo, 16) * a.tilesY * N * zt_cdiv(c16, NT) < 512 && c16 % 2 == 0) NT = 2;
  // 32 couts per workgroup keep every tap's weights of a chunk resident next to the pixel tile (one staging + two barriers per chunk
  // instead of one per kernel row): -65 us over the RAFT encoders' 64-channel 180 x 320 layers (tools/bench_raft.py)
  if (NT == 4 && c16 % 2 == 0 && stride == 1) NT = 2;
  if (stride == 1) {                                            // tuning hooks for the tile shape (tools/bench_small.py, bench_raft.py)
    static const int fmt = getenv("ZT_TILED_MT") ? atoi(getenv("ZT_TILED_MT")) : 0, fnt = getenv("ZT_TILED_NT") ? atoi(getenv("ZT_TILED_NT")) : 0;
    if (fmt == 1 || fmt == 2) MT = fmt;
    if (fnt >= 1 && fnt <= 4 && c16 % fnt == 0) NT = fnt;
  }
  // full-resolution stride-1 layers of the enhancement nets: persistent weight-stationary kernel
  const bool ws_ok = N == 1 && stride == 1 && KH == KW && (KH == 1 || KH == 3) && padH == KH / 2 && padW == KW / 2 && Cin <= 64 && !x2;
  ZT_REQUIRE(variant != 1 || ws_ok);
  // thin-output 1x1 layers with planar fp32 output: streaming kernel
  if (variant == 0 && N == 1 && KH == 1 && KW == 1 && stride == 1 && padH == 0 && padW == 0 && !x2 && Cout <= 8 && out_mode == 1 &&
      epi == 0 && Cin % 8 == 0 && Cin <= 64 && ldk >= Cin && CoutP >= Cout) {
    const unsigned nb = (unsigned)zt_cdiv(a.Ho * a.Wo, 256);
    if (Cout <= 4) hipLaunchKernelGGL(conv1x1_thinout_bf16_kernel<4>, dim3(nb), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(conv1x1_thinout_bf16_kernel<8>, dim3(nb), dim3(256), 0, stream, a);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
  }
  // thin-input 1x1 layers: streaming kernel
  if (variant == 0 && N == 1 && KH == 1 && KW == 1 && stride == 1 && padH == 0 && padW == 0 && !x2 && Cin <= 8 && ldx == 8 && ldk == 8 &&
      out_mode == 0 && act <= 2 && Cout % 8 == 0 && CoutP >= Cout && ldy % 8 == 0 && ((uintptr_t)y & 15) == 0 &&
      (!aux || (ldaux % 8 == 0 && ((uintptr_t)aux & 15) == 0))) {
    const int npg = zt_cdiv(a.Ho * a.Wo, 4);
    const dim3 g1((unsigned)zt_cdivl((long long)npg * (Cout / 8), 256));
    if (Cout == 48 && !bias && act == 0 && alpha == 1.f && epi == 1 && (long long)npg * 6 < 0x7FFFFFFFll)
      hipLaunchKernelGGL(conv1x1_thin_bf16_kernel<true>, g1, dim3(256), 0, stream, a, npg);
    else
      hipLaunchKernelGGL(conv1x1_thin_bf16_kernel<false>, g1, dim3(256), 0, stream, a, npg);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
  }
  // register-stationary kernel: 3x3, bf16 nhwc output, 48 or 64 couts, input channels <= 16, 33..48 (Cout 48) or 49..64 (Cout 64)
  const bool rs_ok = ws_ok && KH == 3 && out_mode == 0 && act <= 2 && alpha == 1.f && ldy % 8 == 0 && ((uintptr_t)y & 15) == 0 &&
                     (!aux || (ldaux % 8 == 0 && ((uintptr_t)aux & 15) == 0)) && ldx >= 8 &&
                     ((Cout == 64 && (Cin <= 16 || Cin == 56 || Cin == 64)) || (Cout == 48 && (Cin <= 16 || Cin == 40 || Cin == 48)));
  ZT_REQUIRE(variant != 3 || rs_ok);
  static const int rs_auto = getenv("ZT_CONV_RS") ? atoi(getenv("ZT_CONV_RS")) : 1;
  if (stats && !(rs_ok && variant == 3)) return ZT_EINVAL;      // fused statistics exist in the register-stationary kernel only
  if (rs_ok && (variant == 3 || (variant == 0 && rs_auto && (long long)zt_cdiv(a.Wo, TW) * zt_cdiv(a.Ho, 8) >= 1024))) {
    int rcp = launch_conv_rs(a, stream);
    if (rcp) return rcp;
    ZT_LAUNCH_CHECK();
    return ZT_OK;
  }
  if (variant != 2 && ws_ok && (variant == 1 || (long long)zt_cdiv(a.Wo, TW) * zt_cdiv(a.Ho, 8) >= 1024)) {
    a.tilesX = zt_cdiv(a.Wo, TW);
    int CCH = Cin <= 32 ? 1 : 2;
    // 8-row tiles, all couts per workgroup: the best of the (rows, couts) configurations measured
    int rcw = (KH == 3) ? launch_conv_ws<3>(a, NT, CCH, 8, stream) : launch_conv_ws<1>(a, NT, CCH, 8, stream);
    if (rcw) return rcw;
    ZT_LAUNCH_CHECK();
    return ZT_OK;
  }
  a.tilesX = zt_cdiv(a.Wo, 16 * MT);
  unsigned gx = (unsigned)(a.tilesX * a.tilesY * N);
  int rc = ZT_EINVAL;
#define ZT_GEO(kh, kw, st)                                                        \
  if (KH == kh && KW == kw && stride == st)                                       \
    rc = (MT == 2) ? launch_conv_h<kh, kw, st, 2>(a, NT, gx, stream) : launch_conv_h<kh, kw, st, 1>(a, NT, gx, stream);
  ZT_GEO(3, 3, 1) ZT_GEO(3, 3, 2) ZT_GEO(1, 1, 1) ZT_GEO(1, 1, 2) ZT_GEO(1, 5, 1) ZT_GEO(5, 1, 1) ZT_GEO(7, 7, 1) ZT_GEO(7, 7, 2)
#undef ZT_GEO
  if (rc) return rc;
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

// motion-encoder pairs (see conv_mfma_bf16_pair_kernel / _pair2_kernel): square kernels KA, KB (pad K / 2, stride 1), bf16 nhwc in / out,
// same map and activation.  Falls back to two launches when the two problems do not take the kernel instantiations built here.
extern "C" int zt_conv2d_pair_nhwc_bf16(const void* xA, int ldxA, int CinA, const void* wA, int CoutPA, int ldkA, const float* biasA, void* yA,
                                        int ldyA, int CoutA, int KA, const void* xB, int ldxB, int CinB, const void* wB, int CoutPB, int ldkB,
                                        const float* biasB, void* yB, int ldyB, int CoutB, int KB, int N, int H, int W, int act, hipStream_t stream) {
  static const int off = getenv("ZT_RAFT_PAIR") ? !atoi(getenv("ZT_RAFT_PAIR")) : 0;     // A/B knob: ZT_RAFT_PAIR=0 -> two launches
  auto single = [&]() {
    int rc = conv2d_bf16_impl(xA, nullptr, 0, ldxA, 0, N, H, W, CinA, wA, CoutPA, ldkA, biasA, yA, ldyA, 0, CoutA, KA, KA, 1, KA / 2, KA / 2, act, 1.f,
                              nullptr, 0, 0, 0, nullptr, 0, 0, stream);
    if (rc) return rc;
    return conv2d_bf16_impl(xB, nullptr, 0, ldxB, 0, N, H, W, CinB, wB, CoutPB, ldkB, biasB, yB, ldyB, 0, CoutB, KB, KB, 1, KB / 2, KB / 2, act, 1.f,
                            nullptr, 0, 0, 0, nullptr, 0, 0, stream);
  };
  ZT_REQUIRE(xA && xB && wA && wB && yA && yB);
  const int tilesY = zt_cdiv(H, TH), tilesX = zt_cdiv(W, 16);
  const int gA = zt_cdiv(zt_cdiv(CoutA, 16), 2), gB = zt_cdiv(zt_cdiv(CoutB, 16), 2);
  // common conditions of the small-map instantiations (conv2d_bf16_impl / launch_conv_h): NT = 2 (32 couts per workgroup), MT = 1
  const bool small = !off && N == 1 && CoutA % 32 == 0 && CoutB % 32 == 0 && CoutA >= 64 && CoutB >= 64 &&
                     (long long)zt_cdiv(W, 32) * tilesY * zt_cdiv(CoutA / 16, 2) < 512 && (long long)zt_cdiv(W, 32) * tilesY * zt_cdiv(CoutB / 16, 2) < 512 &&
                     (long long)tilesX * tilesY * (gA + gB) <= 1024 && ldxA % 8 == 0 && ldxB % 8 == 0 && ldyA % 8 == 0 && ldyB % 8 == 0 && tilesY <= 65535;
  // (3x3, 3x3): both wide (64-channel chunks) and deep (two chunks in flight); (1x1, 7x7): 32-channel chunks, 1x1 deep, 7x7 per-row weights
  const bool p33 = small && KA == 3 && KB == 3 && CinA % 64 == 0 && CinB % 64 == 0 && CinA > 64 && CinB > 64;
  const bool p17 = small && KA == 1 && KB == 7 && CinA % 64 != 0 && CinA > 64 && CinB <= 8;
  if (!p33 && !p17) return single();
  ConvArgsH a[2];
  const void* xs[2] = {xA, xB};
  const void* ws[2] = {wA, wB};
  const float* bs[2] = {biasA, biasB};
  void* ys[2] = {yA, yB};
  const int ldx[2] = {ldxA, ldxB}, Cin[2] = {CinA, CinB}, CoutP[2] = {CoutPA, CoutPB}, ldk[2] = {ldkA, ldkB}, ldy[2] = {ldyA, ldyB}, Cout[2] = {CoutA, CoutB};
  const int Ks[2] = {KA, KB};
  for (int i = 0; i < 2; ++i) {
    ConvArgsH& c = a[i];
    c.x = (const zt_bf16*)xs[i]; c.x2 = nullptr; c.w = (const zt_bf16*)ws[i]; c.bias = bs[i]; c.aux = nullptr; c.y = ys[i];
    c.N = N; c.H = H; c.W = W; c.Cin = Cin[i]; c.ldx = ldx[i]; c.ldx2 = 0; c.csplit = 0;
    c.Ho = H; c.Wo = W; c.Cout = Cout[i]; c.CoutP = CoutP[i]; c.ldk = ldk[i]; c.ldy = ldy[i]; c.ldaux = 0;
    c.padH = Ks[i] / 2; c.padW = Ks[i] / 2; c.act = act; c.epi = 0; c.out_mode = 0; c.dbg = 0; c.alpha = 1.f;
    c.tilesX = tilesX; c.tilesY = tilesY; c.y2 = nullptr; c.ldy2 = 0; c.esplit = 0; c.stats = nullptr;
    c.zprev = nullptr; c.ldz = 0; c.bn_scale = c.bn_shift = c.bn_mean = nullptr;
    ZT_REQUIRE(((uintptr_t)c.x & 15) == 0 && ((uintptr_t)c.w & 15) == 0 && c.ldk % 8 == 0 && ((uintptr_t)c.y & 15) == 0);
  }
  const dim3 grid(tilesX, gA + gB, tilesY);
  if (p33) hipLaunchKernelGGL((conv_mfma_bf16_pair_kernel<3, 3, 1, 2, 1, true, 2, 2>), grid, dim3(256), 0, stream, a[0], a[1], gA);
  else hipLaunchKernelGGL((conv_mfma_bf16_pair2_kernel<1, 1, 2, true, 1, 2, 7, 7, 2, false, 1, 1>), grid, dim3(256), 0, stream, a[0], a[1], gA);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_conv2d_nhwc_bf16_variant(const void* x, const void* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin,
                                           const void* w, int CoutP, int ldk, const float* bias, void* y, int ldy, int out_mode,
                                           int Cout, int KH, int KW, int stride, int padH, int padW, int act, float alpha,
                                           const void* aux, int ldaux, int epi, int variant, hipStream_t stream) {
  ZT_REQUIRE(epi >= 0 && epi <= 3);
  return conv2d_bf16_impl(x, x2, csplit, ldx, ldx2, N, H, W, Cin, w, CoutP, ldk, bias, y, ldy, out_mode, Cout, KH, KW, stride, padH, padW,
                          act, alpha, aux, ldaux, epi, variant, nullptr, 0, 0, stream);
}

extern "C" int zt_conv2d_nhwc_bf16_ex(const void* x, const void* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin,
                                      const void* w, int CoutP, int ldk, const float* bias, void* y, int ldy, int out_mode,
                                      int Cout, int KH, int KW, int stride, int padH, int padW, int act, float alpha,
                                      const void* aux, int ldaux, int epi, void* y2, int ldy2, int esplit, hipStream_t stream) {
  return conv2d_bf16_impl(x, x2, csplit, ldx, ldx2, N, H, W, Cin, w, CoutP, ldk, bias, y, ldy, out_mode, Cout, KH, KW, stride, padH, padW,
                          act, alpha, aux, ldaux, epi, epi >= 4 ? 2 : 0, y2, ldy2, esplit, stream);
}

extern "C" int zt_chan_stats_nhwc(const void* x, int dt, int ldx, int N, int HW, int C, int nblk, float* partial, hipStream_t stream);

extern "C" int zt_conv3x3_bn_stats_bf16(const void* x, int ldx, int H, int W, int Cin, const void* w, int CoutP, int ldk, const float* bias,
                                        void* y, int ldy, int Cout, float* stats, int stats_blocks, hipStream_t stream) {
  ZT_REQUIRE(x && w && y && stats && stats_blocks == 512 && Cout % 8 == 0);
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(float) * (size_t)stats_blocks * 2 * Cout, stream);
  if (e != hipSuccess) return (int)e;
  const char* mt = getenv("ZT_STATS_FUSE_MIN_TILES");          // tests force the fused kernel onto small images
  const long long min_tiles = mt ? atoll(mt) : 1024;
  const bool fused = Cout == 64 && (Cin == 56 || Cin == 64) && ldx >= 8 && ldy % 8 == 0 && ((uintptr_t)y & 15) == 0 &&
                     (long long)zt_cdiv(W, TW) * zt_cdiv(H, 8) >= min_tiles && !(getenv("ZT_CONV_RS") && atoi(getenv("ZT_CONV_RS")) == 0);
  if (fused)
    return conv2d_bf16_impl(x, nullptr, 0, ldx, 0, 1, H, W, Cin, w, CoutP, ldk, bias, y, ldy, 0, Cout, 3, 3, 1, 1, 1, 0, 1.f, nullptr, 0, 0, 3,
                            nullptr, 0, 0, stream, stats);
  int rc = conv2d_bf16_impl(x, nullptr, 0, ldx, 0, 1, H, W, Cin, w, CoutP, ldk, bias, y, ldy, 0, Cout, 3, 3, 1, 1, 1, 0, 1.f, nullptr, 0, 0, 0,
                            nullptr, 0, 0, stream);
  if (rc) return rc;
  const int HW = H * W;
  int nblk = HW / 64 < 1 ? 1 : (HW / 64 > stats_blocks ? stats_blocks : HW / 64);
  return zt_chan_stats_nhwc(y, 1, ldy, 1, HW, Cout, nblk, stats, stream);
}

extern "C" int zt_conv3x3_dgrad_bn_sums_bf16(const void* dz, int lddz, int H, int W, const void* wT, int CoutP, int ldk, void* df, int lddf,
                                             const void* res, int ldres, const void* zprev, int ldz, const float* bn_scale,
                                             const float* bn_shift, const float* bn_mean, float* stats, int stats_blocks, hipStream_t stream) {
  ZT_REQUIRE(dz && wT && df && res && zprev && bn_scale && bn_shift && bn_mean && stats && stats_blocks == 512);
  ZT_REQUIRE(ldz % 8 == 0 && ((uintptr_t)zprev & 15) == 0 && (long long)zt_cdiv(W, TW) * zt_cdiv(H, 4) >= 1);
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(float) * (size_t)stats_blocks * 2 * 64, stream);      // rows beyond the launch's workgroups stay zero
  if (e != hipSuccess) return (int)e;
  BnBwdFuse b = {zprev, ldz, bn_scale, bn_shift, bn_mean};
  return conv2d_bf16_impl(dz, nullptr, 0, lddz, 0, 1, H, W, 64, wT, CoutP, ldk, nullptr, df, lddf, 0, 64, 3, 3, 1, 1, 1, 0, 1.f, res, ldres, 3, 3, nullptr, 0,
                          0, stream, stats, &b);
}

extern "C" int zt_conv2d_nhwc_bf16(const void* x, const void* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin,
                                   const void* w, int CoutP, int ldk, const float* bias, void* y, int ldy, int out_mode, int Cout,
                                   int KH, int KW, int stride, int padH, int padW, int act, float alpha, const void* aux,
                                   int ldaux, int epi, hipStream_t stream) {
  return zt_conv2d_nhwc_bf16_variant(x, x2, csplit, ldx, ldx2, N, H, W, Cin, w, CoutP, ldk, bias, y, ldy, out_mode, Cout, KH, KW,
                                     stride, padH, padW, act, alpha, aux, ldaux, epi, 0, stream);
}

// partial pass: every workgroup writes one slab ([tap][ci16][co16] weights + [co16] bias column sums); -> number of slabs
static int wgrad_partial_bf16(const void* x, int ldx, const void* dz, int lddz, int H, int W, int Cin, int Cout, int KH, int KW,
                              float* slab, size_t slab_bytes, const void* relu_mask, int ldmask, int* nslab_out, hipStream_t stream) {
  ZT_REQUIRE(x && dz && slab && ldx % 8 == 0 && lddz % 8 == 0);
  ZT_REQUIRE(!relu_mask || (ldmask % 8 == 0 && ((uintptr_t)relu_mask & 15) == 0 && KH == 3 && Cin <= 16 && Cout == 64));
  ZT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dz & 15) == 0);
  int CT = (Cin + 15) / 16, NT = (Cout + 15) / 16;
  WgradArgsH a;
  a.x = (const zt_bf16*)x; a.dz = (const zt_bf16*)dz; a.slab = slab; a.H = H; a.W = W; a.Cin = Cin; a.ldx = ldx; a.Cout = Cout;
  a.lddz = lddz; a.mask = (const zt_bf16*)relu_mask; a.ldmask = ldmask;
  a.tilesX = zt_cdiv(W, HTW);
  const bool nw8 = (CT == 4 && NT == 4 && !(getenv("ZT_WGRAD_NW4") && atoi(getenv("ZT_WGRAD_NW4")))) ||
                   (CT == 3 && NT == 3 && KW == KH && wgrad48_dma(KH, Cin, Cout, ldx, lddz)) ||
                   (CT == 1 && NT == 3 && KW == KH && wgrad_thin48_dma(KH, Cin, Cout, ldx, lddz, relu_mask));
  a.ntiles = a.tilesX * zt_cdiv(H, nw8 ? 8 : 4);      // tile rows = waves of the variant (launch_wgrad_h)
  size_t per = ((size_t)KH * KW * CT * 16 * NT * 16 + NT * 16) * sizeof(float);
  // the 8-wave variant runs one workgroup per CU: 256 slabs keep every CU busy and halve its slab traffic (measured 280 -> 266 us);
  // the 4-wave variants co-reside two or three per CU
  int want = nw8 ? 256 : 512;
  if (const char* e = getenv("ZT_WGRAD_BLOCKS")) want = atoi(e) > 0 ? atoi(e) : want;      // tuning hook
  int nblk = a.ntiles < want ? a.ntiles : want;
  if ((size_t)nblk * per > slab_bytes) nblk = (int)(slab_bytes / per);
  ZT_REQUIRE(nblk >= 1);
  int rc = ZT_EINVAL;
  if (KH == 3 && KW == 3) rc = launch_wgrad_h<3, 3>(a, CT, NT, nblk, stream);
  else if (KH == 1 && KW == 1) rc = launch_wgrad_h<1, 1>(a, CT, NT, nblk, stream);
  if (rc) return rc;
  *nslab_out = nblk;
  return ZT_OK;
}

extern "C" int zt_conv2d_wgrad_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, int H, int W, int Cin, int Cout,
                                         int KH, int KW, float* slab, size_t slab_bytes, float* grad_w, float* grad_b,
                                         int accumulate, const void* relu_mask, int ldmask, hipStream_t stream) {
  ZT_REQUIRE(grad_w);
  int nblk = 0;
  int rc = wgrad_partial_bf16(x, ldx, dz, lddz, H, W, Cin, Cout, KH, KW, slab, slab_bytes, relu_mask, ldmask, &nblk, stream);
  if (rc) return rc;
  int CT = (Cin + 15) / 16, NT = (Cout + 15) / 16;
  int total = KH * KW * CT * 16 * NT * 16 + NT * 16;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(zt_cdiv(total, 32)), dim3(256), 0, stream, (const float*)slab, nblk, KH * KW,
                     CT * 16, NT * 16, grad_w, Cout, Cin, accumulate, grad_b);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_thin1x1_bwd_bf16(const void* dr, int Cdr, const void* wT, const void* a2, int lda, void* dz, int lddz, int HW, float* slab,
                                   size_t slab_bytes, int* nslab_out, hipStream_t stream) {
  ZT_REQUIRE(dr && wT && a2 && dz && slab && nslab_out && HW > 0 && (Cdr == 3 || Cdr == 6) && lda % 8 == 0 && lddz % 8 == 0 && lda >= 48 && lddz >= 48);
  ZT_REQUIRE(((uintptr_t)dr & 15) == 0 && ((uintptr_t)wT & 15) == 0 && ((uintptr_t)a2 & 15) == 0 && ((uintptr_t)dz & 15) == 0);
  ThinBwdArgs a;
  a.dr = (const zt_bf16*)dr; a.wT = (const zt_bf16*)wT; a.a2 = (const zt_bf16*)a2; a.dz = (zt_bf16*)dz; a.slab = slab;
  a.HW = HW; a.npg = zt_cdiv(HW, 4); a.lda = lda; a.lddz = lddz; a.Cdr = Cdr;
  const size_t per = (48 * 16 + 16) * sizeof(float);
  int nblk = zt_cdiv(a.npg, 42);
  if (nblk > 512) nblk = 512;
  if ((size_t)nblk * per > slab_bytes) nblk = (int)(slab_bytes / per);
  ZT_REQUIRE(nblk >= 1);
  hipLaunchKernelGGL(thin1x1_bwd_bf16_kernel, dim3(nblk), dim3(256), 0, stream, a);
  ZT_LAUNCH_CHECK();
  *nslab_out = nblk;
  return ZT_OK;
}

extern "C" int zt_conv2d_wgrad_partial_bf16(const void* x, int ldx, const void* dz, int lddz, int H, int W, int Cin, int Cout, int KH,
                                            int KW, float* slab, size_t slab_bytes, const void* relu_mask, int ldmask, int* nslab_out,
                                            hipStream_t stream) {
  ZT_REQUIRE(nslab_out);
  int rc = wgrad_partial_bf16(x, ldx, dz, lddz, H, W, Cin, Cout, KH, KW, slab, slab_bytes, relu_mask, ldmask, nslab_out, stream);
  if (rc) return rc;
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_wgrad_reduce_multi_f32(int nseg, const void* const* slab, const int* nslab, const int* Cin, const int* Cout,
                                         const int* K, void* const* grad_w, void* const* grad_b, int accumulate, hipStream_t stream) {
  ZT_REQUIRE(nseg >= 1 && nseg <= ZT_MAXSEG && slab && nslab && Cin && Cout && K && grad_w && grad_b);
  ReduceTable t;
  int maxtotal = 0;
  for (int i = 0; i < ZT_MAXSEG; ++i) {
    const int j = i < nseg ? i : 0;
    ZT_REQUIRE(slab[j] && grad_w[j] && nslab[j] >= 1);
    t.slab[i] = (const float*)slab[j]; t.gw[i] = (float*)grad_w[j]; t.gb[i] = (float*)grad_b[j];
    t.nslab[i] = nslab[j]; t.ntap[i] = K[j] * K[j]; t.CT16[i] = (Cin[j] + 15) / 16 * 16; t.NT16[i] = (Cout[j] + 15) / 16 * 16;
    t.Cout[i] = Cout[j]; t.Cin[i] = Cin[j];
    const int total = t.ntap[i] * t.CT16[i] * t.NT16[i] + t.NT16[i];
    if (i < nseg && total > maxtotal) maxtotal = total;
  }
  t.accumulate = accumulate;
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(zt_cdiv(maxtotal, 32), nseg), dim3(256), 0, stream, t);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_repack_conv_weights_bf16_multi(int n, const void* const* src, void* const* dst, const int* Cout, const int* Cin,
                                                 const int* K, const int* CoutP, const int* ldk, const int* transpose_flip,
                                                 hipStream_t stream) {
  ZT_REQUIRE(n >= 1 && n <= ZT_MAXREP && src && dst && Cout && Cin && K && CoutP && ldk && transpose_flip);
  RepackTable t;
  int maxtotal = 0;
  for (int i = 0; i < ZT_MAXREP; ++i) {
    const int j = i < n ? i : 0;
    ZT_REQUIRE(src[j] && dst[j] && CoutP[j] % 16 == 0 && ldk[j] % 8 == 0);
    t.src[i] = (const float*)src[j]; t.dst[i] = (zt_bf16*)dst[j]; t.Cout[i] = Cout[j]; t.Cin[i] = Cin[j]; t.K[i] = K[j];
    t.CoutP[i] = CoutP[j]; t.ldk[i] = ldk[j]; t.tflip[i] = transpose_flip[j];
    const int total = Cout[j] * Cin[j] * K[j] * K[j];
    if (i < n && total > maxtotal) maxtotal = total;
  }
  int gx = zt_cdiv(maxtotal, 256);
  hipLaunchKernelGGL(repack_w_bf16_multi_kernel, dim3(gx > 64 ? 64 : gx, n), dim3(256), 0, stream, t);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_repack_conv_weight_bf16(const float* src, void* dst, int Cout, int Cin, int KH, int KW, int CoutP, int ldk,
                                          int co_off, int transpose_flip, hipStream_t stream) {
  ZT_REQUIRE(src && dst && CoutP % 16 == 0 && ldk % 8 == 0);
  int total = Cout * Cin * KH * KW;
  hipLaunchKernelGGL(repack_w_bf16_kernel, dim3(zt_cdiv(total, 256)), dim3(256), 0, stream, src, (zt_bf16*)dst, Cout, Cin, KH, KW,
                     CoutP, ldk, co_off, transpose_flip, total);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
