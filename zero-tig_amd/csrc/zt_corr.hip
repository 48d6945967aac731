// All-pairs correlation volume AND its 3-level average pyramid in ONE pass (reference model/RAFT/corr.py:13-27, 52-60:
// `corr = fmap1^T fmap2 / sqrt(C)`, then three `avg_pool2d(2, 2)` over the image-2 axes), bf16 throughput mode.
//
// Before: a generic 1x1-convolution launch for the GEMM (34 us at 1080p, the 51.8 MB volume written once but both feature maps
// re-fetched by every XCD: 1.7x the algorithmic traffic) + three pooling launches that re-read what was just written (3 x 7 us).
// Here a workgroup owns 64 source pixels x an (8 rows x 32 columns) patch of image 2 -- 8 x 32 because level 3 pools 8 x 8 blocks, so
// every pyramid cell of every level is complete inside one workgroup -- and
//  * K = 256 channels run as 4 chunks of 64 through a double-buffered LDS tile filled by LDS-DMA (hidden from the compiler, see
//    ZT_GLDS16_HIDDEN); 128-byte pixel rows, 16-byte chunks XOR-swizzled by (pixel & 7) on the DMA's source address and on the read:
//    every ds_read_b128 fragment read is conflict-free (brute force over all lane groups: DESIGN section 5);
//  * wave v computes all 64 source pixels x the 8 x 8 sub-patch of columns [8v, 8v + 8): 4 x 4 MFMA tiles, and in that layout the
//    2 x 2 / 4 x 4 / 8 x 8 pooling partners of an element are lanes l^1, l^8 / l^2 and the neighbouring N-tile / l^4 and the
//    N-tile pair: the whole pyramid comes out of the accumulators by cross-lane adds, nothing is re-read from HBM;
//  * all four levels leave through an LDS transpose as full runs per (source pixel, image-2 row): 128 B for level 0.
// HBM-bound by construction: 68.4 MB written at 1080p (829 MB + 271 MB at 4K) against 6.6 GFLOP; the feature maps (1.8 MB each)
// stay in L2.  Pooling sums are (a + b) + (c + d), each level from the rounded level below, like avg_pool2d applied three times.
#include "zt_common.h"

namespace {

__device__ const uint4 zt_corr_zero_chunk = {0u, 0u, 0u, 0u};        // DMA source of out-of-range pixels

struct CorrArgs {
  const zt_bf16* f1;       // [npx][ld1] source-image features (C = 256)
  const zt_bf16* f2;       // [npx][ld2] target-image features
  float* c0;               // [npx][ld0]        level 0: row = source pixel, column = y2 * w + x2
  float* c1;               // [npx][h1 * w1]    h1 = h / 2 ...
  float* c2;
  float* c3;
  int h, w, npx, ld1, ld2, ld0;
  float alpha;
};

constexpr int CR_M = 64, CR_R = 8, CR_C = 32, CR_N = CR_R * CR_C, CR_KC = 64;
constexpr int CR_AE = CR_M * CR_KC, CR_BE = CR_N * CR_KC;                  // bf16 elements per operand chunk
constexpr int CR_STAGE_FLOATS = 32 * (256 + 64 + 16 + 4);                 // epilogue staging of 32 source pixels, all levels

__global__ void __launch_bounds__(256, 2) corr_pyramid_bf16_kernel(CorrArgs a) {
  // operands: 2 x (A 8 KB + B 32 KB) = 80 KB; the epilogue staging (43.5 KB per half) re-uses it
  __shared__ __attribute__((aligned(16))) zt_bf16 smem[2 * (CR_AE + CR_BE)];
  static_assert(CR_STAGE_FLOATS * 4 <= 2 * (CR_AE + CR_BE) * 2, "staging fits the operand buffers");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int x0 = blockIdx.x * CR_C, y0 = blockIdx.y * CR_R, m0 = blockIdx.z * CR_M;

  // ---- DMA slot geometry (chunk-invariant): slot e = 64 (4 i + wave) + lane -> pixel e >> 3, physical 16-byte chunk e & 7, which
  // receives logical chunk (e & 7) ^ (pixel & 7)
  constexpr int NGA = CR_M * 8 / 256, NGB = CR_N * 8 / 256;                  // 2 + 8 wave-instructions per wave and chunk
  const zt_bf16* asrc[NGA];
  const zt_bf16* bsrc[NGB];
#pragma unroll
  for (int i = 0; i < NGA; ++i) {
    const int e = (i * 4 + wave) * 64 + lane, p = e >> 3, cj = (e & 7) ^ (p & 7);
    const int m = m0 + p;
    asrc[i] = m < a.npx ? a.f1 + (size_t)m * a.ld1 + cj * 8 : nullptr;
  }
#pragma unroll
  for (int i = 0; i < NGB; ++i) {
    const int e = (i * 4 + wave) * 64 + lane, p = e >> 3, cj = (e & 7) ^ (p & 7);
    const int y = y0 + (p >> 5), x = x0 + (p & 31);
    bsrc[i] = (y < a.h && x < a.w) ? a.f2 + (size_t)(y * a.w + x) * a.ld2 + cj * 8 : nullptr;
  }
  auto dma_chunk = [&](int kc, int buf) {
    zt_bf16* ab = smem + buf * (CR_AE + CR_BE);
    zt_bf16* bb = ab + CR_AE;
#pragma unroll
    for (int i = 0; i < NGA; ++i) {
      const void* src = asrc[i] ? (const void*)(asrc[i] + kc * CR_KC) : (const void*)&zt_corr_zero_chunk;
      ZT_GLDS16_HIDDEN(src, ab + (i * 4 + wave) * 512);
    }
#pragma unroll
    for (int i = 0; i < NGB; ++i) {
      const void* src = bsrc[i] ? (const void*)(bsrc[i] + kc * CR_KC) : (const void*)&zt_corr_zero_chunk;
      ZT_GLDS16_HIDDEN(src, bb + (i * 4 + wave) * 512);
    }
  };

  // ---- fragment read offsets (elements, inside a chunk buffer), K-step ks in {0, 1}: chunk = 4 ks + l4
  // A: M-tile mt -> pixel 16 mt + l15;  B: N-tile t -> patch row 2 t + (l15 >> 3), column 8 wave + (l15 & 7)
  int aoff[4][2], boff[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int pa = t * 16 + l15;
    const int pbx = (2 * t + (l15 >> 3)) * CR_C + 8 * wave + (l15 & 7);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      aoff[t][ks] = pa * 64 + (((ks * 4 + l4) ^ (pa & 7)) * 8);
      boff[t][ks] = CR_AE + pbx * 64 + (((ks * 4 + l4) ^ (pbx & 7)) * 8);
    }
  }

  zt_f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[mt][t] = (zt_f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int NKC = 256 / CR_KC;
  dma_chunk(0, 0);
#pragma unroll 1
  for (int kc = 0; kc < NKC; ++kc) {
    ZT_WAIT_HIDDEN_DMA();
    __syncthreads();                                              // chunk kc landed; everyone is done reading the other buffer
    if (kc + 1 < NKC) dma_chunk(kc + 1, (kc + 1) & 1);
    const zt_bf16* cb = smem + (kc & 1) * (CR_AE + CR_BE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      zt_s16x8 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = *reinterpret_cast<const zt_s16x8*>(cb + aoff[t][ks]);
        fb[t] = *reinterpret_cast<const zt_s16x8*>(cb + boff[t][ks]);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = zt_mfma_bf16(fa[mt], fb[t], acc[mt][t]);
    }
  }

  // ---- epilogue: two halves of 32 source pixels; all four levels staged in LDS, then written as contiguous runs
  const int h1 = a.h / 2, w1 = a.w / 2, h2 = h1 / 2, w2 = w1 / 2, h3 = h2 / 2, w3 = w2 / 2;
  float* stg = reinterpret_cast<float*>(smem);
  float* s0 = stg;                         // [32][8][32]
  float* s1 = stg + 32 * 256;              // [32][4][16]
  float* s2 = s1 + 32 * 64;                // [32][2][8]
  float* s3 = s2 + 32 * 16;                // [32][4]
  zt_static_for<0, 2>([&](auto halfc) {
    constexpr int half = decltype(halfc)::value;
    __syncthreads();                                              // operand reads / the previous half's stores are done
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ml = mh * 16 + l4 * 4 + j;                      // source pixel inside the half
        float l1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float v = a.alpha * acc[half * 2 + mh][t][j];
          const int r = 2 * t + (l15 >> 3), c = 8 * wave + (l15 & 7);
          s0[(ml * 8 + r) * 32 + c] = v;
          float s = v + __shfl_xor(v, 1);                         // (c, c ^ 1)
          s = s + __shfl_xor(s, 8);                               // rows (2t, 2t + 1)
          l1[t] = 0.25f * s;
          if ((l15 & 9) == 0) s1[(ml * 4 + t) * 16 + 4 * wave + ((l15 & 7) >> 1)] = l1[t];
        }
        float l2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float u0 = l1[2 * k] + __shfl_xor(l1[2 * k], 2), u1 = l1[2 * k + 1] + __shfl_xor(l1[2 * k + 1], 2);
          l2[k] = 0.25f * (u0 + u1);
          if ((l15 & 11) == 0) s2[(ml * 2 + k) * 8 + 2 * wave + ((l15 & 7) >> 2)] = l2[k];
        }
        const float w0 = l2[0] + __shfl_xor(l2[0], 4), w1s = l2[1] + __shfl_xor(l2[1], 4);
        if (l15 == 0) s3[ml * 4 + wave] = 0.25f * (w0 + w1s);
      }
    }
    __syncthreads();
    const int mbase = m0 + half * 32;
    // level 0: 32 px x 8 rows x 8 float4
    for (int e = tid; e < 32 * 8 * 8; e += 256) {
      const int c4 = e & 7, r = (e >> 3) & 7, ml = e >> 6;
      const int m = mbase + ml, y = y0 + r, x = x0 + c4 * 4;
      if (m < a.npx && y < a.h && x < a.w) {
        const float4 v = *reinterpret_cast<const float4*>(s0 + (ml * 8 + r) * 32 + c4 * 4);
        float* dst = a.c0 + (size_t)m * a.ld0 + (size_t)y * a.w + x;
        if (x + 4 <= a.w && ((((size_t)m * a.ld0 + (size_t)y * a.w + x) & 3) == 0)) *reinterpret_cast<float4*>(dst) = v;
        else {
          const float tv[4] = {v.x, v.y, v.z, v.w};
          for (int k = 0; k < 4 && x + k < a.w; ++k) dst[k] = tv[k];
        }
      }
    }
    for (int e = tid; e < 32 * 4 * 16; e += 256) {                // level 1
      const int c = e & 15, r = (e >> 4) & 3, ml = e >> 6;
      const int m = mbase + ml, y = y0 / 2 + r, x = x0 / 2 + c;
      if (m < a.npx && y < h1 && x < w1) a.c1[(size_t)m * (h1 * w1) + y * w1 + x] = s1[(ml * 4 + r) * 16 + c];
    }
    for (int e = tid; e < 32 * 2 * 8; e += 256) {                 // level 2
      const int c = e & 7, r = (e >> 3) & 1, ml = e >> 4;
      const int m = mbase + ml, y = y0 / 4 + r, x = x0 / 4 + c;
      if (m < a.npx && y < h2 && x < w2) a.c2[(size_t)m * (h2 * w2) + y * w2 + x] = s2[(ml * 2 + r) * 8 + c];
    }
    if (tid < 32 * 4) {                                           // level 3
      const int c = tid & 3, ml = tid >> 2;
      const int m = mbase + ml, y = y0 / 8, x = x0 / 8 + c;
      if (m < a.npx && y < h3 && x < w3) a.c3[(size_t)m * (h3 * w3) + y * w3 + x] = s3[ml * 4 + c];
    }
  });
}

}  // namespace

extern "C" int zt_corr_volume_pyramid_bf16(const void* f1, int ld1, const void* f2, int ld2, int h, int w, float alpha, float* c0, int ld0,
                                           float* c1, float* c2, float* c3, hipStream_t stream) {
  const int npx = h * w;
  ZT_REQUIRE(f1 && f2 && c0 && c1 && c2 && c3 && h >= 16 && w >= 16 && ld1 >= 256 && ld2 >= 256 && ld1 % 8 == 0 && ld2 % 8 == 0 && ld0 >= npx);
  ZT_REQUIRE(((uintptr_t)f1 & 15) == 0 && ((uintptr_t)f2 & 15) == 0 && ((uintptr_t)c0 & 15) == 0);
  const int gz = zt_cdiv(npx, CR_M);
  ZT_REQUIRE(gz <= 65535 && zt_cdiv(h, CR_R) <= 65535);
  CorrArgs a;
  a.f1 = (const zt_bf16*)f1; a.f2 = (const zt_bf16*)f2; a.c0 = c0; a.c1 = c1; a.c2 = c2; a.c3 = c3;
  a.h = h; a.w = w; a.npx = npx; a.ld1 = ld1; a.ld2 = ld2; a.ld0 = ld0; a.alpha = alpha;
  hipLaunchKernelGGL(corr_pyramid_bf16_kernel, dim3(zt_cdiv(w, CR_C), zt_cdiv(h, CR_R), gz), dim3(256), 0, stream, a);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
