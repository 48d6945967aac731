// RAFT encoder stem (reference model/RAFT/extractor.py:120, 168-170): conv 7x7, stride 2, padding 3, 3 -> 64 channels, on the
// normalised frames -- bf16 throughput mode.
//
// The generic tiled kernel pads the 3 input channels to a 32-channel MFMA K step for each of the 49 taps and re-stages the
// weights per kernel row (49 K steps, 7 weight stagings with barriers: 60-105 us per call).  Here the frames are NHWC with 8
// channels per pixel (3 valid, 5 ZERO -- written so by zt_raft_pack_input / zt_raft_pack_pair), so the 7 pixels a kernel row
// touches are 56 CONTIGUOUS bf16 values: K per kernel row = 7 px x 8 ch = 56 -> 64 = two K = 32 steps, 14 steps in all instead
// of 49, all weights of the workgroup's 32 couts resident in LDS (one staging, one barrier), A fragments read straight from the
// staged input rows at 16-byte pixel granularity (conflict-free: slot = 2 l15 + l4).
// Weights: zt_repack_stem_weight_bf16 -> [ky][cout 64][k 64], k = kx * 8 + c (zero for c >= 3 and for the padding kx = 7).
#include "zt_common.h"

namespace {

constexpr int ST_TOH = 4, ST_TOW = 32;                 // output tile: one row per wave, two 16-pixel MFMA tiles per wave
constexpr int ST_IR = 2 * ST_TOH + 5, ST_IC = 72;      // input rows / pixels staged (2 * 32 + 5 = 69 used, + the padding tap)
constexpr int ST_WP = 80;                              // weight row pitch (elements): conflict-free ds_read_b128 (see zt_conv.hip)
constexpr int ST_NT = 2;                               // 32 couts per workgroup

__global__ void __launch_bounds__(256) stem7x7s2_bf16_kernel(const zt_bf16* __restrict__ x, int H, int W, const zt_bf16* __restrict__ w,
                                                             const float* __restrict__ bias, zt_bf16* __restrict__ y, int ldy, int Ho,
                                                             int Wo, int tilesY, int relu) {
  __shared__ __attribute__((aligned(16))) zt_bf16 xs[ST_IR * ST_IC * 8];
  __shared__ __attribute__((aligned(16))) zt_bf16 ws[7 * ST_NT * 16 * ST_WP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int tx = blockIdx.x, co0 = blockIdx.y * (ST_NT * 16);
  const int n = blockIdx.z / tilesY, ty = blockIdx.z - n * tilesY;
  const int oy0 = ty * ST_TOH, ox0 = tx * ST_TOW;
  const int gy0 = 2 * oy0 - 3, gx0 = 2 * ox0 - 3;
  const zt_bf16* xn = x + (size_t)n * H * W * 8;
  // stage the input rows (16 bytes per pixel; zeros outside the image and beyond the 69 pixels the tile reaches)
  for (int e = tid; e < ST_IR * ST_IC; e += 256) {
    const int r = e / ST_IC, c = e - r * ST_IC;
    const int gy = gy0 + r, gx = gx0 + c;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (gy >= 0 && gy < H && gx >= 0 && gx < W && c < 2 * ST_TOW + 6) v = *reinterpret_cast<const uint4*>(xn + ((size_t)gy * W + gx) * 8);
    *reinterpret_cast<uint4*>(xs + e * 8) = v;
  }
  // stage this workgroup's weights: [ky][32 couts][64 k]
  for (int e = tid; e < 7 * ST_NT * 16 * 8; e += 256) {
    const int q = e & 7, r = e >> 3;
    const int co = r % (ST_NT * 16), ky = r / (ST_NT * 16);
    *reinterpret_cast<uint4*>(ws + (ky * ST_NT * 16 + co) * ST_WP + q * 8) =
        *reinterpret_cast<const uint4*>(w + ((size_t)ky * 64 + co0 + co) * 64 + q * 8);
  }
  float bias_q[ST_NT];
#pragma unroll
  for (int q = 0; q < ST_NT; ++q) bias_q[q] = bias ? bias[co0 + q * 16 + l15] : 0.f;
  __syncthreads();
  zt_f32x4 acc[2][ST_NT];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < ST_NT; ++q) acc[m][q] = (zt_f32x4){0.f, 0.f, 0.f, 0.f};
  // output pixel (wave, m * 16 + l15), kernel row ky, K step k2: input row 2 wave + ky, pixels 2 (m 16 + l15) + 4 k2 + l4
  const zt_bf16* xa = xs + ((2 * wave) * ST_IC + 2 * l15 + l4) * 8;
  const zt_bf16* wb = ws + l15 * ST_WP + l4 * 8;
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      zt_s16x8 fa[2], fb[ST_NT];
#pragma unroll
      for (int m = 0; m < 2; ++m) fa[m] = *reinterpret_cast<const zt_s16x8*>(xa + (ky * ST_IC + m * 32 + k2 * 4) * 8);
#pragma unroll
      for (int q = 0; q < ST_NT; ++q) fb[q] = *reinterpret_cast<const zt_s16x8*>(wb + (ky * ST_NT * 16 + q * 16) * ST_WP + k2 * 32);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < ST_NT; ++q) acc[m][q] = zt_mfma_bf16(fa[m], fb[q], acc[m][q]);
    }
  const int oy = oy0 + wave;
  if (oy >= Ho) return;
  zt_bf16* yn = y + (size_t)n * Ho * Wo * ldy;
#pragma unroll
  for (int q = 0; q < ST_NT; ++q) {
    const int co = co0 + q * 16 + l15;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = ox0 + m * 16 + l4 * 4 + j;
        if (ox < Wo) {
          const float v = acc[m][q][j] + bias_q[q];
          yn[((size_t)oy * Wo + ox) * ldy + co] = zt_f2bf(relu ? fmaxf(v, 0.f) : v);
        }
      }
  }
}

// torch [64][3][7][7] fp32 -> [ky][cout][kx * 8 + c] bf16 (64 k per row, zero padded)
__global__ void __launch_bounds__(256) repack_stem_kernel(const float* __restrict__ src, zt_bf16* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 7 * 64 * 64) return;
  const int k = i & 63, co = (i >> 6) & 63, ky = i >> 12;
  const int kx = k >> 3, c = k & 7;
  const float v = (kx < 7 && c < 3) ? src[((co * 3 + c) * 7 + ky) * 7 + kx] : 0.f;
  dst[i] = zt_f2bf(v);
}

}  // namespace

extern "C" int zt_repack_stem_weight_bf16(const float* src, void* dst, hipStream_t stream) {
  ZT_REQUIRE(src && dst);
  hipLaunchKernelGGL(repack_stem_kernel, dim3(zt_cdiv(7 * 64 * 64, 256)), dim3(256), 0, stream, src, (zt_bf16*)dst);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_raft_stem_conv_bf16(const void* x, int N, int H, int W, const void* w, const float* bias, void* y, int ldy,
                                      int relu, hipStream_t stream) {
  ZT_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && ldy >= 64 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0);
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tilesX = zt_cdiv(Wo, ST_TOW), tilesY = zt_cdiv(Ho, ST_TOH);
  ZT_REQUIRE((long long)tilesY * N <= 65535);
  hipLaunchKernelGGL(stem7x7s2_bf16_kernel, dim3(tilesX, 64 / (ST_NT * 16), tilesY * N), dim3(256), 0, stream, (const zt_bf16*)x, H, W,
                     (const zt_bf16*)w, bias, (zt_bf16*)y, ldy, Ho, Wo, tilesY, relu);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
