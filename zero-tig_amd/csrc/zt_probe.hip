// Hardware self-test probes (used by tests/ to pin the emulator's model of gfx950-specific instructions to the real chip).
#include "zt_common.h"

namespace {

// out[lane*4 + q] = element q returned to `lane` by ds_read_b64_tr_b16 on a [rows][64] image of 16-bit codes, where lane
// 4q+p of group g addresses row 4g+q, columns 4p..4p+3 (+ col0).
__global__ void __launch_bounds__(64) probe_tr16_kernel(const unsigned short* __restrict__ img, unsigned short* __restrict__ out,
                                                        int col0) {
  __shared__ zt_bf16 lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = img[i];
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  zt_s16x4 v = zt_lds_read_tr16(lds + (4 * g + q) * 64 + col0 + 4 * p);
  for (int k = 0; k < 4; ++k) out[l * 4 + k] = (unsigned short)v[k];
}

// one bf16 MFMA 16x16x32 with explicit A [16][32] and B [32][16] (bf16 bits) -> D [16][16] fp32
__global__ void __launch_bounds__(64) probe_mfma_bf16_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                             float* __restrict__ D) {
  int l = threadIdx.x;
  zt_s16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = (short)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  zt_f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = zt_mfma_bf16(a, b, c);
  for (int j = 0; j < 4; ++j) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
}

}  // namespace

extern "C" int zt_probe_tr16(const unsigned short* img, unsigned short* out, int col0, hipStream_t stream) {
  ZT_REQUIRE(img && out && col0 >= 0 && col0 % 4 == 0 && col0 <= 48);
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, stream, img, out, col0);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_probe_mfma_bf16(const unsigned short* A, const unsigned short* B, float* D, hipStream_t stream) {
  ZT_REQUIRE(A && B && D);
  hipLaunchKernelGGL(probe_mfma_bf16_kernel, dim3(1), dim3(64), 0, stream, A, B, D);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
