// Output side of predict.py / evals.py on the device: the enhanced frames leave HBM as interleaved 8-bit RGB
// (6 MB instead of 25 MB per 1080p frame over PCIe, no host-side float pass), and the PSNR of evals.py:83-85 is an exact
// integer reduction.
//   predict.py:57-61  save_images: clip(x * 255, 0, 255).astype(uint8)      -> truncation        (mode 0)
//   evals.py:83-84    np.round(x * 255).astype(np.uint8)                    -> round-half-even   (mode 1)
//   evals.py:85       cv2.PSNR(img, gt) = 10 log10(255^2 / mean((img - gt)^2)) over all elements
#include "zt_common.h"

namespace {

__device__ __forceinline__ int quant_u8(float v, int mode) {
  const float s = v * 255.f;
  if (mode == 0) return (int)fminf(fmaxf(s, 0.f), 255.f);       // np.clip then C truncation
  // np.round(..).astype(np.uint8): round half to even, then the wrap-around of a uint8 cast (values are clamped upstream to
  // [1e-4, 1], so the wrap never triggers; it is reproduced anyway)
  return (int)rintf(s) & 255;
}

// planar fp32 [3][H][W] -> interleaved uint8 [H][W][3]; thread = 4 pixels = 12 output bytes = three 4-byte stores
__global__ void __launch_bounds__(256) quantize_hwc_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, long long HW,
                                                           int mode) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;       // group of 4 pixels
  const long long p = g * 4;
  if (p >= HW) return;
  unsigned char b[12];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long pp = p + j < HW ? p + j : HW - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) b[j * 3 + c] = (unsigned char)quant_u8(src[(size_t)c * HW + pp], mode);
  }
  if (p + 4 <= HW) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      dst[g * 3 + k] = (unsigned)b[4 * k] | ((unsigned)b[4 * k + 1] << 8) | ((unsigned)b[4 * k + 2] << 16) | ((unsigned)b[4 * k + 3] << 24);
  } else {
    unsigned char* d = reinterpret_cast<unsigned char*>(dst) + p * 3;
    for (long long k = 0; k < (HW - p) * 3; ++k) d[k] = b[k];
  }
}

// sum over n elements of (round(a * 255) - round(b * 255))^2 as unsigned 64-bit partials (exact, order-independent)
__global__ void __launch_bounds__(256) sqdiff_u8_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                        unsigned long long* __restrict__ partial) {
  __shared__ unsigned long long red[256];
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int d = quant_u8(a[i], 1) - quant_u8(b[i], 1);
    s += (unsigned long long)(d * d);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void sqdiff_final_kernel(const unsigned long long* __restrict__ partial, int nblk, unsigned long long* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long s = 0;
  for (int i = 0; i < nblk; ++i) s += partial[i];
  out[0] = s;
}

}  // namespace

extern "C" int zt_quantize_u8_hwc(const float* src, unsigned char* dst, int H, int W, int mode, hipStream_t stream) {
  ZT_REQUIRE(src && dst && H > 0 && W > 0 && (mode == 0 || mode == 1) && ((uintptr_t)dst & 3) == 0);
  const long long HW = (long long)H * W;
  hipLaunchKernelGGL(quantize_hwc_kernel, dim3((unsigned)zt_cdivl(zt_cdivl(HW, 4), 256)), dim3(256), 0, stream, src, (unsigned*)dst, HW, mode);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_sqdiff_u8_f32(const float* a, const float* b, long long n, unsigned long long* partial, int nblk,
                                unsigned long long* out, hipStream_t stream) {
  ZT_REQUIRE(a && b && partial && out && n > 0 && nblk > 0 && nblk <= 4096);
  hipLaunchKernelGGL(sqdiff_u8_kernel, dim3(nblk), dim3(256), 0, stream, a, b, n, partial);
  hipLaunchKernelGGL(sqdiff_final_kernel, dim3(1), dim3(64), 0, stream, (const unsigned long long*)partial, nblk, out);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
