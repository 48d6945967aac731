// Input side of train.py / predict.py on the device: the loader workers only DECODE (PNG/JPEG -> interleaved
// 8-bit RGB); the frame crosses PCIe as 6 MB of bytes instead of 25 MB of floats, and what the reference does on the host after
// decoding (dataloader/multi_read_data.py:127-132) runs here:
//   im.resize((1920, 1080))        PIL's default filter for RGB images = BICUBIC, the 8-bit path of Pillow's Resample.c:
//                                  horizontal pass into an 8-bit image, then vertical pass; per output sample
//                                  clip8((2^21 + sum_k src[xmin + k] * coef[k]) >> 22) with integer coefficients
//                                  round(w_k * 2^22) -- the coefficient / bounds tables are computed on the host in double
//                                  precision exactly as Pillow's precompute_coeffs does (zero-tig_amd/ingest.py) and passed in;
//   transforms.ToTensor()          uint8 HWC -> float CHW, x / 255 (IEEE division: a 256-entry table of float(k) / 255.f
//                                  computed by the host's correctly rounded division, so the result is bit-identical to torch).
// Integer / byte work, HBM-bound and tiny (6 MB per pass): coalesced 4-byte accesses along the contiguous axis, tables in LDS.
#include "zt_common.h"

namespace {

__device__ __forceinline__ unsigned clip8(int v) {                     // Resample.c clip8(): table lookup of v >> PRECISION_BITS
  v >>= 22;
  return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [H][Wi][3] u8 -> dst [H][Wo][3] u8.  Block = 64 output pixels x 4 rows; the 64 pixels' coefficient rows
// and bounds sit in LDS; a thread produces one pixel (3 channels) of one row.
template <int KS>
__global__ void __launch_bounds__(256) resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int H,
                                                         int Wi, int Wo, const int* __restrict__ coef, const int* __restrict__ bounds,
                                                         int ksize) {
  __shared__ int sk[64 * KS];
  __shared__ int sb[64 * 2];
  const int x0 = blockIdx.x * 64;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int i = tid; i < 64 * KS; i += 256) {
    const int px = x0 + i / KS, k = i % KS;
    sk[i] = (px < Wo && k < ksize) ? coef[(size_t)px * ksize + k] : 0;
  }
  if (tid < 128) sb[tid] = (x0 + tid / 2 < Wo) ? bounds[(size_t)(x0 + tid / 2) * 2 + (tid & 1)] : 0;
  __syncthreads();
  const int x = x0 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= Wo || y >= H) return;
  const int xmin = sb[threadIdx.x * 2], n = sb[threadIdx.x * 2 + 1];
  const unsigned char* row = src + ((size_t)y * Wi + xmin) * 3;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    if (k < n) {
      const int c = sk[threadIdx.x * KS + k];
      a0 += (int)row[3 * k] * c;
      a1 += (int)row[3 * k + 1] * c;
      a2 += (int)row[3 * k + 2] * c;
    }
  }
  unsigned char* d = dst + ((size_t)y * Wo + x) * 3;
  d[0] = (unsigned char)clip8(a0);
  d[1] = (unsigned char)clip8(a1);
  d[2] = (unsigned char)clip8(a2);
}

// vertical pass: src [Hi][RB] u8 -> dst [Ho][RB] u8 (RB = row bytes = 3 * W, a multiple of 4): a thread produces 4 consecutive
// bytes of one output row from one 4-byte load per tap row; the tap loop is wave-uniform (one output row per block row).
__global__ void __launch_bounds__(256) resample_v_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int Hi, int Ho,
                                                         int RW, const int* __restrict__ coef, const int* __restrict__ bounds,
                                                         int ksize) {
  const int y = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= RW) return;
  const int ymin = bounds[2 * y], n = bounds[2 * y + 1];
  const int* k = coef + (size_t)y * ksize;
  int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21, a3 = 1 << 21;
  for (int t = 0; t < n; ++t) {
    const unsigned v = src[(size_t)(ymin + t) * RW + i];
    const int c = k[t];
    a0 += (int)(v & 255u) * c;
    a1 += (int)((v >> 8) & 255u) * c;
    a2 += (int)((v >> 16) & 255u) * c;
    a3 += (int)(v >> 24) * c;
  }
  dst[(size_t)y * RW + i] = clip8(a0) | (clip8(a1) << 8) | (clip8(a2) << 16) | (clip8(a3) << 24);
}

// ToTensor: interleaved u8 [H][W][3] -> planar fp32 [3][H*W] through the 256-entry division table; a thread converts 4 pixels:
// three 4-byte loads, three 16-byte stores (one per plane).  HW must be a multiple of 4.
__global__ void __launch_bounds__(256) u8hwc_to_planar_kernel(const unsigned* __restrict__ src, float* __restrict__ dst, long long HW,
                                                              const float* __restrict__ lut) {
  __shared__ float sl[256];
  sl[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g * 4 >= HW) return;
  const unsigned w0 = src[g * 3], w1 = src[g * 3 + 1], w2 = src[g * 3 + 2];
  // bytes: p0.rgb p1.r | p1.gb p2.rg | p2.b p3.rgb
  const float4 r = make_float4(sl[w0 & 255u], sl[w0 >> 24], sl[(w1 >> 16) & 255u], sl[(w2 >> 8) & 255u]);
  const float4 gch = make_float4(sl[(w0 >> 8) & 255u], sl[w1 & 255u], sl[w1 >> 24], sl[(w2 >> 16) & 255u]);
  const float4 b = make_float4(sl[(w0 >> 16) & 255u], sl[(w1 >> 8) & 255u], sl[w2 & 255u], sl[w2 >> 24]);
  *reinterpret_cast<float4*>(dst + g * 4) = r;
  *reinterpret_cast<float4*>(dst + HW + g * 4) = gch;
  *reinterpret_cast<float4*>(dst + 2 * HW + g * 4) = b;
}

}  // namespace

extern "C" int zt_resample_u8_hwc(const unsigned char* src, unsigned char* dst, int Hi, int Wi, int Ho, int Wo, int axis, const int* coef,
                                  const int* bounds, int ksize, hipStream_t stream) {
  ZT_REQUIRE(src && dst && coef && bounds && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && ksize > 0 && ksize <= 33);
  if (axis == 1) {                      // horizontal: rows kept (Ho == Hi)
    ZT_REQUIRE(Ho == Hi);
    const dim3 grid(zt_cdiv(Wo, 64), zt_cdiv(Hi, 4)), block(64, 4);
#define ZT_RH(KS) hipLaunchKernelGGL(resample_h_kernel<KS>, grid, block, 0, stream, src, dst, Hi, Wi, Wo, coef, bounds, ksize)
    if (ksize <= 5) ZT_RH(5);
    else if (ksize <= 9) ZT_RH(9);
    else if (ksize <= 17) ZT_RH(17);
    else ZT_RH(33);
#undef ZT_RH
  } else {                              // vertical: row length kept (Wo == Wi), rows are 3 * W bytes = RW dwords
    ZT_REQUIRE(axis == 0 && Wo == Wi && (3 * Wi) % 4 == 0 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 3) == 0);
    const int RW = 3 * Wi / 4;
    hipLaunchKernelGGL(resample_v_kernel, dim3(zt_cdiv(RW, 256), Ho), dim3(256), 0, stream, (const unsigned*)src, (unsigned*)dst, Hi, Ho,
                       RW, coef, bounds, ksize);
  }
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_u8hwc_to_planar_f32(const unsigned char* src, float* dst, int H, int W, const float* lut256, hipStream_t stream) {
  const long long HW = (long long)H * W;
  ZT_REQUIRE(src && dst && lut256 && H > 0 && W > 0 && HW % 4 == 0 && ((uintptr_t)src & 3) == 0 && ((uintptr_t)dst & 15) == 0);
  hipLaunchKernelGGL(u8hwc_to_planar_kernel, dim3((unsigned)zt_cdivl(HW / 4, 256)), dim3(256), 0, stream, (const unsigned*)src, dst, HW,
                     lut256);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
