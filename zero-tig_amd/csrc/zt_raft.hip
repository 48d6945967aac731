// RAFT-specific kernels of Network.update_cache (reference model/model.py:221-259): input preparation (bilinear
// down-scale, uint8 truncation + histogram equalisation, replicate padding), correlation pyramid + fused 4-level
// 9x9 lookup (model/RAFT/corr.py:12-50), SepConvGRU point-wise stages (update.py:33-60), flow bookkeeping and the
// convex 8x up-sampling (raft.py:64-75).  Convolutions / the correlation GEMM live in zt_conv.hip.
#include "zt_common.h"

namespace {

struct Lin1 {
  int i0, i1;
  float w0, w1;
};

__device__ __forceinline__ Lin1 lin_index(int dst, int in_size, int out_size, float scale) {
  Lin1 r;
  if (in_size == out_size) {
    r.i0 = r.i1 = dst;
    r.w0 = 1.f;
    r.w1 = 0.f;
    return r;
  }
  float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
  src = src < 0.f ? 0.f : src;
  int i0 = (int)floorf(src);
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  float lam = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  r.i0 = i0;
  r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  r.w1 = lam;
  r.w0 = 1.f - lam;
  return r;
}

// F.interpolate(bilinear, align_corners=False) * mul  (model.py:226-227, 231) -- ATen CPU arithmetic
__global__ void __launch_bounds__(256) resize_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                              int H, int W, int h, int w, float sc_h, float sc_w, float mul) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= w || y >= h) return;
  Lin1 ly = lin_index(y, H, h, sc_h), lx = lin_index(x, W, w, sc_w);
  for (int c = 0; c < C; ++c) {
    const float* p = src + (size_t)c * H * W;
    float a = p[(size_t)ly.i0 * W + lx.i0], b = p[(size_t)ly.i0 * W + lx.i1];
    float cc = p[(size_t)ly.i1 * W + lx.i0], d = p[(size_t)ly.i1 * W + lx.i1];
    float v;
    if (H == h && W == w) v = a;
    else {
      float r0 = fmaf(a, lx.w0, b * lx.w1), r1 = fmaf(cc, lx.w0, d * lx.w1);
      v = fmaf(r0, ly.w0, r1 * ly.w1);
    }
    dst[(size_t)c * h * w + (size_t)y * w + x] = v * mul;
  }
}

// (x).to(uint8) (truncation) + 256-bin histogram per channel  (model.py:234 feeding torchvision equalize)
__global__ void __launch_bounds__(256) quantize_hist_kernel(const float* __restrict__ src, unsigned char* __restrict__ q,
                                                            int* __restrict__ hist, int hw) {
  __shared__ int lh[256];
  const int c = blockIdx.y;
  lh[threadIdx.x] = 0;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
    float v = src[(size_t)c * hw + i];
    int b = (int)v;                       // C-style truncation == torch .to(uint8) for values in [0, 256)
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    q[(size_t)c * hw + i] = (unsigned char)b;
    atomicAdd(&lh[b], 1);
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[c * 256 + threadIdx.x], lh[threadIdx.x]);
}

// torchvision 0.18.1 _scale_channel: step = floor(sum(nonzero_hist[:-1]) / 255); lut = floor((cumsum + step//2) / step),
// shifted right by one, clamped; identity when step == 0.  One 256-lane workgroup per channel: LDS scan of the 256 bins
// (Hillis-Steele, 8 rounds), one integer division per lane.  Counts are pixel counts (< 2^31): 32-bit arithmetic is exact.
__global__ void __launch_bounds__(256) equalize_lut_kernel(const int* __restrict__ hist, int* __restrict__ lut, int C) {
  __shared__ int cum[2][256];
  __shared__ int lastk[256];
  const int c = blockIdx.x, k = threadIdx.x;
  const int hk = hist[c * 256 + k];
  cum[0][k] = hk;
  lastk[k] = hk != 0 ? k : -1;
  __syncthreads();
  int cur = 0;
  for (int off = 1; off < 256; off <<= 1) {
    const int v = cum[cur][k] + (k >= off ? cum[cur][k - off] : 0);
    const int m = max(lastk[k], k >= off ? lastk[k - off] : -1);
    __syncthreads();
    cum[cur ^ 1][k] = v;
    lastk[k] = m;
    __syncthreads();
    cur ^= 1;
  }
  const int total = cum[cur][255], last = lastk[255];                 // inclusive scans: sum / running max of non-empty bins
  const int step = last >= 0 ? (total - hist[c * 256 + last]) / 255 : 0;
  int v = k;
  if (step != 0) {
    v = k == 0 ? 0 : (cum[cur][k - 1] + step / 2) / step;          // pad-left by one: lut[k] = value computed for bin k-1
    v = v > 255 ? 255 : v;
  }
  lut[c * 256 + k] = v;
}

// RAFT.forward head (raft.py:80-83, 132-138): centred replicate pad to /8 and 2*(x/255)-1, both frames into one NHWC4
// batch [2][Hp][Wp][4]; frame 2 goes through the equalisation LUT.
template <typename T>
__device__ __forceinline__ void store_px(T* __restrict__ d, int ld, float a, float b, float c) {
  ZtIO<T>::st(d + 0, a);
  ZtIO<T>::st(d + 1, b);
  ZtIO<T>::st(d + 2, c);
  for (int k = 3; k < ld; ++k) ZtIO<T>::st(d + k, 0.f);
}

template <typename T>
__global__ void __launch_bounds__(256) raft_pack_kernel(const float* __restrict__ img1, const unsigned char* __restrict__ q2,
                                                        const int* __restrict__ lut, T* __restrict__ dst, int ld, int h, int w,
                                                        int Hp, int Wp, int top, int left) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= Wp || y >= Hp) return;
  int sy = min(max(y - top, 0), h - 1), sx = min(max(x - left, 0), w - 1);
  size_t so = (size_t)sy * w + sx;
  float v[3], u[3];
  for (int c = 0; c < 3; ++c) {
    v[c] = 2.f * (img1[(size_t)c * h * w + so] / 255.f) - 1.f;
    float e = (float)lut[c * 256 + q2[(size_t)c * h * w + so]];
    u[c] = 2.f * (e / 255.f) - 1.f;
  }
  size_t o = ((size_t)y * Wp + x) * ld;
  store_px<T>(dst + o, ld, v[0], v[1], v[2]);
  store_px<T>(dst + (size_t)Hp * Wp * ld + o, ld, u[0], u[1], u[2]);
}

// same head for two float frames (RAFT.forward called directly, raft.py:77-83)
template <typename T>
__global__ void __launch_bounds__(256) raft_pack_pair_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                             T* __restrict__ dst, int ld, int h, int w, int Hp, int Wp, int top,
                                                             int left) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= Wp || y >= Hp) return;
  int sy = min(max(y - top, 0), h - 1), sx = min(max(x - left, 0), w - 1);
  size_t so = (size_t)sy * w + sx, hw = (size_t)h * w;
  size_t o = ((size_t)y * Wp + x) * ld;
  store_px<T>(dst + o, ld, 2.f * (img1[so] / 255.f) - 1.f, 2.f * (img1[hw + so] / 255.f) - 1.f, 2.f * (img1[2 * hw + so] / 255.f) - 1.f);
  store_px<T>(dst + (size_t)Hp * Wp * ld + o, ld, 2.f * (img2[so] / 255.f) - 1.f, 2.f * (img2[hw + so] / 255.f) - 1.f,
              2.f * (img2[2 * hw + so] / 255.f) - 1.f);
}

// corr.py:25-27: avg_pool2d(2, stride 2) over the (h2, w2) axes of [npx][hin][win] -> [npx][hin/2][win/2]
__global__ void __launch_bounds__(256) corr_pool_kernel(const float* __restrict__ src, float* __restrict__ dst, int npx,
                                                        int hin, int win, int ldin, int hout, int wout) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)npx * hout * wout;
  if (i >= total) return;
  int x = (int)(i % wout);
  int y = (int)((i / wout) % hout);
  long long n = i / ((long long)wout * hout);
  const float* p = src + n * ldin + (size_t)(2 * y) * win + 2 * x;
  dst[i] = (((p[0] + p[1]) + p[win]) + p[win + 1]) * 0.25f;
}

struct LookupArgs {
  const float* lvl[4];
  int h[4], w[4], ld[4];
  const float* coords;      // [npx][2] (x, y)
  void* out;                // [npx][ldo], fp32 or bf16
  int npx, ldo;
  // fused flow bookkeeping of the PREVIOUS refinement iteration (raft.py:112-120; zt_corr_lookup_step): the coordinates looked up
  // are coords + delta, written to coords_out (a different buffer: other threads of the pixel still read coords) together with
  // flow = coords_out - grid for the up-sampler (f4, fp32) and the two nhwc consumers of the same iteration (fhx, fin; type T)
  const float* delta;
  float* coords_out;
  float* f4;
  void* fhx;
  void* fin;
  int ldd, ldf4, ldfhx, ldfin, w0;
};

// corr.py:29-50 + utils.py:285-299: 4 levels x 9x9 window, bilinear, align_corners=True, zeros padding.
// channel = level*81 + i*9 + j with the FIRST window axis (i) moving x (corr.py:37-43).
template <typename T>
__global__ void __launch_bounds__(256) corr_lookup_kernel(LookupArgs a) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)a.npx * 324) return;
  int ch = (int)(t % 324);
  int n = (int)(t / 324);
  int l = ch / 81, r = ch - l * 81;
  int i = r / 9, j = r - i * 9;
  const int H = a.h[l], W = a.w[l];
  float px = a.coords[n * 2 + 0], py = a.coords[n * 2 + 1];
  if (a.delta) {                                                  // same fp32 add as flow_step_kernel: results are bit-identical
    px += a.delta[(size_t)n * a.ldd + 0];
    py += a.delta[(size_t)n * a.ldd + 1];
  }
  if (ch == 0 && a.coords_out) {
    a.coords_out[n * 2 + 0] = px;
    a.coords_out[n * 2 + 1] = py;
    const float fx = px - (float)(n % a.w0), fy = py - (float)(n / a.w0);
    if (a.f4) {
      a.f4[(size_t)n * a.ldf4 + 0] = fx;
      a.f4[(size_t)n * a.ldf4 + 1] = fy;
    }
    if (a.fhx) {
      ZtIO<T>::st((T*)a.fhx + (size_t)n * a.ldfhx + 0, fx);
      ZtIO<T>::st((T*)a.fhx + (size_t)n * a.ldfhx + 1, fy);
    }
    if (a.fin) {
      ZtIO<T>::st((T*)a.fin + (size_t)n * a.ldfin + 0, fx);
      ZtIO<T>::st((T*)a.fin + (size_t)n * a.ldfin + 1, fy);
    }
  }
  float cx = px / (float)(1 << l) + (float)(i - 4);
  float cy = py / (float)(1 << l) + (float)(j - 4);
  float gx = 2.f * cx / (float)(W - 1) - 1.f;
  float gy = 2.f * cy / (float)(H - 1) - 1.f;
  float ix = (gx + 1.f) * ((float)(W - 1) / 2.f);
  float iy = (gy + 1.f) * ((float)(H - 1) / 2.f);
  float fx0 = floorf(ix), fy0 = floorf(iy);
  float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(fy0, -2.f), (float)H + 1.f);
  const float* p = a.lvl[l] + (size_t)n * a.ld[l];
#define ZT_TAP(xx, yy) (((xx) >= 0 && (xx) < W && (yy) >= 0 && (yy) < H) ? p[(size_t)(yy) * W + (xx)] : 0.f)
  float v = ZT_TAP(x0, y0) * (wx0 * wy0);
  v = fmaf(ZT_TAP(x0 + 1, y0), wx1 * wy0, v);
  v = fmaf(ZT_TAP(x0, y0 + 1), wx0 * wy1, v);
  v = fmaf(ZT_TAP(x0 + 1, y0 + 1), wx1 * wy1, v);
#undef ZT_TAP
  ZtIO<T>::st((T*)a.out + (size_t)n * a.ldo + ch, v);
}

// update.py:42-45, 50-53: rh = r * h  (zr = [z | r] after the sigmoid)
template <typename T>
__global__ void __launch_bounds__(256) gru_rh_kernel(const T* __restrict__ zr, int ldzr, const T* __restrict__ hbuf,
                                                     int ldh, T* __restrict__ rh, int ldrh, int C, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int c = (int)(i % C);
  long long p = i / C;
  ZtIO<T>::st(rh + p * ldrh + c, ZtIO<T>::ld(zr + p * ldzr + C + c) * ZtIO<T>::ld(hbuf + p * ldh + c));
}

// h = (1 - z) * h + z * q
template <typename T>
__global__ void __launch_bounds__(256) gru_update_kernel(const T* __restrict__ zr, int ldzr, const T* __restrict__ q,
                                                         int ldq, T* __restrict__ hbuf, int ldh, int C, long long total) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int c = (int)(i % C);
  long long p = i / C;
  float z = ZtIO<T>::ld(zr + p * ldzr + c), hv = ZtIO<T>::ld(hbuf + p * ldh + c);
  ZtIO<T>::st(hbuf + p * ldh + c, (1.f - z) * hv + z * ZtIO<T>::ld(q + p * ldq + c));
}

// raft.py:112-120: coords1 += delta_flow; flow = coords1 - coords0.  Writes flow to two NHWC destinations.
template <typename T>
__global__ void __launch_bounds__(256) flow_step_kernel(float* __restrict__ coords1, const float* __restrict__ delta, int ldd,
                                                        int w, int npx, float* __restrict__ f4, int ldf4,
                                                        T* __restrict__ fhx, int ldfhx, T* __restrict__ fin, int ldfin) {
  int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= npx) return;
  float x0 = (float)(n % w), y0 = (float)(n / w);
  float cx = coords1[n * 2 + 0], cy = coords1[n * 2 + 1];
  if (delta) {
    cx += delta[(size_t)n * ldd + 0];
    cy += delta[(size_t)n * ldd + 1];
    coords1[n * 2 + 0] = cx;
    coords1[n * 2 + 1] = cy;
  }
  float fx = cx - x0, fy = cy - y0;
  f4[(size_t)n * ldf4 + 0] = fx;
  f4[(size_t)n * ldf4 + 1] = fy;
  if (fhx) {
    ZtIO<T>::st(fhx + (size_t)n * ldfhx + 0, fx);
    ZtIO<T>::st(fhx + (size_t)n * ldfhx + 1, fy);
  }
  if (fin) {
    ZtIO<T>::st(fin + (size_t)n * ldfin + 0, fx);
    ZtIO<T>::st(fin + (size_t)n * ldfin + 1, fy);
  }
}

__global__ void __launch_bounds__(256) coords_init_kernel(float* __restrict__ coords, int w, int npx) {
  int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= npx) return;
  coords[n * 2 + 0] = (float)(n % w);
  coords[n * 2 + 1] = (float)(n / w);
}

// raft.py:64-75 upsample_flow: softmax over the 9 neighbours of mask.view(1,9,8,8,H,W), convex combination of 8*flow
__global__ void __launch_bounds__(256) convex_upsample_kernel(const float* __restrict__ f4, int ldf, const float* __restrict__ mask,
                                                              int ldm, float* __restrict__ up, float* __restrict__ flow_low,
                                                              int h, int w) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)h * w * 64;
  if (t >= total) return;
  int sub = (int)(t & 63);
  int n = (int)(t >> 6);
  int i = sub >> 3, j = sub & 7;
  int y = n / w, x = n - y * w;
  float m[9], mx = -3.4e38f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    m[k] = mask[(size_t)n * ldm + k * 64 + sub];
    mx = fmaxf(mx, m[k]);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    m[k] = expf(m[k] - mx);
    s += m[k];
  }
  float ax = 0.f, ay = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    float fx = 0.f, fy = 0.f;
    if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
      fx = 8.f * f4[(size_t)(yy * w + xx) * ldf + 0];
      fy = 8.f * f4[(size_t)(yy * w + xx) * ldf + 1];
    }
    float wk = m[k] / s;
    ax += wk * fx;
    ay += wk * fy;
  }
  size_t H8 = (size_t)8 * h, W8 = (size_t)8 * w;
  size_t o = (size_t)(8 * y + i) * W8 + 8 * x + j;
  up[o] = ax;
  up[H8 * W8 + o] = ay;
  if (flow_low && sub == 0) {
    flow_low[n] = f4[(size_t)n * ldf + 0];
    flow_low[(size_t)h * w + n] = f4[(size_t)n * ldf + 1];
  }
}

}  // namespace

extern "C" int zt_resize_bilinear_f32(const float* src, float* dst, int C, int H, int W, int h, int w, float mul,
                                      hipStream_t stream) {
  ZT_REQUIRE(src && dst && C > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  float sc_h = (float)H / (float)h, sc_w = (float)W / (float)w;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(zt_cdiv(w, 64), zt_cdiv(h, 4)), dim3(64, 4), 0, stream, src, dst, C, H, W, h,
                     w, sc_h, sc_w, mul);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_equalize_prepare_u8(const float* src, unsigned char* q, int* hist, int* lut, int C, int hw,
                                      hipStream_t stream) {
  ZT_REQUIRE(src && q && hist && lut && C > 0 && C <= 64 && hw > 0);
  hipError_t e = hipMemsetAsync(hist, 0, sizeof(int) * 256 * C, stream);
  if (e != hipSuccess) return (int)e;
  int nb = zt_cdiv(hw, 256 * 8);
  nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);
  hipLaunchKernelGGL(quantize_hist_kernel, dim3(nb, C), dim3(256), 0, stream, src, q, hist, hw);
  hipLaunchKernelGGL(equalize_lut_kernel, dim3(C), dim3(256), 0, stream, (const int*)hist, lut, C);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_raft_pack_input(const float* img1, const unsigned char* q2, const int* lut, void* dst, int dt, int ld, int h,
                                  int w, int Hp, int Wp, hipStream_t stream) {
  ZT_REQUIRE(img1 && q2 && lut && dst && Hp >= h && Wp >= w && Hp % 8 == 0 && Wp % 8 == 0 && Hp - h < 8 && Wp - w < 8 && ld >= 3);
  int top = (Hp - h) / 2, left = (Wp - w) / 2;
  dim3 grid(zt_cdiv(Wp, 64), zt_cdiv(Hp, 4)), block(64, 4);
  if (dt == 0) hipLaunchKernelGGL(raft_pack_kernel<float>, grid, block, 0, stream, img1, q2, lut, (float*)dst, ld, h, w, Hp, Wp, top, left);
  else hipLaunchKernelGGL(raft_pack_kernel<zt_bf16>, grid, block, 0, stream, img1, q2, lut, (zt_bf16*)dst, ld, h, w, Hp, Wp, top, left);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_raft_pack_pair(const float* img1, const float* img2, void* dst, int dt, int ld, int h, int w, int Hp, int Wp,
                                 hipStream_t stream) {
  ZT_REQUIRE(img1 && img2 && dst && Hp >= h && Wp >= w && Hp % 8 == 0 && Wp % 8 == 0 && Hp - h < 8 && Wp - w < 8 && ld >= 3);
  dim3 grid(zt_cdiv(Wp, 64), zt_cdiv(Hp, 4)), block(64, 4);
  if (dt == 0) hipLaunchKernelGGL(raft_pack_pair_kernel<float>, grid, block, 0, stream, img1, img2, (float*)dst, ld, h, w, Hp, Wp, (Hp - h) / 2, (Wp - w) / 2);
  else hipLaunchKernelGGL(raft_pack_pair_kernel<zt_bf16>, grid, block, 0, stream, img1, img2, (zt_bf16*)dst, ld, h, w, Hp, Wp, (Hp - h) / 2, (Wp - w) / 2);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_corr_pool_f32(const float* src, float* dst, int npx, int hin, int win, int ldin, hipStream_t stream) {
  ZT_REQUIRE(src && dst && hin >= 2 && win >= 2 && ldin >= hin * win);
  int hout = hin / 2, wout = win / 2;
  long long total = (long long)npx * hout * wout;
  hipLaunchKernelGGL(corr_pool_kernel, dim3((unsigned)zt_cdivl(total, 256)), dim3(256), 0, stream, src, dst, npx, hin, win, ldin,
                     hout, wout);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_corr_lookup_step(const float* l0, const float* l1, const float* l2, const float* l3, int h, int w, int ld0,
                                   const float* coords, void* out, int dt, int ldo, int npx, const float* delta, int ldd, float* coords_out,
                                   float* f4, int ldf4, void* fhx, int ldfhx, void* fin, int ldfin, hipStream_t stream) {
  ZT_REQUIRE(l0 && l1 && l2 && l3 && coords && out && ldo >= 324);
  ZT_REQUIRE(!coords_out || coords_out != coords);
  ZT_REQUIRE(!delta || coords_out);                 // a delta that is looked up but not recorded would be lost
  LookupArgs a;
  a.delta = delta; a.ldd = ldd; a.coords_out = coords_out; a.f4 = f4; a.ldf4 = ldf4; a.fhx = fhx; a.ldfhx = ldfhx; a.fin = fin;
  a.ldfin = ldfin; a.w0 = w;
  a.lvl[0] = l0; a.lvl[1] = l1; a.lvl[2] = l2; a.lvl[3] = l3;
  int hh = h, ww = w;
  for (int l = 0; l < 4; ++l) {
    a.h[l] = hh; a.w[l] = ww; a.ld[l] = l == 0 ? ld0 : hh * ww;
    ZT_REQUIRE(hh >= 2 && ww >= 2);        // a 1-pixel level divides by zero in the reference (SURVEY A-13)
    hh /= 2; ww /= 2;
  }
  a.coords = coords; a.out = out; a.npx = npx; a.ldo = ldo;
  long long total = (long long)npx * 324;
  if (dt == 0) hipLaunchKernelGGL(corr_lookup_kernel<float>, dim3((unsigned)zt_cdivl(total, 256)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(corr_lookup_kernel<zt_bf16>, dim3((unsigned)zt_cdivl(total, 256)), dim3(256), 0, stream, a);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3, int h, int w, int ld0,
                              const float* coords, void* out, int dt, int ldo, int npx, hipStream_t stream) {
  return zt_corr_lookup_step(l0, l1, l2, l3, h, w, ld0, coords, out, dt, ldo, npx, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, stream);
}

extern "C" int zt_gru_rh(const void* zr, int dt, int ldzr, const void* hbuf, int ldh, void* rh, int ldrh, int C, int npx,
                         hipStream_t stream) {
  ZT_REQUIRE(zr && hbuf && rh);
  long long total = (long long)npx * C;
  dim3 grid((unsigned)zt_cdivl(total, 256));
  if (dt == 0) hipLaunchKernelGGL(gru_rh_kernel<float>, grid, dim3(256), 0, stream, (const float*)zr, ldzr, (const float*)hbuf, ldh, (float*)rh, ldrh, C, total);
  else hipLaunchKernelGGL(gru_rh_kernel<zt_bf16>, grid, dim3(256), 0, stream, (const zt_bf16*)zr, ldzr, (const zt_bf16*)hbuf, ldh, (zt_bf16*)rh, ldrh, C, total);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_gru_update(const void* zr, int dt, int ldzr, const void* q, int ldq, void* hbuf, int ldh, int C, int npx,
                             hipStream_t stream) {
  ZT_REQUIRE(zr && q && hbuf);
  long long total = (long long)npx * C;
  dim3 grid((unsigned)zt_cdivl(total, 256));
  if (dt == 0) hipLaunchKernelGGL(gru_update_kernel<float>, grid, dim3(256), 0, stream, (const float*)zr, ldzr, (const float*)q, ldq, (float*)hbuf, ldh, C, total);
  else hipLaunchKernelGGL(gru_update_kernel<zt_bf16>, grid, dim3(256), 0, stream, (const zt_bf16*)zr, ldzr, (const zt_bf16*)q, ldq, (zt_bf16*)hbuf, ldh, C, total);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_raft_coords_init_f32(float* coords, int h, int w, hipStream_t stream) {
  ZT_REQUIRE(coords && h > 0 && w > 0);
  hipLaunchKernelGGL(coords_init_kernel, dim3(zt_cdiv(h * w, 256)), dim3(256), 0, stream, coords, w, h * w);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_raft_flow_step(float* coords1, const float* delta, int ldd, int h, int w, float* f4, int ldf4, void* fhx,
                                 int ldfhx, void* fin, int ldfin, int dt, hipStream_t stream) {
  ZT_REQUIRE(coords1 && f4);
  dim3 grid(zt_cdiv(h * w, 256));
  if (dt == 0) hipLaunchKernelGGL(flow_step_kernel<float>, grid, dim3(256), 0, stream, coords1, delta, ldd, w, h * w, f4, ldf4, (float*)fhx, ldfhx, (float*)fin, ldfin);
  else hipLaunchKernelGGL(flow_step_kernel<zt_bf16>, grid, dim3(256), 0, stream, coords1, delta, ldd, w, h * w, f4, ldf4, (zt_bf16*)fhx, ldfhx, (zt_bf16*)fin, ldfin);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_convex_upsample_f32(const float* f4, int ldf, const float* mask, int ldm, float* up, float* flow_low, int h,
                                      int w, hipStream_t stream) {
  ZT_REQUIRE(f4 && mask && up && ldm >= 576);
  long long total = (long long)h * w * 64;
  hipLaunchKernelGGL(convex_upsample_kernel, dim3((unsigned)zt_cdivl(total, 256)), dim3(256), 0, stream, f4, ldf, mask, ldm, up,
                     flow_low, h, w);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
