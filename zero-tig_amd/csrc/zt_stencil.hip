// Stencil primitives of the Zero-TIG hot path on planar fp32 [C][H][W] tensors, with the adjoints the
// hand-written backward pass needs.  All are HBM-bound (a few reads + one write per element).
//   pair_downsampler      utils/utils.py:15-24
//   blur (21x21 Gaussian) utils/utils.py:26-39, 52-58      (rank-1 kernel -> two 21-tap passes)
//   LocalMean             utils/utils.py:41-50
//   calculate_local_variance utils/utils.py:60-79
//   TextureDifference     loss.py:99-136
//   SmoothLoss.rgb2yCbCr  loss.py:178-190
#include "zt_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ pair downsample
__global__ void __launch_bounds__(256) pair_down_kernel(const float* __restrict__ src, float* __restrict__ o1,
                                                        float* __restrict__ o2, int C, int H, int W, int h, int w) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= w || y >= h) return;
  for (int c = 0; c < C; ++c) {
    const float* p = src + (size_t)c * H * W + (size_t)(2 * y) * W + 2 * x;
    float a = p[0], b = p[1], cc = p[W], d = p[W + 1];
    o1[(size_t)c * h * w + (size_t)y * w + x] = 0.5f * b + 0.5f * cc;
    o2[(size_t)c * h * w + (size_t)y * w + x] = 0.5f * a + 0.5f * d;
  }
}

// dst (+)= adjoint(pair_down)(g1, g2); rows/cols beyond 2h/2w receive zero
__global__ void __launch_bounds__(256) pair_down_adj_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                            float* __restrict__ dst, int C, int H, int W, int h, int w,
                                                            int accumulate) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= W || y >= H) return;
  int hy = y >> 1, hx = x >> 1;
  bool inside = hy < h && hx < w;
  bool diag = ((y ^ x) & 1) == 0;       // (even,even) and (odd,odd) feed output 2; the anti-diagonal feeds output 1
  for (int c = 0; c < C; ++c) {
    float v = 0.f;
    if (inside) v = 0.5f * (diag ? g2 : g1)[(size_t)c * h * w + (size_t)hy * w + hx];
    size_t o = (size_t)c * H * W + (size_t)y * W + x;
    dst[o] = accumulate ? dst[o] + v : v;
  }
}

// ------------------------------------------------------------------------------------------------ 21-tap separable blur
struct Taps21 {
  float t[21];
};

// adjoint of one reflect-padded pass: g_ext(u) = sum_j t_j * gy(u - j) on the extended domain, folded back
template <bool VERT>
__device__ __forceinline__ float blur_ext(const float* __restrict__ p, int fixed, int u, int n, int W, const Taps21& k) {
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 21; ++j) {
    int i = u - (j - 10);
    if (i >= 0 && i < n) acc = fmaf(k.t[j], VERT ? p[(size_t)i * W + fixed] : p[(size_t)fixed * W + i], acc);
  }
  return acc;
}

// Vertical passes as a register sliding window: a thread owns one column and BRY consecutive output rows, loads the BRY + 20
// inputs it needs once (coalesced across the wave) and keeps them in registers -- 2.25 loads per output instead of 21 that
// miss L1 (rows are 7.7 KB apart).  Same fmaf order as the plain kernels: bit-identical results.
constexpr int BRY = 16;

template <bool ADJ>
__global__ void __launch_bounds__(256) blur_vert_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                                        Taps21 k, int accumulate) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int r0 = (blockIdx.y * 4 + threadIdx.y) * BRY;
  if (x >= W || r0 >= H) return;
  for (int c = 0; c < C; ++c) {
    const float* p = src + (size_t)c * H * W;
    float win[BRY + 20];
#pragma unroll
    for (int m = 0; m < BRY + 20; ++m) {
      const int i = r0 - 10 + m;
      if (ADJ) win[m] = (i >= 0 && i < H) ? p[(size_t)i * W + x] : 0.f;                // extended-domain correlation: zero outside
      else win[m] = p[(size_t)zt_reflect(i < H + 10 ? i : H + 9, H) * W + x];           // reflect padding (rows past the last segment unused)
    }
#pragma unroll
    for (int r = 0; r < BRY; ++r) {
      const int y = r0 + r;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 21; ++j) acc = fmaf(k.t[j], ADJ ? win[r + 20 - j] : win[r + j], acc);
      if (y < H) {
        const size_t o = (size_t)c * H * W + (size_t)y * W + x;
        dst[o] = (ADJ && accumulate) ? dst[o] + acc : acc;
      }
    }
  }
}

// Horizontal passes through an LDS row segment (256 outputs + 20 halo): one coalesced global load per input, 21 conflict-free
// LDS reads per output; same fmaf order as the plain kernels.
template <bool ADJ>
__global__ void __launch_bounds__(256) blur_horz_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                                        Taps21 k, int accumulate) {
  __shared__ float buf[256 + 20];
  const int x0 = blockIdx.x * 256, y = blockIdx.y, x = x0 + threadIdx.x;
  for (int c = 0; c < C; ++c) {
    const float* p = src + ((size_t)c * H + y) * W;
    for (int i = threadIdx.x; i < 256 + 20; i += 256) {
      const int gx = x0 - 10 + i;
      if (ADJ) buf[i] = (gx >= 0 && gx < W) ? p[gx] : 0.f;
      else buf[i] = p[zt_reflect(gx < W + 10 ? gx : W + 9, W)];
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 21; ++j) acc = fmaf(k.t[j], ADJ ? buf[threadIdx.x + 20 - j] : buf[threadIdx.x + j], acc);
    if (x < W) {
      const size_t o = ((size_t)c * H + y) * W + x;
      dst[o] = (ADJ && accumulate) ? dst[o] + acc : acc;
    }
    __syncthreads();
  }
}

// second half of the horizontal adjoint: fold columns 1..10 and W-11..W-2, then (optionally) accumulate into the destination
__global__ void __launch_bounds__(256) blur_horz_fold_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                                             Taps21 k) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  const bool left = blockIdx.y < 10;                            // W > 21: the two column groups are disjoint
  const int x = left ? 1 + blockIdx.y : W - 11 + (blockIdx.y - 10);
  if (y >= H) return;
  for (int c = 0; c < C; ++c) {
    const float* p = src + (size_t)c * H * W;
    dst[(size_t)c * H * W + (size_t)y * W + x] += blur_ext<false>(p, y, left ? -x : 2 * (W - 1) - x, W, W, k);
  }
}

// second half of the vertical adjoint: fold the reflected borders back into rows 1..10 and H-11..H-2 (20 rows only)
__global__ void __launch_bounds__(256) blur_vert_fold_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int H, int W,
                                                             Taps21 k) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const bool top = blockIdx.y < 10;                             // H > 21 (checked by the launcher): the two row groups are disjoint
  const int y = top ? 1 + blockIdx.y : H - 11 + (blockIdx.y - 10);
  if (x >= W) return;
  for (int c = 0; c < C; ++c) {
    const float* p = src + (size_t)c * H * W;
    const size_t o = (size_t)c * H * W + (size_t)y * W + x;
    dst[o] += blur_ext<true>(p, x, top ? -y : 2 * (H - 1) - y, H, W, k);
  }
}

// ------------------------------------------------------------------------------------------------ 5x5 reflect mean
__global__ void __launch_bounds__(256) box5_reflect_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                           int H, int W) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= W || y >= H) return;
  for (int c = 0; c < C; ++c) {
    const float* p = src + (size_t)c * H * W;
    float acc = 0.f;
    for (int dy = -2; dy <= 2; ++dy) {
      int yy = zt_reflect(y + dy, H);
      for (int dx = -2; dx <= 2; ++dx) acc += p[(size_t)yy * W + zt_reflect(x + dx, W)];
    }
    dst[(size_t)c * H * W + (size_t)y * W + x] = acc / 25.f;
  }
}

__device__ __forceinline__ int fold_sources(int k, int n, int* u) {
  int m = 0;
  u[m++] = k;
  if (k >= 1 && k <= 2) u[m++] = -k;
  if (k <= n - 2 && k >= n - 3) u[m++] = 2 * (n - 1) - k;
  return m;
}

// dst = scale_out * adjoint(box5_reflect)(src)   (accumulate: dst += ...)
__global__ void __launch_bounds__(256) box5_reflect_adj_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               int C, int H, int W, float scale, int accumulate) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= W || y >= H) return;
  int uy[3], ux[3];
  int ny = fold_sources(y, H, uy), nx = fold_sources(x, W, ux);
  for (int c = 0; c < C; ++c) {
    const float* p = src + (size_t)c * H * W;
    float acc = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b)
        for (int dy = -2; dy <= 2; ++dy) {
          int yy = uy[a] - dy;
          if (yy < 0 || yy >= H) continue;
          for (int dx = -2; dx <= 2; ++dx) {
            int xx = ux[b] - dx;
            if (xx >= 0 && xx < W) acc += p[(size_t)yy * W + xx];
          }
        }
    acc = acc / 25.f * scale;
    size_t o = (size_t)c * H * W + (size_t)y * W + x;
    dst[o] = accumulate ? dst[o] + acc : acc;
  }
}

// ------------------------------------------------------------------------------------------------ local variance (zero pad)
// D = x - box0(x)/25 ; V = box0(D^2)/25     (x = a - b when b != nullptr)
// 64 x 16 tiles, both 5x5 boxes separable in LDS (5 + 5 reads per output instead of 25, halo ratio 1.7 instead of 2.5): the
// 32 x 8 / 25-tap form ran at ~0.9 TB/s of its 50-100 MB.
#define LV_TX 64
#define LV_TY 16
constexpr int LV_XW = LV_TX + 8, LV_XH = LV_TY + 8;      // input tile with the +-4 halo of the two stacked boxes
constexpr int LV_DW = LV_TX + 4, LV_DH = LV_TY + 4;      // intermediate (D / E) tile with the +-2 halo of the second box

// hs[r][c] = sum_{dx < 5} xs[r][c + dx]  (LV_XH x LV_DW)
__device__ __forceinline__ void lv_hsum_in(const float (*xs)[LV_XW], float (*hs)[LV_DW], int tid) {
  for (int i = tid; i < LV_XH * LV_DW; i += 256) {
    const int r = i / LV_DW, c = i - r * LV_DW;
    hs[r][c] = (((xs[r][c] + xs[r][c + 1]) + xs[r][c + 2]) + xs[r][c + 3]) + xs[r][c + 4];
  }
}

__global__ void __launch_bounds__(256) localvar_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ D, float* __restrict__ V, int C, int H,
                                                           int W) {
  __shared__ float xs[LV_XH][LV_XW];
  __shared__ float hs[LV_XH][LV_DW];
  __shared__ float ds[LV_DH][LV_DW];
  __shared__ float h2[LV_DH][LV_TX];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * LV_TX, y0 = blockIdx.y * LV_TY;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const float* pa = a + (size_t)c * H * W;
  const float* pb = b ? b + (size_t)c * H * W : nullptr;
  for (int i = tid; i < LV_XH * LV_XW; i += 256) {
    const int ly = i / LV_XW, lx = i - ly * LV_XW;
    const int gy = y0 + ly - 4, gx = x0 + lx - 4;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t o = (size_t)(in ? gy : 0) * W + (in ? gx : 0);
    float v = pa[o];
    if (pb) v -= pb[o];
    xs[ly][lx] = in ? v : 0.f;
  }
  __syncthreads();
  lv_hsum_in(xs, hs, tid);
  __syncthreads();
  for (int i = tid; i < LV_DH * LV_DW; i += 256) {
    const int r = i / LV_DW, cc = i - r * LV_DW;
    const int gy = y0 + r - 2, gx = x0 + cc - 2;
    float d = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const float sum = (((hs[r][cc] + hs[r + 1][cc]) + hs[r + 2][cc]) + hs[r + 3][cc]) + hs[r + 4][cc];
      d = xs[r + 2][cc + 2] - sum / 25.f;
      if (D && r >= 2 && r < LV_TY + 2 && cc >= 2 && cc < LV_TX + 2) D[(size_t)c * H * W + (size_t)gy * W + gx] = d;
    }
    ds[r][cc] = d;
  }
  __syncthreads();
  for (int i = tid; i < LV_DH * LV_TX; i += 256) {
    const int r = i / LV_TX, cc = i - r * LV_TX;
    const float d0 = ds[r][cc], d1 = ds[r][cc + 1], d2 = ds[r][cc + 2], d3 = ds[r][cc + 3], d4 = ds[r][cc + 4];
    h2[r][cc] = (((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3) + d4 * d4;
  }
  __syncthreads();
  const int gx = x0 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < LV_TY / 4; ++j) {
    const int ly = threadIdx.y + 4 * j, gy = y0 + ly;
    if (gx < W && gy < H) {
      const float sum = (((h2[ly][threadIdx.x] + h2[ly + 1][threadIdx.x]) + h2[ly + 2][threadIdx.x]) + h2[ly + 3][threadIdx.x]) + h2[ly + 4][threadIdx.x];
      V[(size_t)c * H * W + (size_t)gy * W + gx] = sum / 25.f;
    }
  }
}

// backward: xbar (+)= sign * (E - box0(E)/25), E = 2 D box0(gV)/25
__global__ void __launch_bounds__(256) localvar_bwd_kernel(const float* __restrict__ D, const float* __restrict__ gV,
                                                           float* __restrict__ xbar, int C, int H, int W, float sign,
                                                           int accumulate) {
  __shared__ float gs[LV_XH][LV_XW];
  __shared__ float hs[LV_XH][LV_DW];
  __shared__ float es[LV_DH][LV_DW];
  __shared__ float h2[LV_DH][LV_TX];
  const int c = blockIdx.z;
  const int x0 = blockIdx.x * LV_TX, y0 = blockIdx.y * LV_TY;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const float* pg = gV + (size_t)c * H * W;
  const float* pd = D + (size_t)c * H * W;
  for (int i = tid; i < LV_XH * LV_XW; i += 256) {
    const int ly = i / LV_XW, lx = i - ly * LV_XW;
    const int gy = y0 + ly - 4, gx = x0 + lx - 4;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float v = pg[(size_t)(in ? gy : 0) * W + (in ? gx : 0)];
    gs[ly][lx] = in ? v : 0.f;
  }
  __syncthreads();
  lv_hsum_in(gs, hs, tid);
  __syncthreads();
  for (int i = tid; i < LV_DH * LV_DW; i += 256) {
    const int r = i / LV_DW, cc = i - r * LV_DW;
    const int gy = y0 + r - 2, gx = x0 + cc - 2;
    float e = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const float sum = (((hs[r][cc] + hs[r + 1][cc]) + hs[r + 2][cc]) + hs[r + 3][cc]) + hs[r + 4][cc];
      e = 2.f * pd[(size_t)gy * W + gx] * (sum / 25.f);
    }
    es[r][cc] = e;
  }
  __syncthreads();
  for (int i = tid; i < LV_DH * LV_TX; i += 256) {
    const int r = i / LV_TX, cc = i - r * LV_TX;
    h2[r][cc] = (((es[r][cc] + es[r][cc + 1]) + es[r][cc + 2]) + es[r][cc + 3]) + es[r][cc + 4];
  }
  __syncthreads();
  const int gx = x0 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < LV_TY / 4; ++j) {
    const int ly = threadIdx.y + 4 * j, gy = y0 + ly;
    if (gx < W && gy < H) {
      const float sum = (((h2[ly][threadIdx.x] + h2[ly + 1][threadIdx.x]) + h2[ly + 2][threadIdx.x]) + h2[ly + 3][threadIdx.x]) + h2[ly + 4][threadIdx.x];
      const float v = sign * (es[ly + 2][threadIdx.x + 2] - sum / 25.f);
      const size_t o = (size_t)c * H * W + (size_t)gy * W + gx;
      xbar[o] = accumulate ? xbar[o] + v : v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ texture mask
__device__ __forceinline__ float gray144(const float* __restrict__ p, size_t plane, size_t o) {
  return 0.144f * p[o] + 0.587f * p[plane + o] + 0.299f * p[2 * plane + o];
}

__device__ __forceinline__ float local_std5(const float* __restrict__ p, int x, int y, int H, int W) {
  size_t plane = (size_t)H * W;
  float v[25];
  float s = 0.f;
  int n = 0;
  for (int dy = -2; dy <= 2; ++dy) {
    int yy = zt_reflect(y + dy, H);
    for (int dx = -2; dx <= 2; ++dx) {
      float g = gray144(p, plane, (size_t)yy * W + zt_reflect(x + dx, W));
      v[n++] = g;
      s += g;
    }
  }
  float mu = s / 25.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    float d = v[i] - mu;
    q += d * d;
  }
  return sqrtf(q / 25.f + 1e-9f);
}

__global__ void __launch_bounds__(256) texture_mask_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ mask, float* __restrict__ ratio, int H,
                                                           int W) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= W || y >= H) return;
  float s1 = local_std5(a, x, y, H, W), s2 = local_std5(b, x, y, H, W);
  float r = (2.f * s1 * s2) / (s1 * s1 + s2 * s2 + 1e-5f);
  size_t o = (size_t)y * W + x;
  mask[o] = r > 0.975f ? 1.f : 0.f;
  if (ratio) ratio[o] = r;
}

// ------------------------------------------------------------------------------------------------ "YCbCr" over flat memory
__global__ void __launch_bounds__(256) ycc_flat_kernel(const float* __restrict__ src, float* __restrict__ dst, long long ntriples) {
  long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= ntriples) return;
  float r = src[3 * t], g = src[3 * t + 1], b = src[3 * t + 2];
  // im_flat.mm(mat) + bias, mat rows = input element, cols = output element (loss.py:182-186)
  dst[3 * t + 0] = (r * 0.257f + g * 0.564f + b * 0.098f) + (float)(16.0 / 255.0);
  dst[3 * t + 1] = (r * -0.148f + g * -0.291f + b * 0.439f) + (float)(128.0 / 255.0);
  dst[3 * t + 2] = (r * 0.439f + g * -0.368f + b * -0.071f) + (float)(128.0 / 255.0);
}

inline dim3 grid2d(int W, int H) { return dim3(zt_cdiv(W, 64), zt_cdiv(H, 4)); }

}  // namespace

extern "C" int zt_pair_down_f32(const float* src, float* o1, float* o2, int C, int H, int W, hipStream_t stream) {
  ZT_REQUIRE(src && o1 && o2 && C > 0 && H >= 2 && W >= 2);
  int h = H / 2, w = W / 2;
  hipLaunchKernelGGL(pair_down_kernel, grid2d(w, h), dim3(64, 4), 0, stream, src, o1, o2, C, H, W, h, w);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_pair_down_adj_f32(const float* g1, const float* g2, float* dst, int C, int H, int W, int accumulate,
                                    hipStream_t stream) {
  ZT_REQUIRE(g1 && g2 && dst && C > 0 && H >= 2 && W >= 2);
  hipLaunchKernelGGL(pair_down_adj_kernel, grid2d(W, H), dim3(64, 4), 0, stream, g1, g2, dst, C, H, W, H / 2, W / 2, accumulate);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_blur21_f32(const float* src, float* tmp, float* dst, const float* taps21_host, int C, int H, int W,
                             hipStream_t stream) {
  ZT_REQUIRE(src && tmp && dst && taps21_host && H > 10 && W > 10);
  Taps21 k;
  for (int i = 0; i < 21; ++i) k.t[i] = taps21_host[i];
  hipLaunchKernelGGL(blur_horz_kernel<false>, dim3(zt_cdiv(W, 256), H), dim3(256), 0, stream, src, tmp, C, H, W, k, 0);
  hipLaunchKernelGGL(blur_vert_kernel<false>, dim3(zt_cdiv(W, 64), zt_cdiv(H, 4 * BRY)), dim3(64, 4), 0, stream, (const float*)tmp, dst, C, H, W, k, 0);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_blur21_adj_f32(const float* g, float* tmp, float* dst, const float* taps21_host, int C, int H, int W,
                                 int accumulate, hipStream_t stream) {
  ZT_REQUIRE(g && tmp && dst && taps21_host && H > 21 && W > 21);
  Taps21 k;
  for (int i = 0; i < 21; ++i) k.t[i] = taps21_host[i];
  hipLaunchKernelGGL(blur_vert_kernel<true>, dim3(zt_cdiv(W, 64), zt_cdiv(H, 4 * BRY)), dim3(64, 4), 0, stream, g, tmp, C, H, W, k, 0);
  hipLaunchKernelGGL(blur_vert_fold_kernel, dim3(zt_cdiv(W, 256), 20), dim3(256), 0, stream, g, tmp, C, H, W, k);
  hipLaunchKernelGGL(blur_horz_kernel<true>, dim3(zt_cdiv(W, 256), H), dim3(256), 0, stream, (const float*)tmp, dst, C, H, W, k, accumulate);
  hipLaunchKernelGGL(blur_horz_fold_kernel, dim3(zt_cdiv(H, 256), 20), dim3(256), 0, stream, (const float*)tmp, dst, C, H, W, k);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_box5_reflect_f32(const float* src, float* dst, int C, int H, int W, hipStream_t stream) {
  ZT_REQUIRE(src && dst && H > 2 && W > 2);
  hipLaunchKernelGGL(box5_reflect_kernel, grid2d(W, H), dim3(64, 4), 0, stream, src, dst, C, H, W);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_box5_reflect_adj_f32(const float* src, float* dst, int C, int H, int W, float scale, int accumulate,
                                       hipStream_t stream) {
  ZT_REQUIRE(src && dst && H > 5 && W > 5);
  hipLaunchKernelGGL(box5_reflect_adj_kernel, grid2d(W, H), dim3(64, 4), 0, stream, src, dst, C, H, W, scale, accumulate);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_localvar_fwd_f32(const float* a, const float* b, float* D, float* V, int C, int H, int W,
                                   hipStream_t stream) {
  ZT_REQUIRE(a && V && C > 0);
  dim3 grid(zt_cdiv(W, LV_TX), zt_cdiv(H, LV_TY), C);
  hipLaunchKernelGGL(localvar_fwd_kernel, grid, dim3(64, 4), 0, stream, a, b, D, V, C, H, W);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_localvar_bwd_f32(const float* D, const float* gV, float* xbar, int C, int H, int W, float sign,
                                   int accumulate, hipStream_t stream) {
  ZT_REQUIRE(D && gV && xbar && C > 0);
  dim3 grid(zt_cdiv(W, LV_TX), zt_cdiv(H, LV_TY), C);
  hipLaunchKernelGGL(localvar_bwd_kernel, grid, dim3(64, 4), 0, stream, D, gV, xbar, C, H, W, sign, accumulate);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_texture_mask_f32(const float* a, const float* b, float* mask, float* ratio, int H, int W,
                                   hipStream_t stream) {
  ZT_REQUIRE(a && b && mask && H > 2 && W > 2);
  hipLaunchKernelGGL(texture_mask_kernel, grid2d(W, H), dim3(64, 4), 0, stream, a, b, mask, ratio, H, W);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_ycc_flat_f32(const float* src, float* dst, long long nelem, hipStream_t stream) {
  ZT_REQUIRE(src && dst && nelem % 3 == 0);
  long long nt = nelem / 3;
  hipLaunchKernelGGL(ycc_flat_kernel, dim3((unsigned)zt_cdivl(nt, 256)), dim3(256), 0, stream, src, dst, nt);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
