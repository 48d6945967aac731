// Backward warp of the recurrent cache (reference utils/utils.py:203-230 warp_tensor, called twice with the same
// flow at model.py:249-250) fused into ONE pass: coordinate-map construction, bilinear up-sampling of the maps,
// grid normalisation and grid_sample(bilinear, zeros, align_corners=False) for both cached tensors.
//
// Integer contract: the tap indices floor(ix), floor(iy) reproduce ATen's CPU arithmetic bit for bit
// (fma source index, fma row interpolation, ix = fma(g+1, W/2, -0.5)); this file is compiled with
// -ffp-contract=off so only the explicit fmaf() calls fuse.
#include "zt_common.h"
#include <stdlib.h>

namespace {

struct Lin1 {
  int i0, i1;
  float w0, w1;
};

// ATen compute_source_index_and_lambda (UpSample.h:451-476), align_corners=False
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
  float lam = src - (float)i0;
  lam = fminf(fmaxf(lam, 0.f), 1.f);
  r.i0 = i0;
  r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  r.w1 = lam;
  r.w0 = 1.f - lam;
  return r;
}

__device__ __forceinline__ float tap(const float* __restrict__ img, int x, int y, int W, int H) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.f;
}

__global__ void __launch_bounds__(256) warp2_kernel(const float* __restrict__ flow, int Hf, int Wf,
                                                    const float* __restrict__ imgA, const float* __restrict__ imgB,
                                                    float* __restrict__ outA, float* __restrict__ outB,
                                                    int* __restrict__ taps, int C, int H, int W,
                                                    float h_scale, float w_scale, float sc_h, float sc_w,
                                                    float xden, float yden, float halfW, float halfH) {
  int x = blockIdx.x * 64 + threadIdx.x;
  int y = blockIdx.y * 4 + threadIdx.y;
  if (x >= W || y >= H) return;
  Lin1 ly = lin_index(y, Hf, H, sc_h);
  Lin1 lx = lin_index(x, Wf, W, sc_w);
  const float* fx = flow;
  const float* fy = flow + (size_t)Hf * Wf;
  // coordinate maps at flow resolution (utils.py:215-216; x scaled by h_scale, y by w_scale -- sic)
  float ax = ((float)lx.i0 - fx[(size_t)ly.i0 * Wf + lx.i0]) * h_scale;
  float bx = ((float)lx.i1 - fx[(size_t)ly.i0 * Wf + lx.i1]) * h_scale;
  float cx = ((float)lx.i0 - fx[(size_t)ly.i1 * Wf + lx.i0]) * h_scale;
  float dx = ((float)lx.i1 - fx[(size_t)ly.i1 * Wf + lx.i1]) * h_scale;
  float ay = ((float)ly.i0 - fy[(size_t)ly.i0 * Wf + lx.i0]) * w_scale;
  float by = ((float)ly.i0 - fy[(size_t)ly.i0 * Wf + lx.i1]) * w_scale;
  float cy = ((float)ly.i1 - fy[(size_t)ly.i1 * Wf + lx.i0]) * w_scale;
  float dy = ((float)ly.i1 - fy[(size_t)ly.i1 * Wf + lx.i1]) * w_scale;
  float mx, my;
  if (Hf == H && Wf == W) {
    mx = ax;
    my = ay;
  } else {
    // ATen upsample_bilinear2d (CPU): row = fma(a, w0, b*w1); out = fma(r0, wy0, r1*wy1)
    float r0 = fmaf(ax, lx.w0, bx * lx.w1), r1 = fmaf(cx, lx.w0, dx * lx.w1);
    mx = fmaf(r0, ly.w0, r1 * ly.w1);
    r0 = fmaf(ay, lx.w0, by * lx.w1);
    r1 = fmaf(cy, lx.w0, dy * lx.w1);
    my = fmaf(r0, ly.w0, r1 * ly.w1);
  }
  float gx = mx / xden - 1.f;       // utils.py:221 (align_corners=True style normalisation)
  float gy = my / yden - 1.f;
  float ix = fmaf(gx + 1.f, halfW, -0.5f);   // ATen grid_sampler unnormalize, align_corners=False
  float iy = fmaf(gy + 1.f, halfH, -0.5f);
  float fx0 = floorf(ix), fy0 = floorf(iy);
  float wx1 = ix - fx0, wy1 = iy - fy0;
  float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
  // clamp before the int conversion so wild flows cannot overflow; anything outside [-1, size] contributes zero anyway
  int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)W + 1.f);
  int y0 = (int)fminf(fmaxf(fy0, -2.f), (float)H + 1.f);
  if (taps) {
    taps[((size_t)y * W + x) * 2 + 0] = (int)fminf(fmaxf(fx0, -2147483000.f), 2147483000.f);
    taps[((size_t)y * W + x) * 2 + 1] = (int)fminf(fmaxf(fy0, -2147483000.f), 2147483000.f);
  }
  float wnw = wx0 * wy0, wne = wx1 * wy0, wsw = wx0 * wy1, wse = wx1 * wy1;
  size_t plane = (size_t)H * W;
  size_t o = (size_t)y * W + x;
  for (int c = 0; c < C; ++c) {
    const float* p = imgA + c * plane;
    float v = tap(p, x0, y0, W, H) * wnw;
    v = fmaf(tap(p, x0 + 1, y0, W, H), wne, v);
    v = fmaf(tap(p, x0, y0 + 1, W, H), wsw, v);
    v = fmaf(tap(p, x0 + 1, y0 + 1, W, H), wse, v);
    outA[c * plane + o] = v;
    if (imgB) {
      const float* q = imgB + c * plane;
      float u = tap(q, x0, y0, W, H) * wnw;
      u = fmaf(tap(q, x0 + 1, y0, W, H), wne, u);
      u = fmaf(tap(q, x0, y0 + 1, W, H), wsw, u);
      u = fmaf(tap(q, x0 + 1, y0 + 1, W, H), wse, u);
      outB[c * plane + o] = u;
    }
  }
}

// Same arithmetic, restructured for memory-level parallelism (the plain kernel above is one pixel per thread with every tap
// behind a bounds branch, i.e. a chain of exposed latencies: 2.7 TB/s of its 101 MB at 1080p).  Here a workgroup owns a
// 64 x (4 PPT) output tile:
//  * the flow window of the tile (both components, <= a few dozen low-resolution rows) is staged ONCE in LDS with coalesced
//    loads, so the coordinate construction reads LDS instead of 8 dependent global gathers per pixel;
//  * a thread handles PPT pixels (rows ty, ty + 4, ..) and issues ALL of their 4 taps x 3 channels x 2 images loads with
//    clamped addresses and no branches before the first use (out-of-image taps are zeroed afterwards: 0 * w and fma(0, w, v)
//    are exact, so results and tap indices stay bit-identical to the reference sequence).
constexpr int WARP_FWW = 68;
template <int PPT, bool TWO>
__global__ void __launch_bounds__(256) warp2_tiled_kernel(const float* __restrict__ flow, int Hf, int Wf,
                                                          const float* __restrict__ imgA, const float* __restrict__ imgB,
                                                          float* __restrict__ outA, float* __restrict__ outB,
                                                          int* __restrict__ taps, int H, int W,
                                                          float h_scale, float w_scale, float sc_h, float sc_w,
                                                          float xden, float yden, float halfW, float halfH) {
  constexpr int fwin_w = WARP_FWW, fwin_h = 4 * PPT + 4;
  __shared__ float fs[2 * fwin_h * fwin_w];                      // flow window [2][fwin_h][fwin_w] (flow resolution <= image resolution)
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
  const int xb = blockIdx.x * 64, yb = blockIdx.y * (4 * PPT);
  // flow window: source rows / columns touched by the tile's first and last pixel (lin_index is monotone in dst)
  const int xe = min(xb + 63, W - 1), ye = min(yb + 4 * PPT - 1, H - 1);
  const int fx0 = lin_index(xb, Wf, W, sc_w).i0, fx1 = lin_index(xe, Wf, W, sc_w).i1;
  const int fy0 = lin_index(yb, Hf, H, sc_h).i0, fy1 = lin_index(ye, Hf, H, sc_h).i1;
  const int ww = fx1 - fx0 + 1, wh = fy1 - fy0 + 1;
  const size_t fplane = (size_t)Hf * Wf;
  for (int e = tid; e < 2 * wh * ww; e += 256) {
    const int c = e / (wh * ww), r = e - c * wh * ww;
    const int yy = r / ww, xx = r - yy * ww;
    fs[(c * fwin_h + yy) * fwin_w + xx] = flow[c * fplane + (size_t)(fy0 + yy) * Wf + fx0 + xx];
  }
  __syncthreads();
  const int x = xb + tx;
  const size_t plane = (size_t)H * W;
  int px0[PPT], py0[PPT];
  float wnw[PPT], wne[PPT], wsw[PPT], wse[PPT];
  const Lin1 lx = lin_index(min(x, W - 1), Wf, W, sc_w);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int y = min(yb + ty + 4 * j, H - 1);
    const Lin1 ly = lin_index(y, Hf, H, sc_h);
    const float* fx = fs + ((ly.i0 - fy0) * fwin_w - fx0);
    const float* gx = fs + ((ly.i1 - fy0) * fwin_w - fx0);
    const float* fy = fx + fwin_h * fwin_w;
    const float* gy = gx + fwin_h * fwin_w;
    const float ax = ((float)lx.i0 - fx[lx.i0]) * h_scale, bx = ((float)lx.i1 - fx[lx.i1]) * h_scale;
    const float cx = ((float)lx.i0 - gx[lx.i0]) * h_scale, dx = ((float)lx.i1 - gx[lx.i1]) * h_scale;
    const float ay = ((float)ly.i0 - fy[lx.i0]) * w_scale, by = ((float)ly.i0 - fy[lx.i1]) * w_scale;
    const float cy = ((float)ly.i1 - gy[lx.i0]) * w_scale, dy = ((float)ly.i1 - gy[lx.i1]) * w_scale;
    float mx, my;
    if (Hf == H && Wf == W) {
      mx = ax;
      my = ay;
    } else {
      float r0 = fmaf(ax, lx.w0, bx * lx.w1), r1 = fmaf(cx, lx.w0, dx * lx.w1);
      mx = fmaf(r0, ly.w0, r1 * ly.w1);
      r0 = fmaf(ay, lx.w0, by * lx.w1);
      r1 = fmaf(cy, lx.w0, dy * lx.w1);
      my = fmaf(r0, ly.w0, r1 * ly.w1);
    }
    const float gxn = mx / xden - 1.f, gyn = my / yden - 1.f;
    const float ix = fmaf(gxn + 1.f, halfW, -0.5f), iy = fmaf(gyn + 1.f, halfH, -0.5f);
    const float fx0f = floorf(ix), fy0f = floorf(iy);
    const float wx1 = ix - fx0f, wy1 = iy - fy0f;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    px0[j] = (int)fminf(fmaxf(fx0f, -2.f), (float)W + 1.f);
    py0[j] = (int)fminf(fmaxf(fy0f, -2.f), (float)H + 1.f);
    if (taps && x < W && yb + ty + 4 * j < H) {
      taps[((size_t)y * W + x) * 2 + 0] = (int)fminf(fmaxf(fx0f, -2147483000.f), 2147483000.f);
      taps[((size_t)y * W + x) * 2 + 1] = (int)fminf(fmaxf(fy0f, -2147483000.f), 2147483000.f);
    }
    wnw[j] = wx0 * wy0;
    wne[j] = wx1 * wy0;
    wsw[j] = wx0 * wy1;
    wse[j] = wx1 * wy1;
  }
  constexpr int NI = TWO ? 2 : 1;
  float t[PPT][NI][3][4];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int xa = min(max(px0[j], 0), W - 1), xc = min(max(px0[j] + 1, 0), W - 1);
    const int ya = min(max(py0[j], 0), H - 1), yc = min(max(py0[j] + 1, 0), H - 1);
    const size_t o00 = (size_t)ya * W + xa, o01 = (size_t)ya * W + xc, o10 = (size_t)yc * W + xa, o11 = (size_t)yc * W + xc;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float* img = i ? imgB : imgA;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* p = img + c * plane;
        t[j][i][c][0] = p[o00];
        t[j][i][c][1] = p[o01];
        t[j][i][c][2] = p[o10];
        t[j][i][c][3] = p[o11];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int y = yb + ty + 4 * j;
    const bool xin0 = px0[j] >= 0 && px0[j] < W, xin1 = px0[j] + 1 >= 0 && px0[j] + 1 < W;
    const bool yin0 = py0[j] >= 0 && py0[j] < H, yin1 = py0[j] + 1 >= 0 && py0[j] + 1 < H;
    if (x < W && y < H) {
      const size_t o = (size_t)y * W + x;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float* out = i ? outB : outA;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v = ((xin0 && yin0) ? t[j][i][c][0] : 0.f) * wnw[j];
          v = fmaf((xin1 && yin0) ? t[j][i][c][1] : 0.f, wne[j], v);
          v = fmaf((xin0 && yin1) ? t[j][i][c][2] : 0.f, wsw[j], v);
          v = fmaf((xin1 && yin1) ? t[j][i][c][3] : 0.f, wse[j], v);
          out[c * plane + o] = v;
        }
      }
    }
  }
}

}  // namespace

extern "C" int zt_warp2_f32(const float* flow, int Hf, int Wf, const float* imgA, const float* imgB, float* outA,
                            float* outB, int* taps, int C, int H, int W, hipStream_t stream) {
  ZT_REQUIRE(flow && imgA && outA && Hf > 0 && Wf > 0 && H > 1 && W > 1 && C > 0);
  ZT_REQUIRE((imgB == nullptr) == (outB == nullptr));
  float h_scale = (float)((double)H / (double)Hf);       // python float -> fp32 scalar
  float w_scale = (float)((double)W / (double)Wf);
  float sc_h = (float)Hf / (float)H, sc_w = (float)Wf / (float)W;   // area_pixel_compute_scale<float>
  float xden = (float)(((double)W - 1.0) / 2.0), yden = (float)(((double)H - 1.0) / 2.0);
  float halfW = (float)W / 2.f, halfH = (float)H / 2.f;
  if (C == 3) {       // the path's only use (3-channel frames): tiled kernel, flow window in LDS
    static const int ppt = getenv("ZT_WARP_PPT") ? atoi(getenv("ZT_WARP_PPT")) : 2;      // tuning hook: pixels (rows) per thread, 2 or 4
    const int PPT = ppt == 4 ? 4 : 2;
    // upper bound of the flow window of a 64 x (4 PPT) tile: ceil(extent * scale) + 2 (the i0..i1 span of both ends)
    const int fwin_w = (int)(64.0 * Wf / W) + 3, fwin_h = (int)(4.0 * PPT * Hf / H) + 3;
    if (fwin_w <= WARP_FWW && fwin_h <= 4 * PPT + 4) {
      dim3 grid(zt_cdiv(W, 64), zt_cdiv(H, 4 * PPT)), block(64, 4);
#define ZT_WARP_LAUNCH(P, T)                                                                                                          \
  hipLaunchKernelGGL((warp2_tiled_kernel<P, T>), grid, block, 0, stream, flow, Hf, Wf, imgA, imgB, outA, outB, taps, H, W, h_scale, w_scale, \
                     sc_h, sc_w, xden, yden, halfW, halfH)
      if (PPT == 4) {
        if (imgB) ZT_WARP_LAUNCH(4, true); else ZT_WARP_LAUNCH(4, false);
      } else {
        if (imgB) ZT_WARP_LAUNCH(2, true); else ZT_WARP_LAUNCH(2, false);
      }
#undef ZT_WARP_LAUNCH
      ZT_LAUNCH_CHECK();
      return ZT_OK;
    }
  }
  dim3 grid(zt_cdiv(W, 64), zt_cdiv(H, 4)), block(64, 4);
  hipLaunchKernelGGL(warp2_kernel, grid, block, 0, stream, flow, Hf, Wf, imgA, imgB, outA, outB, taps, C, H, W, h_scale,
                     w_scale, sc_h, sc_w, xden, yden, halfW, halfH);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
