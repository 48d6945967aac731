// Backward warp of the recurrent cache (reference utils/utils.py:203-230 warp_tensor, called twice with the same
// flow at model.py:249-250) fused into ONE pass: coordinate-map construction, bilinear up-sampling of the maps,
// grid normalisation and grid_sample(bilinear, zeros, align_corners=False) for both cached tensors.
//
// Integer contract: the tap indices floor(ix), floor(iy) reproduce ATen's CPU arithmetic bit for bit
// (fma source index, fma row interpolation, ix = fma(g+1, W/2, -0.5)); this file is compiled with
// -ffp-contract=off so only the explicit fmaf() calls fuse.
#include "zt_common.h"

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
  dim3 grid(zt_cdiv(W, 64), zt_cdiv(H, 4)), block(64, 4);
  hipLaunchKernelGGL(warp2_kernel, grid, block, 0, stream, flow, Hf, Wf, imgA, imgB, outA, outB, taps, C, H, W, h_scale,
                     w_scale, sc_h, sc_w, xden, yden, halfW, halfH);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
