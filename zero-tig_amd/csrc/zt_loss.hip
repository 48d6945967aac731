// LossFunction.forward (reference loss.py:12-78) and its gradient w.r.t. every network output, evaluated together:
// each kernel produces the weighted partial sums of its terms AND the gradients of those terms, so the 22 input
// tensors are read once instead of being expanded into the reference's ~100 temporaries.
// Term order (17): 0 enh_s2 1 enh_norm 2 smooth 3 tv | 4..7 res1_a..d | 8..11 res2_a..d | 12 color 13 ill |
// 14 inter_a 15 inter_b | 16 var.   Every partial is already multiplied by its weight / element count.
#include "zt_common.h"

namespace {

// ---- per-channel sums of a planar tensor: partial[blk][c]
// All C <= 4 planes of a block's pixel range in one sweep, 16-byte loads, eight independent loads in flight per thread and plane
// (the one-plane-at-a-time scalar loop with a block reduction per plane read 25 MB in 44 us).
__global__ void __launch_bounds__(256) plane_sums_kernel(const float* __restrict__ x, int C, long long HW, int nblk,
                                                         float* __restrict__ partial) {
  __shared__ float red[16 * 4];
  long long chunk = ((HW + nblk - 1) / nblk + 3) & ~3ll;         // multiple of 4: 16-byte aligned block ranges (HW % 4 == 0 checked by the host)
  const long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk < HW ? p0 + chunk : HW;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (long long p = p0 + 4 * threadIdx.x; p < p1; p += 8 * 1024) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < C) {
        float4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long long q = p + j * 1024;
          t[j] = *reinterpret_cast<const float4*>(x + (size_t)c * HW + (q < p1 ? q : p0));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c] += (p + j * 1024 < p1) ? ((t[j].x + t[j].y) + (t[j].z + t[j].w)) : 0.f;
      }
    }
  }
  zt_block_sum<4>(v, red);
  if (threadIdx.x == 0)
    for (int c = 0; c < C; ++c) partial[blockIdx.x * C + c] = v[c];
}

__global__ void __launch_bounds__(256) plane_sums_scalar_kernel(const float* __restrict__ x, int C, long long HW, int nblk,
                                                                float* __restrict__ partial) {
  __shared__ float red[16 * 4];
  long long chunk = (HW + nblk - 1) / nblk;
  long long p0 = (long long)blockIdx.x * chunk, p1 = p0 + chunk < HW ? p0 + chunk : HW;
  for (int c = 0; c < C; ++c) {
    float v[1] = {0.f};
    for (long long p = p0 + threadIdx.x; p < p1; p += 256) v[0] += x[(size_t)c * HW + p];
    zt_block_sum<1>(v, red);
    if (threadIdx.x == 0) partial[blockIdx.x * C + c] = v[0];
    __syncthreads();
  }
}

// loss.py:26-37: enhancement_factor ef[3] and adjustment_ratio[3] -> scal[0..2], scal[3..5]
__global__ void loss_scalars_kernel(const float* __restrict__ partial, int nblk, double HW, int is_WB, float* __restrict__ scal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s[3] = {0, 0, 0};
  for (int b = 0; b < nblk; ++b)
    for (int c = 0; c < 3; ++c) s[c] += (double)partial[b * 3 + c];
  float ef[3];
  if (is_WB) {
    for (int c = 0; c < 3; ++c) ef[c] = 0.3f / ((float)(s[c] / HW) + 1e-9f);
  } else {
    float m = (float)((0.299 * s[2] + 0.587 * s[1] + 0.144 * s[0]) / HW);
    ef[0] = ef[1] = ef[2] = 0.5f / (m + 1e-9f);
  }
  for (int c = 0; c < 3; ++c) {
    float e = fminf(fmaxf(ef[c], 1.f), 25.f);
    scal[c] = e;
    scal[3 + c] = powf(0.7f, -e) / e;
  }
}

constexpr int k_off_dy[12] = {1, 0, 1, 1, 2, 0, 2, 2, 1, 1, 2, 2};
constexpr int k_off_dx[12] = {0, 1, 1, -1, 0, 2, 1, -1, 2, -2, 2, -2};
struct SmoothCoef {
  float coef[12];
};

// Terms 0-3 (loss.py:46-49) and their direct gradient w.r.t. s2.  L2 is detached in all four.
// A workgroup owns a 64 x 16 tile; the s2 and Y planes of the tile plus the +-2 halo that the 24 shifted differences of
// SmoothLoss and the TV neighbours reach are staged ONCE in LDS (6 planes x 20 x 68 floats = 32 KB, coalesced loads); the
// 150 neighbour reads per pixel then hit LDS instead of the texture path (the per-pixel global gathers made this kernel run
// at 0.7 TB/s of its 100 MB).  Per-pixel arithmetic and its order are unchanged.
constexpr int LS_TX = 64, LS_TY = 16, LS_HALO = 2, LS_PW = LS_TX + 2 * LS_HALO, LS_PH = LS_TY + 2 * LS_HALO;

// FAST (bf16 throughput mode): the 24 bilateral weights per pixel use the hardware exponential (__expf: v_exp_f32 of x log2 e,
// ~1e-7 relative for these |x| < 1) instead of libm's ~10-instruction expansion -- the kernel is bound by vector-instruction issue
template <bool FAST>
__global__ void __launch_bounds__(256) loss_s2_kernel(const float* __restrict__ L2, const float* __restrict__ s2,
                                                      const float* __restrict__ Y, const float* __restrict__ scal, int H,
                                                      int W, float* __restrict__ ds2, float* __restrict__ partial, int nrow4, SmoothCoef sc) {
  __shared__ float red[16 * 4];
  __shared__ float sS[3][LS_PH][LS_PW];
  __shared__ float sY[3][LS_PH][LS_PW];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int x0 = blockIdx.x * LS_TX, y0 = blockIdx.y * LS_TY;
  const size_t HW = (size_t)H * W;
  for (int e = tid; e < 6 * LS_PH * LS_PW; e += 256) {
    const int pl = e / (LS_PH * LS_PW), r = e - pl * (LS_PH * LS_PW);
    const int ly = r / LS_PW, lx = r - ly * LS_PW;
    const int gy = y0 + ly - LS_HALO, gx = x0 + lx - LS_HALO;
    const int gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
    const float* src = pl < 3 ? s2 + pl * HW : Y + (pl - 3) * HW;
    const float v = src[(size_t)gyc * W + gxc];                 // out-of-image slots are never used (bounds tests below)
    if (pl < 3) sS[pl][ly][lx] = v;
    else sY[pl - 3][ly][lx] = v;
  }
  __syncthreads();
  const float N3 = 3.f * (float)HW;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  const int x = x0 + threadIdx.x, lx = threadIdx.x + LS_HALO;
#pragma unroll 1
  for (int j = 0; j < LS_TY / 4; ++j) {
    const int y = y0 + threadIdx.y + 4 * j, ly = threadIdx.y + 4 * j + LS_HALO;
    if (x < W && y < H) {
      const size_t o = (size_t)y * W + x;
      float g[3] = {0.f, 0.f, 0.f};
      float sq[3], yq[3];
      for (int c = 0; c < 3; ++c) {
        sq[c] = sS[c][ly][lx];
        yq[c] = sY[c][ly][lx];
        float l = L2[c * HW + o], ef = scal[c], ratio = scal[3 + c];
        float ceb = zt_clampf(powf(l * ef, ef) * ratio, 1e-9f, 1.f);
        float d1 = sq[c] - ceb;
        t[0] += d1 * d1 * (700.f / N3);
        g[c] += d1 * (1400.f / N3);
        float q = l / sq[c];
        float nl = zt_clampf(q, 1e-9f, 0.8f), cal = zt_clampf(l * ef, 1e-9f, 1.f);
        float d2 = nl - cal;
        t[1] += d2 * d2 * (1000.f / N3);
        if (q >= 1e-9f && q <= 0.8f) g[c] += d2 * (2000.f / N3) * (-q / sq[c]);
        // total variation (loss.py:139-152), weight 1600
        const float ch = 3200.f / ((float)(H - 1) * (float)W), cw = 3200.f / ((float)H * (float)(W - 1));
        if (y + 1 < H) {
          float d = sS[c][ly + 1][lx] - sq[c];
          t[3] += d * d * ch;
          g[c] -= 2.f * d * ch;
        }
        if (y >= 1) g[c] += 2.f * (sq[c] - sS[c][ly - 1][lx]) * ch;
        if (x + 1 < W) {
          float d = sS[c][ly][lx + 1] - sq[c];
          t[3] += d * d * cw;
          g[c] -= 2.f * d * cw;
        }
        if (x >= 1) g[c] += 2.f * (sq[c] - sS[c][ly][lx - 1]) * cw;
      }
      // bilateral smoothness (loss.py:173-311): 12 offsets, each counted twice, weight 5.  Offsets are compile-time constants
      // (immediate LDS offsets, no index arithmetic) and the per-offset normalisation 10 / ((H - dy)(W - |dx|)) arrives
      // precomputed: the kernel is bound by vector-instruction issue (24 exp + ~150 neighbour operations per pixel), not memory.
      zt_static_for<0, 12>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int dy = k_off_dy[k], dx = k_off_dx[k];
        const float coef = sc.coef[k];
        // pair (q, q+d)
        if (y + dy < H && x + dx >= 0 && x + dx < W) {
          float e = 0.f, a = 0.f;
          float sg[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float dyv = yq[c] - sY[c][ly + dy][lx + dx];
            e += dyv * dyv;
            float ds = sq[c] - sS[c][ly + dy][lx + dx];
            a += fabsf(ds);
            sg[c] = ds > 0.f ? 1.f : (ds < 0.f ? -1.f : 0.f);
          }
          float wgt = (FAST ? __expf(e * -0.005f) : expf(e * -0.005f)) * coef;
          t[2] += wgt * a;
#pragma unroll
          for (int c = 0; c < 3; ++c) g[c] += wgt * sg[c];
        }
        // pair (q-d, q): gradient only (the forward term belongs to pixel q-d)
        if (y - dy >= 0 && x - dx >= 0 && x - dx < W) {
          float e = 0.f;
          float sg[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float dyv = sY[c][ly - dy][lx - dx] - yq[c];
            e += dyv * dyv;
            float ds = sq[c] - sS[c][ly - dy][lx - dx];
            sg[c] = ds > 0.f ? 1.f : (ds < 0.f ? -1.f : 0.f);
          }
          float wgt = (FAST ? __expf(e * -0.005f) : expf(e * -0.005f)) * coef;
#pragma unroll
          for (int c = 0; c < 3; ++c) g[c] += wgt * sg[c];
        }
      });
      for (int c = 0; c < 3; ++c) ds2[c * HW + o] = g[c];
    }
  }
  zt_block_sum<4>(t, red);
  // the caller's partial buffer has one slot per 64 x 4 strip ([ceil(H/4)][ceil(W/64)][4]): this tile fills its first strip's slot
  // and zeroes the other three
  if (tid < 4 * (LS_TY / 4)) {
    const int strip = blockIdx.y * (LS_TY / 4) + (tid >> 2);
    if (strip < nrow4) partial[((size_t)strip * gridDim.x + blockIdx.x) * 4 + (tid & 3)] = 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    float* p = partial + ((size_t)(blockIdx.y * (LS_TY / 4)) * gridDim.x + blockIdx.x) * 4;
    p[0] = t[0]; p[1] = t[1]; p[2] = t[2]; p[3] = t[3];
  }
}

struct HalfArgs {
  const float *Lq11, *Lq12, *Lp1, *Lp2, *den1, *den2, *H3p, *H4p, *H11, *s21, *H12, *s22, *H3d1, *H3d2, *mask, *LM1, *LM2;
  float *dLp1, *dLp2, *dden1, *dden2, *dH3p, *dH4p, *dH3d1, *dH3d2, *u1, *u2, *partial;
  long long hw;
};

// Terms 4-11, 14, 15 (loss.py:51-62, 68-73) on the half-resolution tensors, with direct gradients.
__global__ void __launch_bounds__(256) loss_half_kernel(HalfArgs a) {
  __shared__ float red[16 * 10];
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long hw = a.hw;
  const float N3 = 3.f * (float)hw, N6 = 6.f * (float)hw;
  float t[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) t[k] = 0.f;
  if (i < hw) {
    const float m = a.mask[i];
    for (int c = 0; c < 3; ++c) {
      const long long o = c * hw + i;
      float lp1 = a.Lp1[o], lp2 = a.Lp2[o];
      float ra = a.Lq11[o] - lp2, rb = a.Lq12[o] - lp1, rc = lp1 - a.den1[o], rd = lp2 - a.den2[o];
      t[0] += ra * ra * (1000.f / N3);
      t[1] += rb * rb * (1000.f / N3);
      t[2] += rc * rc * (1000.f / N3);
      t[3] += rd * rd * (1000.f / N3);
      a.dLp2[o] = (-ra + rd) * (2000.f / N3);
      a.dLp1[o] = (-rb + rc) * (2000.f / N3);
      a.dden1[o] = -rc * (2000.f / N3);
      a.dden2[o] = -rd * (2000.f / N3);
      // res2: H3_pred vs cat[H12, s22]; H4_pred vs cat[H11, s21]; first three channels vs pd(H3)
      float p3a = a.H3p[o], p3b = a.H3p[o + 3 * hw], p4a = a.H4p[o], p4b = a.H4p[o + 3 * hw];
      float e3a = p3a - a.H12[o], e3b = p3b - a.s22[o], e4a = p4a - a.H11[o], e4b = p4b - a.s21[o];
      t[4] += (e3a * e3a + e3b * e3b) * (1000.f / N6);
      t[5] += (e4a * e4a + e4b * e4b) * (1000.f / N6);
      float d1 = a.H3d1[o], d2 = a.H3d2[o];
      float f3 = p3a - d1, f4 = p4a - d2;
      t[6] += f3 * f3 * (1000.f / N3);
      t[7] += f4 * f4 * (1000.f / N3);
      a.dH3p[o] = e3a * (2000.f / N6) + f3 * (2000.f / N3);
      a.dH3p[o + 3 * hw] = e3b * (2000.f / N6);
      a.dH4p[o] = e4a * (2000.f / N6) + f4 * (2000.f / N3);
      a.dH4p[o + 3 * hw] = e4b * (2000.f / N6);
      // inter (loss.py:68-73): wd1 = (1-m) LM1 + H3d1 m ; wd2 = (1-m) LM2 + H3d1 m  (sic)
      float e1 = d1 - ((1.f - m) * a.LM1[o] + d1 * m);
      float e2 = d2 - ((1.f - m) * a.LM2[o] + d1 * m);
      t[8] += e1 * e1 * (10000.f / N3);
      t[9] += e2 * e2 * (10000.f / N3);
      float de1 = e1 * (20000.f / N3), de2 = e2 * (20000.f / N3);
      a.dH3d1[o] = -f3 * (2000.f / N3) + de1 * (1.f - m) - m * de2;
      a.dH3d2[o] = -f4 * (2000.f / N3) + de2;
      a.u1[o] = (1.f - m) * de1;
      a.u2[o] = (1.f - m) * de2;
    }
  }
  zt_block_sum<10>(t, red);
  if (threadIdx.x == 0) {
    float* p = a.partial + (size_t)blockIdx.x * 10;
#pragma unroll
    for (int k = 0; k < 10; ++k) p[k] = t[k];
  }
}

// Terms 12, 13, 16 (loss.py:64, 66, 75-77) on full-resolution tensors.
__global__ void __launch_bounds__(256) loss_full_kernel(const float* __restrict__ H2b, const float* __restrict__ H3b,
                                                        const float* __restrict__ s2, const float* __restrict__ s3,
                                                        const float* __restrict__ VH2, const float* __restrict__ VN,
                                                        float* __restrict__ dH3b, float* __restrict__ ds3,
                                                        float* __restrict__ gV, long long n, float* __restrict__ partial) {
  __shared__ float red[16 * 3];
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const float N3 = (float)n;
  float t[3] = {0.f, 0.f, 0.f};
  if (i < n) {
    float a = H2b[i] - H3b[i], b = s2[i] - s3[i], c = VH2[i] - VN[i];
    t[0] = a * a * (10000.f / N3);
    t[1] = b * b * (1000.f / N3);
    t[2] = c * c * (1000.f / N3);
    dH3b[i] = -a * (20000.f / N3);
    ds3[i] = -b * (2000.f / N3);
    gV[i] = c * (2000.f / N3);
  }
  zt_block_sum<3>(t, red);
  if (threadIdx.x == 0) {
    float* p = partial + (size_t)blockIdx.x * 3;
    p[0] = t[0]; p[1] = t[1]; p[2] = t[2];
  }
}

}  // namespace

extern "C" int zt_plane_sums_f32(const float* x, int C, long long HW, int nblk, float* partial, hipStream_t stream) {
  ZT_REQUIRE(x && partial && nblk > 0);
  if (C <= 4 && HW % 4 == 0 && ((uintptr_t)x & 15) == 0) hipLaunchKernelGGL(plane_sums_kernel, dim3(nblk), dim3(256), 0, stream, x, C, HW, nblk, partial);
  else hipLaunchKernelGGL(plane_sums_scalar_kernel, dim3(nblk), dim3(256), 0, stream, x, C, HW, nblk, partial);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_loss_scalars_f32(const float* partial, int nblk, long long HW, int is_WB, float* scal, hipStream_t stream) {
  ZT_REQUIRE(partial && scal);
  hipLaunchKernelGGL(loss_scalars_kernel, dim3(1), dim3(64), 0, stream, partial, nblk, (double)HW, is_WB, scal);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_loss_s2_f32(const float* L2, const float* s2, const float* Y, const float* scal, int H, int W, float* ds2,
                              float* partial, int fast_exp, hipStream_t stream) {
  ZT_REQUIRE(L2 && s2 && Y && scal && ds2 && partial && H > 2 && W > 2);
  SmoothCoef sc;
  for (int k = 0; k < 12; ++k) {
    const int adx = k_off_dx[k] < 0 ? -k_off_dx[k] : k_off_dx[k];
    sc.coef[k] = 10.f / ((float)(H - k_off_dy[k]) * (float)(W - adx));      // same fp32 expression the kernel used to evaluate per pixel
  }
  const dim3 grid(zt_cdiv(W, LS_TX), zt_cdiv(H, LS_TY));
  if (fast_exp) hipLaunchKernelGGL(loss_s2_kernel<true>, grid, dim3(64, 4), 0, stream, L2, s2, Y, scal, H, W, ds2, partial, zt_cdiv(H, 4), sc);
  else hipLaunchKernelGGL(loss_s2_kernel<false>, grid, dim3(64, 4), 0, stream, L2, s2, Y, scal, H, W, ds2, partial, zt_cdiv(H, 4), sc);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_loss_half_f32(const float* Lq11, const float* Lq12, const float* Lp1, const float* Lp2, const float* den1,
                                const float* den2, const float* H3p, const float* H4p, const float* H11, const float* s21,
                                const float* H12, const float* s22, const float* H3d1, const float* H3d2, const float* mask,
                                const float* LM1, const float* LM2, float* dLp1, float* dLp2, float* dden1, float* dden2,
                                float* dH3p, float* dH4p, float* dH3d1, float* dH3d2, float* u1, float* u2, long long hw,
                                float* partial, hipStream_t stream) {
  HalfArgs a;
  a.Lq11 = Lq11; a.Lq12 = Lq12; a.Lp1 = Lp1; a.Lp2 = Lp2; a.den1 = den1; a.den2 = den2; a.H3p = H3p; a.H4p = H4p;
  a.H11 = H11; a.s21 = s21; a.H12 = H12; a.s22 = s22; a.H3d1 = H3d1; a.H3d2 = H3d2; a.mask = mask; a.LM1 = LM1; a.LM2 = LM2;
  a.dLp1 = dLp1; a.dLp2 = dLp2; a.dden1 = dden1; a.dden2 = dden2; a.dH3p = dH3p; a.dH4p = dH4p; a.dH3d1 = dH3d1;
  a.dH3d2 = dH3d2; a.u1 = u1; a.u2 = u2; a.partial = partial; a.hw = hw;
  ZT_REQUIRE(Lq11 && Lq12 && Lp1 && Lp2 && den1 && den2 && H3p && H4p && H11 && s21 && H12 && s22 && H3d1 && H3d2 && mask && LM1 && LM2);
  ZT_REQUIRE(dLp1 && dLp2 && dden1 && dden2 && dH3p && dH4p && dH3d1 && dH3d2 && u1 && u2 && partial && hw > 0);
  hipLaunchKernelGGL(loss_half_kernel, dim3((unsigned)zt_cdivl(hw, 256)), dim3(256), 0, stream, a);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_loss_full_f32(const float* H2b, const float* H3b, const float* s2, const float* s3, const float* VH2,
                                const float* VN, float* dH3b, float* ds3, float* gV, long long n, float* partial,
                                hipStream_t stream) {
  ZT_REQUIRE(H2b && H3b && s2 && s3 && VH2 && VN && dH3b && ds3 && gV && partial && n > 0);
  hipLaunchKernelGGL(loss_full_kernel, dim3((unsigned)zt_cdivl(n, 256)), dim3(256), 0, stream, H2b, H3b, s2, s3, VH2, VN,
                     dH3b, ds3, gV, n, partial);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
