// Element-wise stages of Network.forward (reference model/model.py:144-203) and their hand-derived backward,
// fused so that every full-resolution 3-channel plane is read / written once per stage.  Planar fp32 [3][H][W],
// H and W even (the reference always feeds 1920x1080).  Half resolution: h = H/2, w = W/2.
#include "zt_common.h"

#define ZT_EPS 1e-4f

namespace {

__device__ __forceinline__ bool in_clamp(float v, float lo, float hi) { return v >= lo && v <= hi; }

// model.py:145-148 + loss.py:25,51: x = inp + 1e-4; (L11,L12) = pd(x); (Lq11,Lq12) = pd(inp + 1e-9)
__global__ void __launch_bounds__(256) prep_input_kernel(const float* __restrict__ inp, float* __restrict__ x,
                                                         float* __restrict__ L11, float* __restrict__ L12,
                                                         float* __restrict__ Lq11, float* __restrict__ Lq12, int H, int W) {
  const int h = H >> 1, w = W >> 1;
  int hx = blockIdx.x * 64 + threadIdx.x, hy = blockIdx.y * 4 + threadIdx.y;
  if (hx >= w || hy >= h) return;
  for (int c = 0; c < 3; ++c) {
    size_t o = (size_t)c * H * W + (size_t)(2 * hy) * W + 2 * hx;
    float a = inp[o], b = inp[o + 1], cc = inp[o + W], d = inp[o + W + 1];
    float xa = a + ZT_EPS, xb = b + ZT_EPS, xc = cc + ZT_EPS, xd = d + ZT_EPS;
    x[o] = xa; x[o + 1] = xb; x[o + W] = xc; x[o + W + 1] = xd;
    size_t ho = (size_t)c * h * w + (size_t)hy * w + hx;
    L11[ho] = 0.5f * xb + 0.5f * xc;
    L12[ho] = 0.5f * xa + 0.5f * xd;
    float qa = a + 1e-9f, qb = b + 1e-9f, qc = cc + 1e-9f, qd = d + 1e-9f;
    Lq11[ho] = 0.5f * qb + 0.5f * qc;
    Lq12[ho] = 0.5f * qa + 0.5f * qd;
  }
}

struct PackSrc {
  const float* p[4];
  int c[4];
};

// dst[pix][0..ld): planar sources concatenated channel-wise, remaining channels zero
template <typename T>
__global__ void __launch_bounds__(256) pack_nhwc_kernel(T* __restrict__ dst, int ld, long long HW, PackSrc s) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  T* d = dst + i * ld;
  int k = 0;
  for (int j = 0; j < 4; ++j)
    for (int c = 0; c < s.c[j]; ++c) ZtIO<T>::st(d + k++, s.p[j][(size_t)c * HW + i]);
  for (; k < ld; ++k) ZtIO<T>::st(d + k, 0.f);
}

// bf16 fast path (LD = 8 or 16 channels per pixel: every thin network input): all channel planes of the pixel are loaded first
// (coalesced, independent), packed to bf16 and written with one or two 16-byte stores (the generic kernel issues LD 2-byte
// stores per pixel and ran at 1.8 TB/s).
struct PackPlanes {
  const float* ch[16];
  int nch;
};

template <int LD>
__global__ void __launch_bounds__(256) pack_nhwc_bf16_kernel(zt_bf16* __restrict__ dst, long long HW, PackPlanes s) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  float v[LD];
#pragma unroll
  for (int k = 0; k < LD; ++k) v[k] = k < s.nch ? s.ch[k][i] : 0.f;          // uniform predicate
  unsigned w[LD / 2];
#pragma unroll
  for (int k = 0; k < LD / 2; ++k) w[k] = zt_f2bf2(v[2 * k], v[2 * k + 1]);
  uint4* d = reinterpret_cast<uint4*>(dst + i * LD);
#pragma unroll
  for (int k = 0; k < LD / 8; ++k) d[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

// model.py:149-152 (+ loss.py:54): L2 = clamp(x - n); L_pred1/2 = L11/12 - n11/12; (den1, den2) = pd(L2)
__global__ void __launch_bounds__(256) d1_tail_kernel(const float* __restrict__ x, const float* __restrict__ n,
                                                      const float* __restrict__ L11, const float* __restrict__ n11,
                                                      const float* __restrict__ L12, const float* __restrict__ n12,
                                                      float* __restrict__ L2, float* __restrict__ Lp1, float* __restrict__ Lp2,
                                                      float* __restrict__ den1, float* __restrict__ den2, int H, int W) {
  const int h = H >> 1, w = W >> 1;
  int hx = blockIdx.x * 64 + threadIdx.x, hy = blockIdx.y * 4 + threadIdx.y;
  if (hx >= w || hy >= h) return;
  for (int c = 0; c < 3; ++c) {
    size_t o = (size_t)c * H * W + (size_t)(2 * hy) * W + 2 * hx;
    float a = zt_clampf(x[o] - n[o], ZT_EPS, 1.f), b = zt_clampf(x[o + 1] - n[o + 1], ZT_EPS, 1.f);
    float cc = zt_clampf(x[o + W] - n[o + W], ZT_EPS, 1.f), d = zt_clampf(x[o + W + 1] - n[o + W + 1], ZT_EPS, 1.f);
    L2[o] = a; L2[o + 1] = b; L2[o + W] = cc; L2[o + W + 1] = d;
    size_t ho = (size_t)c * h * w + (size_t)hy * w + hx;
    den1[ho] = 0.5f * b + 0.5f * cc;
    den2[ho] = 0.5f * a + 0.5f * d;
    Lp1[ho] = L11[ho] - n11[ho];
    Lp2[ho] = L12[ho] - n12[ho];
  }
}

// model.py:169-177,198-199: (s21,s22) = pd(s2); H2 = clamp(x/s2); H11 = clamp(L11/s21); H12 = clamp(L12/s22); H1 = clamp(L2/s2,0,1)
__global__ void __launch_bounds__(256) post_enh_kernel(const float* __restrict__ x, const float* __restrict__ s2,
                                                       const float* __restrict__ L2, const float* __restrict__ L11,
                                                       const float* __restrict__ L12, float* __restrict__ s21,
                                                       float* __restrict__ s22, float* __restrict__ H2, float* __restrict__ H11,
                                                       float* __restrict__ H12, float* __restrict__ H1, int H, int W) {
  const int h = H >> 1, w = W >> 1;
  int hx = blockIdx.x * 64 + threadIdx.x, hy = blockIdx.y * 4 + threadIdx.y;
  if (hx >= w || hy >= h) return;
  for (int c = 0; c < 3; ++c) {
    size_t o = (size_t)c * H * W + (size_t)(2 * hy) * W + 2 * hx;
    const size_t offs[4] = {o, o + 1, o + (size_t)W, o + (size_t)W + 1};
    float sv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = s2[offs[k]];
      sv[k] = s;
      H2[offs[k]] = zt_clampf(x[offs[k]] / s, ZT_EPS, 1.f);
      H1[offs[k]] = zt_clampf(L2[offs[k]] / s, 0.f, 1.f);
    }
    size_t ho = (size_t)c * h * w + (size_t)hy * w + hx;
    float a21 = 0.5f * sv[1] + 0.5f * sv[2], a22 = 0.5f * sv[0] + 0.5f * sv[3];
    s21[ho] = a21;
    s22[ho] = a22;
    H11[ho] = zt_clampf(L11[ho] / a21, ZT_EPS, 1.f);
    H12[ho] = zt_clampf(L12[ho] / a22, ZT_EPS, 1.f);
  }
}

// model.py:179-192: out6 = clamp(cat[A,B] - r, 1e-4, 1) for a 6-channel planar residual r; A,B 3-channel planar.
// outA/outB receive channels 0-2 / 3-5 (they may alias one contiguous 6-channel buffer).
__global__ void __launch_bounds__(256) clamp_sub6_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         const float* __restrict__ r, float* __restrict__ outA,
                                                         float* __restrict__ outB, long long HW) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  for (int c = 0; c < 3; ++c) {
    outA[c * HW + i] = zt_clampf(A[c * HW + i] - r[c * HW + i], ZT_EPS, 1.f);
    outB[c * HW + i] = zt_clampf(B[c * HW + i] - r[(c + 3) * HW + i], ZT_EPS, 1.f);
  }
}

// backward of the above into the NHWC (ld) gradient of the residual r: dr[c] = -[1e-4 <= skip_c - r_c <= 1] * g_c
template <typename T>
__global__ void __launch_bounds__(256) clamp_sub6_bwd_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                             const float* __restrict__ r, const float* __restrict__ gA,
                                                             const float* __restrict__ gB, T* __restrict__ dr, int ld,
                                                             long long HW) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  T* d = dr + i * ld;
  for (int c = 0; c < 3; ++c) {
    float pa = A[c * HW + i] - r[c * HW + i], pb = B[c * HW + i] - r[(c + 3) * HW + i];
    ZtIO<T>::st(d + c, in_clamp(pa, ZT_EPS, 1.f) ? -gA[c * HW + i] : 0.f);
    ZtIO<T>::st(d + c + 3, in_clamp(pb, ZT_EPS, 1.f) ? -gB[c * HW + i] : 0.f);
  }
  for (int k = 6; k < ld; ++k) ZtIO<T>::st(d + k, 0.f);
}

// Backward through H2 = clamp(x/s2), H11/H12 = clamp(L1x/s2x), (s21,s22) = pd(s2) and the Enhancer's
// clamp(sigmoid) (model.py:79, 169-177): collects every gradient that reaches s2 and emits the Enhancer's
// output-layer gradient  dO = ds2 * s2 (1 - s2) [s2 > 1e-4]  as NHWC (ld).
// dIn5: planar [12][H][W] gradient of the full-res Denoise_2 input (ch 6-8 = dH2, 9-11 = ds2); dIn3/dIn4: half-res twins.
template <typename T>
__global__ void __launch_bounds__(256) post_enh_bwd_kernel(const float* __restrict__ x, const float* __restrict__ s2,
                                                           const float* __restrict__ L11, const float* __restrict__ L12,
                                                           const float* __restrict__ s21, const float* __restrict__ s22,
                                                           const float* __restrict__ dIn5, const float* __restrict__ dH2x,
                                                           const float* __restrict__ dIn3, const float* __restrict__ dIn4,
                                                           const float* __restrict__ ds2_direct, T* __restrict__ dO,
                                                           int ld, float* __restrict__ ds2_total, int H, int W) {
  const int h = H >> 1, w = W >> 1;
  int hx = blockIdx.x * 64 + threadIdx.x, hy = blockIdx.y * 4 + threadIdx.y;
  if (hx >= w || hy >= h) return;
  const size_t HW = (size_t)H * W, hw = (size_t)h * w;
  for (int c = 0; c < 3; ++c) {
    size_t ho = (size_t)c * hw + (size_t)hy * w + hx;
    size_t hp = (size_t)hy * w + hx;
    float a21 = s21[ho], a22 = s22[ho];
    float q1 = L11[ho] / a21, q2 = L12[ho] / a22;
    float g21 = dIn3[(9 + c) * hw + hp] + (in_clamp(q1, ZT_EPS, 1.f) ? dIn3[(6 + c) * hw + hp] * (-q1 / a21) : 0.f);
    float g22 = dIn4[(9 + c) * hw + hp] + (in_clamp(q2, ZT_EPS, 1.f) ? dIn4[(6 + c) * hw + hp] * (-q2 / a22) : 0.f);
    size_t o = (size_t)c * HW + (size_t)(2 * hy) * W + 2 * hx;
    size_t p = (size_t)(2 * hy) * W + 2 * hx;
    const size_t offs[4] = {0, 1, (size_t)W, (size_t)W + 1};
    const float pdadj[4] = {0.5f * g22, 0.5f * g21, 0.5f * g21, 0.5f * g22};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      size_t oo = o + offs[k], pp = p + offs[k];
      float s = s2[oo];
      float q = x[oo] / s;
      float gH2 = dIn5[(6 + c) * HW + pp] + dH2x[oo];
      float g = ds2_direct[oo] + dIn5[(9 + c) * HW + pp] + pdadj[k] + (in_clamp(q, ZT_EPS, 1.f) ? gH2 * (-q / s) : 0.f);
      if (ds2_total) ds2_total[oo] = g;
      ZtIO<T>::st(dO + pp * ld + c, (s > ZT_EPS) ? g * s * (1.f - s) : 0.f);
    }
  }
  if (ld > 3) {
    size_t p = (size_t)(2 * hy) * W + 2 * hx;
    const size_t offs[4] = {0, 1, (size_t)W, (size_t)W + 1};
    for (int k = 0; k < 4; ++k)
      for (int j = 3; j < ld; ++j) ZtIO<T>::st(dO + (p + offs[k]) * ld + j, 0.f);
  }
}

// Gradients entering the three Denoise_1 invocations (model.py:149-152): dn11 = -dLp1, dn12 = -dLp2 (half res),
// dn = -[1e-4 <= x - n <= 1] * pd^T(dden1, dden2) (full res); all NHWC (ld).
template <typename T>
__global__ void __launch_bounds__(256) d1_bwd_prep_kernel(const float* __restrict__ x, const float* __restrict__ n,
                                                          const float* __restrict__ dLp1, const float* __restrict__ dLp2,
                                                          const float* __restrict__ dden1, const float* __restrict__ dden2,
                                                          T* __restrict__ dn, T* __restrict__ dn11,
                                                          T* __restrict__ dn12, int ld, int H, int W) {
  const int h = H >> 1, w = W >> 1;
  int hx = blockIdx.x * 64 + threadIdx.x, hy = blockIdx.y * 4 + threadIdx.y;
  if (hx >= w || hy >= h) return;
  const size_t HW = (size_t)H * W, hw = (size_t)h * w;
  size_t hp = (size_t)hy * w + hx;
  size_t p = (size_t)(2 * hy) * W + 2 * hx;
  const size_t offs[4] = {0, 1, (size_t)W, (size_t)W + 1};
  for (int c = 0; c < 3; ++c) {
    float g1 = dden1[c * hw + hp], g2 = dden2[c * hw + hp];
    const float adj[4] = {0.5f * g2, 0.5f * g1, 0.5f * g1, 0.5f * g2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      size_t oo = c * HW + p + offs[k];
      float pre = x[oo] - n[oo];
      ZtIO<T>::st(dn + (p + offs[k]) * ld + c, in_clamp(pre, ZT_EPS, 1.f) ? -adj[k] : 0.f);
    }
    ZtIO<T>::st(dn11 + hp * ld + c, -dLp1[c * hw + hp]);
    ZtIO<T>::st(dn12 + hp * ld + c, -dLp2[c * hw + hp]);
  }
  for (int j = 3; j < ld; ++j) {
    for (int k = 0; k < 4; ++k) ZtIO<T>::st(dn + (p + offs[k]) * ld + j, 0.f);
    ZtIO<T>::st(dn11 + hp * ld + j, 0.f);
    ZtIO<T>::st(dn12 + hp * ld + j, 0.f);
  }
}

// out = a + b (+ c)  element-wise on flat fp32 buffers
__global__ void __launch_bounds__(256) add3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                   const float* __restrict__ c, float* __restrict__ out, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = a[i] + b[i];
  if (c) v += c[i];
  out[i] = v;
}

// out = g * [a > 0]  (ReLU backward) on NHWC buffers
template <typename T>
__global__ void __launch_bounds__(256) relu_mask_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ a,
                                                        int lda, T* __restrict__ out, int ldo, int C, long long total4) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int Q = C >> 2;
  int q = (int)(i % Q);
  long long p = i / Q;
  float4 gv = ZtIO<T>::ld4(g + p * ldg + q * 4);
  float4 av = ZtIO<T>::ld4(a + p * lda + q * 4);
  gv.x = av.x > 0.f ? gv.x : 0.f; gv.y = av.y > 0.f ? gv.y : 0.f; gv.z = av.z > 0.f ? gv.z : 0.f; gv.w = av.w > 0.f ? gv.w : 0.f;
  ZtIO<T>::st4(out + p * ldo + q * 4, gv);
}

// mode 0: a + p0   1: clamp(a - b, p0, p1)   2: clamp(a / b, p0, p1)
__global__ void __launch_bounds__(256) ew_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                 int mode, float p0, float p1, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = a[i];
  if (mode == 0) v = v + p0;
  else if (mode == 1) v = zt_clampf(v - b[i], p0, p1);
  else v = zt_clampf(v / b[i], p0, p1);
  out[i] = v;
}

// out = alpha * a + beta * b
__global__ void __launch_bounds__(256) axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                    float alpha, float beta, long long n) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = alpha * a[i] + beta * b[i];
}

inline dim3 grid_half(int H, int W) { return dim3(zt_cdiv(W / 2, 64), zt_cdiv(H / 2, 4)); }

}  // namespace

extern "C" int zt_prep_input_f32(const float* inp, float* x, float* L11, float* L12, float* Lq11, float* Lq12, int H, int W,
                                 hipStream_t stream) {
  ZT_REQUIRE(inp && x && L11 && L12 && Lq11 && Lq12 && H % 2 == 0 && W % 2 == 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(prep_input_kernel, grid_half(H, W), dim3(64, 4), 0, stream, inp, x, L11, L12, Lq11, Lq12, H, W);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_pack_nhwc(void* dst, int dt, int ld, long long HW, const float* s0, int c0, const float* s1, int c1,
                                const float* s2, int c2, const float* s3, int c3, hipStream_t stream) {
  ZT_REQUIRE(dst && c0 + c1 + c2 + c3 <= ld && c0 >= 0 && c1 >= 0 && c2 >= 0 && c3 >= 0);
  ZT_REQUIRE((c0 == 0 || s0) && (c1 == 0 || s1) && (c2 == 0 || s2) && (c3 == 0 || s3));
  PackSrc s;
  s.p[0] = s0; s.p[1] = s1; s.p[2] = s2; s.p[3] = s3;
  s.c[0] = c0; s.c[1] = c1; s.c[2] = c2; s.c[3] = c3;
  if (dt != 0 && (ld == 8 || ld == 16) && ((uintptr_t)dst & 15) == 0) {
    PackPlanes pl;
    pl.nch = 0;
    for (int j = 0; j < 4; ++j)
      for (int c = 0; c < s.c[j]; ++c) pl.ch[pl.nch++] = s.p[j] + (size_t)c * HW;
    for (int k = pl.nch; k < 16; ++k) pl.ch[k] = nullptr;
    const dim3 grid((unsigned)zt_cdivl(HW, 256));
    if (ld == 8) hipLaunchKernelGGL(pack_nhwc_bf16_kernel<8>, grid, dim3(256), 0, stream, (zt_bf16*)dst, HW, pl);
    else hipLaunchKernelGGL(pack_nhwc_bf16_kernel<16>, grid, dim3(256), 0, stream, (zt_bf16*)dst, HW, pl);
    ZT_LAUNCH_CHECK();
    return ZT_OK;
  }
  if (dt == 0) hipLaunchKernelGGL(pack_nhwc_kernel<float>, dim3((unsigned)zt_cdivl(HW, 256)), dim3(256), 0, stream, (float*)dst, ld, HW, s);
  else hipLaunchKernelGGL(pack_nhwc_kernel<zt_bf16>, dim3((unsigned)zt_cdivl(HW, 256)), dim3(256), 0, stream, (zt_bf16*)dst, ld, HW, s);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_d1_tail_f32(const float* x, const float* n, const float* L11, const float* n11, const float* L12,
                              const float* n12, float* L2, float* Lp1, float* Lp2, float* den1, float* den2, int H, int W,
                              hipStream_t stream) {
  ZT_REQUIRE(x && n && L11 && n11 && L12 && n12 && L2 && Lp1 && Lp2 && den1 && den2 && H % 2 == 0 && W % 2 == 0);
  hipLaunchKernelGGL(d1_tail_kernel, grid_half(H, W), dim3(64, 4), 0, stream, x, n, L11, n11, L12, n12, L2, Lp1, Lp2, den1,
                     den2, H, W);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_post_enh_f32(const float* x, const float* s2, const float* L2, const float* L11, const float* L12,
                               float* s21, float* s22, float* H2, float* H11, float* H12, float* H1, int H, int W,
                               hipStream_t stream) {
  ZT_REQUIRE(x && s2 && L2 && L11 && L12 && s21 && s22 && H2 && H11 && H12 && H1 && H % 2 == 0 && W % 2 == 0);
  hipLaunchKernelGGL(post_enh_kernel, grid_half(H, W), dim3(64, 4), 0, stream, x, s2, L2, L11, L12, s21, s22, H2, H11, H12,
                     H1, H, W);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_clamp_sub6_f32(const float* A, const float* B, const float* r, float* outA, float* outB, long long HW,
                                 hipStream_t stream) {
  ZT_REQUIRE(A && B && r && outA && outB);
  hipLaunchKernelGGL(clamp_sub6_kernel, dim3((unsigned)zt_cdivl(HW, 256)), dim3(256), 0, stream, A, B, r, outA, outB, HW);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_clamp_sub6_bwd(const float* A, const float* B, const float* r, const float* gA, const float* gB,
                                 void* dr, int dt, int ld, long long HW, hipStream_t stream) {
  ZT_REQUIRE(A && B && r && gA && gB && dr && ld >= 6);
  if (dt == 0) hipLaunchKernelGGL(clamp_sub6_bwd_kernel<float>, dim3((unsigned)zt_cdivl(HW, 256)), dim3(256), 0, stream, A, B, r, gA, gB, (float*)dr, ld, HW);
  else hipLaunchKernelGGL(clamp_sub6_bwd_kernel<zt_bf16>, dim3((unsigned)zt_cdivl(HW, 256)), dim3(256), 0, stream, A, B, r, gA, gB, (zt_bf16*)dr, ld, HW);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_post_enh_bwd(const float* x, const float* s2, const float* L11, const float* L12, const float* s21,
                               const float* s22, const float* dIn5, const float* dH2x, const float* dIn3,
                               const float* dIn4, const float* ds2_direct, void* dO, int dt, int ld, float* ds2_total, int H,
                               int W, hipStream_t stream) {
  ZT_REQUIRE(x && s2 && L11 && L12 && s21 && s22 && dIn5 && dH2x && dIn3 && dIn4 && ds2_direct && dO && ld >= 3);
  ZT_REQUIRE(H % 2 == 0 && W % 2 == 0);
  if (dt == 0)
    hipLaunchKernelGGL(post_enh_bwd_kernel<float>, grid_half(H, W), dim3(64, 4), 0, stream, x, s2, L11, L12, s21, s22, dIn5, dH2x,
                       dIn3, dIn4, ds2_direct, (float*)dO, ld, ds2_total, H, W);
  else
    hipLaunchKernelGGL(post_enh_bwd_kernel<zt_bf16>, grid_half(H, W), dim3(64, 4), 0, stream, x, s2, L11, L12, s21, s22, dIn5, dH2x,
                       dIn3, dIn4, ds2_direct, (zt_bf16*)dO, ld, ds2_total, H, W);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_d1_bwd_prep(const float* x, const float* n, const float* dLp1, const float* dLp2, const float* dden1,
                              const float* dden2, void* dn, void* dn11, void* dn12, int dt, int ld, int H, int W,
                              hipStream_t stream) {
  ZT_REQUIRE(x && n && dLp1 && dLp2 && dden1 && dden2 && dn && dn11 && dn12 && ld >= 3 && H % 2 == 0 && W % 2 == 0);
  if (dt == 0)
    hipLaunchKernelGGL(d1_bwd_prep_kernel<float>, grid_half(H, W), dim3(64, 4), 0, stream, x, n, dLp1, dLp2, dden1, dden2, (float*)dn,
                       (float*)dn11, (float*)dn12, ld, H, W);
  else
    hipLaunchKernelGGL(d1_bwd_prep_kernel<zt_bf16>, grid_half(H, W), dim3(64, 4), 0, stream, x, n, dLp1, dLp2, dden1, dden2,
                       (zt_bf16*)dn, (zt_bf16*)dn11, (zt_bf16*)dn12, ld, H, W);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_add3_f32(const float* a, const float* b, const float* c, float* out, long long n, hipStream_t stream) {
  ZT_REQUIRE(a && b && out);
  hipLaunchKernelGGL(add3_kernel, dim3((unsigned)zt_cdivl(n, 256)), dim3(256), 0, stream, a, b, c, out, n);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_relu_mask_nhwc(const void* g, int dt, int ldg, const void* a, int lda, void* out, int ldo, long long npix, int C,
                                 hipStream_t stream) {
  ZT_REQUIRE(g && a && out && C % 4 == 0 && ldg % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0);
  long long total4 = npix * (C / 4);
  if (dt == 0)
    hipLaunchKernelGGL(relu_mask_kernel<float>, dim3((unsigned)zt_cdivl(total4, 256)), dim3(256), 0, stream, (const float*)g, ldg,
                       (const float*)a, lda, (float*)out, ldo, C, total4);
  else
    hipLaunchKernelGGL(relu_mask_kernel<zt_bf16>, dim3((unsigned)zt_cdivl(total4, 256)), dim3(256), 0, stream, (const zt_bf16*)g, ldg,
                       (const zt_bf16*)a, lda, (zt_bf16*)out, ldo, C, total4);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_ew_f32(const float* a, const float* b, float* out, int mode, float p0, float p1, long long n, hipStream_t stream) {
  ZT_REQUIRE(a && out && (mode == 0 || b) && mode >= 0 && mode <= 2);
  hipLaunchKernelGGL(ew_kernel, dim3((unsigned)zt_cdivl(n, 256)), dim3(256), 0, stream, a, b, out, mode, p0, p1, n);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_axpby_f32(const float* a, const float* b, float* out, float alpha, float beta, long long n, hipStream_t stream) {
  ZT_REQUIRE(a && b && out);
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)zt_cdivl(n, 256)), dim3(256), 0, stream, a, b, out, alpha, beta, n);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
