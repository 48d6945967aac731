// Per-channel statistics + normalisation on NHWC fp32 activations:
//   train-mode BatchNorm of the Enhancer's shared block (reference model.py:60-67; torch BatchNorm2d semantics:
//   biased batch variance for normalisation, unbiased for running_var, momentum 0.1) incl. its backward,
//   eval-mode BatchNorm of RAFT's context encoder and InstanceNorm of its feature encoder (extractor.py:117-191).
// Two-stage reductions (per-workgroup fp32 partials, fp64 combine) -> deterministic, no atomics.
#include "zt_common.h"

namespace {

// InstanceNorm in one launch: tickets != nullptr makes the last workgroup of sample n to finish fold the partial rows into
// scale = rstd, shift = -mean * rstd (what norm_finalize_kernel computes in mode 0), saving the finalize launch on RAFT's feature
// encoder (15 norm layers per call).  tickets[n] counts finished workgroups and is left at zero again.
struct NormTail {
  unsigned* tickets;
  float* scale;
  float* shift;
  double count;
  float eps;
};

__device__ __forceinline__ void instnorm_tail(const NormTail& t, const float* partial, int nblk, int C, int n) {
  __shared__ unsigned last_s;
  __shared__ double tail_s[256];
  __shared__ double tail_q[256];
  __threadfence();                                              // this workgroup's partial row is visible device-wide ...
  __syncthreads();
  if (threadIdx.x == 0) last_s = atomicAdd(t.tickets + n, 1u) == (unsigned)(nblk - 1);      // ... before its ticket is
  __syncthreads();
  if (!last_s) return;
  __threadfence();                                              // acquire: the other workgroups' rows
  const int Cb = C < 256 ? C : 256, G = 256 / Cb;
  const int cl = threadIdx.x % Cb, grp = threadIdx.x / Cb;
  for (int cbase = 0; cbase < C; cbase += Cb) {
    const int c = cbase + cl;
    double s = 0.0, q = 0.0;
    if (grp < G && c < C) {
      const float* pc = partial + (size_t)n * nblk * 2 * C + c;
      for (int b = grp; b < nblk; b += 8 * G) {                 // eight rows in flight, clamped loads masked afterwards
        float ps[8], pq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int bb = b + j * G;
          const unsigned off = (unsigned)(bb < nblk ? bb : b) * 2u * (unsigned)C;
          ps[j] = pc[off];
          pq[j] = pc[off + C];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = b + j * G < nblk;
          s += ok ? (double)ps[j] : 0.0;
          q += ok ? (double)pq[j] : 0.0;
        }
      }
    }
    tail_s[threadIdx.x] = s;
    tail_q[threadIdx.x] = q;
    __syncthreads();
    if (grp == 0 && c < C) {
      for (int k = 1; k < G; ++k) {
        s += tail_s[k * Cb + cl];
        q += tail_q[k * Cb + cl];
      }
      const double m = s / t.count;
      double var = q / t.count - m * m;
      if (var < 0.0) var = 0.0;
      const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)t.eps));
      t.scale[n * C + c] = rstd;
      t.shift[n * C + c] = 0.f - mean * rstd;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) t.tickets[n] = 0u;
}

// partial[((n*nblk + blk)*2 + {0:sum,1:sumsq})*C + c]
template <typename T>
__global__ void __launch_bounds__(256) chan_stats_kernel(const T* __restrict__ x, int ldx, int HW, int C, int nblk,
                                                         float* __restrict__ partial, NormTail tail) {
  __shared__ float4 sh_s[256];
  __shared__ float4 sh_q[256];
  const int Q = C >> 2, R = 256 / Q;
  const int tid = threadIdx.x;
  const int row = tid / Q, q = tid - row * Q;
  const int n = blockIdx.y, blk = blockIdx.x;
  const int chunk = (HW + nblk - 1) / nblk;
  const int p0 = blk * chunk, p1 = min(HW, p0 + chunk);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
  if (row < R) {
    const T* base = x + (size_t)n * HW * ldx + q * 4;
    for (int p = p0 + row; p < p1; p += R) {
      float4 v = ZtIO<T>::ld4(base + (size_t)p * ldx);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      ss.x += v.x * v.x; ss.y += v.y * v.y; ss.z += v.z * v.z; ss.w += v.w * v.w;
    }
  }
  sh_s[tid] = s;
  sh_q[tid] = ss;
  __syncthreads();
  if (row == 0) {
    for (int r = 1; r < R; ++r) {
      float4 a = sh_s[r * Q + q], b = sh_q[r * Q + q];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      ss.x += b.x; ss.y += b.y; ss.z += b.z; ss.w += b.w;
    }
    float* o = partial + ((size_t)(n * nblk + blk) * 2) * C + q * 4;
    *reinterpret_cast<float4*>(o) = s;
    *reinterpret_cast<float4*>(o + C) = ss;
  }
  if (tail.tickets) instnorm_tail(tail, partial, nblk, C, n);
}

// mode 0: instance norm (per n,c; no affine)   mode 1: train BN (N must be 1; updates running stats)   mode 2: eval BN
// 256 threads = G groups x C channels (G = 256 / C): each group sums a strided share of the partials, fp64 combine in LDS.
__global__ void __launch_bounds__(1024) norm_finalize_kernel(const float* __restrict__ partial, int nblk, int C,
                                                            double count, float eps, int mode,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var,
                                                            long long* __restrict__ nbt, float momentum,
                                                            float* __restrict__ scale, float* __restrict__ shift,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  __shared__ double sh_s[1024];
  __shared__ double sh_q[1024];
  const int n = blockIdx.x;
  const int Cb = C < 256 ? C : 256;                 // channels handled per pass
  const int G = 1024 / Cb;                          // partial-row groups: 1024 threads keep the serial chain short
  for (int cbase = 0; cbase < C; cbase += Cb) {
    const int cl = threadIdx.x % Cb, grp = threadIdx.x / Cb;
    const int c = cbase + cl;
    double s = 0.0, q = 0.0;
    if (mode != 2 && grp < G && c < C) {
      const float* pc = partial + (size_t)n * nblk * 2 * C + c;
      for (int b = grp; b < nblk; b += 8 * G) {              // eight partial rows in flight; same summation order as a plain loop
        float ps[8], pq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                          // unconditional loads (clamped row), masked afterwards
          const int bb = b + j * G;
          const unsigned off = (unsigned)(bb < nblk ? bb : b) * 2u * (unsigned)C;
          ps[j] = pc[off];
          pq[j] = pc[off + C];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = b + j * G < nblk;
          s += ok ? (double)ps[j] : 0.0;
          q += ok ? (double)pq[j] : 0.0;
        }
      }
    }
    sh_s[threadIdx.x] = s;
    sh_q[threadIdx.x] = q;
    __syncthreads();
    if (grp == 0 && c < C) {
      float mean, rstd;
      if (mode == 2) {
        mean = running_mean[c];
        rstd = 1.f / sqrtf(running_var[c] + eps);
      } else {
        for (int k = 1; k < G; ++k) {
          s += sh_s[k * Cb + cl];
          q += sh_q[k * Cb + cl];
        }
        double m = s / count;
        double var = q / count - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)eps));
        if (mode == 1) {
          running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
          float unbiased = (float)(var * count / (count - 1.0));
          running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
      }
      float g = (mode == 0) ? 1.f : gamma[c];
      float b = (mode == 0) ? 0.f : beta[c];
      float sc = g * rstd;
      scale[n * C + c] = sc;
      shift[n * C + c] = b - mean * sc;
      if (mean_out) mean_out[n * C + c] = mean;
      if (rstd_out) rstd_out[n * C + c] = rstd;
    }
    __syncthreads();
  }
  if (mode == 1 && nbt && threadIdx.x == 0 && blockIdx.x == 0) *nbt += 1;
}

// y = [outer_relu]( [res +] [inner_relu](x*scale + shift) )
template <typename T>
__global__ void __launch_bounds__(256) norm_apply_kernel(const T* __restrict__ x, int ldx,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const T* __restrict__ res, int ldres, T* __restrict__ y,
                                                         int ldy, int HW, int C, int inner_relu, int outer_relu,
                                                         long long total4) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int Q = C >> 2;
  int q = (int)(i % Q);
  long long p = i / Q;            // global pixel index over N*HW
  int n = (int)(p / HW);
  float4 v = ZtIO<T>::ld4(x + p * ldx + q * 4);
  float4 sc = *reinterpret_cast<const float4*>(scale + n * C + q * 4);
  float4 sf = *reinterpret_cast<const float4*>(shift + n * C + q * 4);
  v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
  if (inner_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  if (res) {
    float4 r = ZtIO<T>::ld4(res + p * ldres + q * 4);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (outer_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  ZtIO<T>::st4(y + p * ldy + q * 4, v);
}

// BN(+ReLU) backward, stage 1: partial[(blk*2 + {0: sum dyh, 1: sum dyh*zhat})*C + c], dyh = dy * [x*scale+shift > 0]
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* __restrict__ dy, int lddy,
                                                            const T* __restrict__ z, int ldz,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int HW, int C, int nblk, float* __restrict__ partial) {
  __shared__ float4 sh_a[256];
  __shared__ float4 sh_b[256];
  const int Q = C >> 2, R = 256 / Q;
  const int tid = threadIdx.x;
  const int row = tid / Q, q = tid - row * Q;
  const int blk = blockIdx.x;
  const int chunk = (HW + nblk - 1) / nblk;
  const int p0 = blk * chunk, p1 = min(HW, p0 + chunk);
  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
  if (row < R) {
    float4 sc = *reinterpret_cast<const float4*>(scale + q * 4), sf = *reinterpret_cast<const float4*>(shift + q * 4);
    float4 mu = *reinterpret_cast<const float4*>(mean + q * 4), rs = *reinterpret_cast<const float4*>(rstd + q * 4);
    for (int p = p0 + row; p < p1; p += R) {
      float4 g = ZtIO<T>::ld4(dy + (size_t)p * lddy + q * 4);
      float4 v = ZtIO<T>::ld4(z + (size_t)p * ldz + q * 4);
      float gx = (v.x * sc.x + sf.x > 0.f) ? g.x : 0.f, gy = (v.y * sc.y + sf.y > 0.f) ? g.y : 0.f;
      float gz = (v.z * sc.z + sf.z > 0.f) ? g.z : 0.f, gw = (v.w * sc.w + sf.w > 0.f) ? g.w : 0.f;
      sa.x += gx; sa.y += gy; sa.z += gz; sa.w += gw;
      sb.x += gx * ((v.x - mu.x) * rs.x); sb.y += gy * ((v.y - mu.y) * rs.y);
      sb.z += gz * ((v.z - mu.z) * rs.z); sb.w += gw * ((v.w - mu.w) * rs.w);
    }
  }
  sh_a[tid] = sa;
  sh_b[tid] = sb;
  __syncthreads();
  if (row == 0) {
    for (int r = 1; r < R; ++r) {
      float4 a = sh_a[r * Q + q], b = sh_b[r * Q + q];
      sa.x += a.x; sa.y += a.y; sa.z += a.z; sa.w += a.w;
      sb.x += b.x; sb.y += b.y; sb.z += b.z; sb.w += b.w;
    }
    float* o = partial + ((size_t)blk * 2) * C + q * 4;
    *reinterpret_cast<float4*>(o) = sa;
    *reinterpret_cast<float4*>(o + C) = sb;
  }
}

// out[j] (+)= sum_b partial[b*stride + j]   (fp64 combine); optional second destination without accumulation.
// One workgroup per output element: 256 threads stride over the partials, fixed-order LDS tree (deterministic).
__global__ void __launch_bounds__(256) partial_reduce_kernel(const float* __restrict__ partial, int nblk, int stride, int n,
                                                             float* __restrict__ out, int accumulate,
                                                             float* __restrict__ out2) {
  __shared__ double sh[256];
  const int j = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partial[(size_t)b * stride + j];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float v = (float)sh[0];
    if (out) out[j] = accumulate ? out[j] + v : v;
    if (out2) out2[j] = v;
  }
}

// BatchNorm backward: the three column sums one layer needs from its [nblk][2][C] partials in one launch --
// sums[0:2C] = (sum dyh, sum dyh*zhat), dbeta += sums[0:C], dgamma += sums[C:2C].  Same summation as partial_reduce_kernel.
// rstd != NULL: the second half of the partials holds sum g (z - mean) (the data-gradient kernel's fused form), converted here to
// sum g xhat = rstd * sum g (z - mean)
__global__ void __launch_bounds__(256) bn_bwd_sums_kernel(const float* __restrict__ partial, int nblk, int C,
                                                          float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                          float* __restrict__ sums, const float* __restrict__ rstd) {
  __shared__ double sh[256];
  const int j = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += (double)partial[(size_t)b * 2 * C + j];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float v = (float)sh[0];
    if (rstd && j >= C) v *= rstd[j - C];
    sums[j] = v;
    if (j < C) dbeta[j] += v;
    else dgamma[j - C] += v;
  }
}

// dz = gamma*rstd * (dyh - sum_dyh/n - zhat * sum_dyh_zhat/n)
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ z,
                                                           int ldz, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ sums,
                                                           float inv_n, T* __restrict__ dz, int lddz, int C,
                                                           int eval_mode, long long total4) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int Q = C >> 2;
  int q = (int)(i % Q);
  long long p = i / Q;
  float4 g = ZtIO<T>::ld4(dy + p * lddy + q * 4);
  float4 v = ZtIO<T>::ld4(z + p * ldz + q * 4);
  float4 sc = *reinterpret_cast<const float4*>(scale + q * 4), sf = *reinterpret_cast<const float4*>(shift + q * 4);
  float4 mu = *reinterpret_cast<const float4*>(mean + q * 4), rs = *reinterpret_cast<const float4*>(rstd + q * 4);
  float4 s1 = *reinterpret_cast<const float4*>(sums + q * 4), s2 = *reinterpret_cast<const float4*>(sums + C + q * 4);
  float4 o;
#define ZT_BN1(f)                                             \
  {                                                           \
    float gg = (v.f * sc.f + sf.f > 0.f) ? g.f : 0.f;         \
    float zh = (v.f - mu.f) * rs.f;                           \
    o.f = eval_mode ? sc.f * gg : sc.f * (gg - s1.f * inv_n - zh * (s2.f * inv_n));   \
  }
  ZT_BN1(x) ZT_BN1(y) ZT_BN1(z) ZT_BN1(w)
#undef ZT_BN1
  ZtIO<T>::st4(dz + p * lddz + q * 4, o);
}

}  // namespace


// ---- bf16 fast paths for the full-resolution tensors (C % 8 == 0, 16-byte aligned rows): one 16-byte access per lane, PIX
// pixels per thread so the per-channel parameters are loaded once and several loads are in flight.  Thread i: channel octet
// i % Q8, pixels (i / Q8) + j * npg, j < PIX -- load j of a wave covers 64 / Q8 consecutive pixels (contiguous KBs).
constexpr int NPIX = 4;

__global__ void __launch_bounds__(256) norm_apply_bf16x8_kernel(const zt_bf16* __restrict__ x, int ldx,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const zt_bf16* __restrict__ res, int ldres,
                                                                zt_bf16* __restrict__ y, int ldy, int HW, int C, int npg,
                                                                int inner_relu, int outer_relu) {
  const int Q8 = C >> 3;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int o = (int)(i % Q8), pg = (int)(i / Q8);
  if (pg >= npg) return;
  const int n = blockIdx.y;                                     // sample: its own scale / shift row (InstanceNorm on a batch)
  x += (size_t)n * HW * ldx;
  y += (size_t)n * HW * ldy;
  if (res) res += (size_t)n * HW * ldres;
  scale += n * C;
  shift += n * C;
  float sc[8], sf[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    sc[c] = scale[o * 8 + c];
    sf[c] = shift[o * 8 + c];
  }
  float v[NPIX][8], r[NPIX][8];
#pragma unroll
  for (int j = 0; j < NPIX; ++j) {
    const int p = min(pg + j * npg, HW - 1);
    zt_ld8(x + (size_t)p * ldx + o * 8, v[j]);
    if (res) zt_ld8(res + (size_t)p * ldres + o * 8, r[j]);
  }
#pragma unroll
  for (int j = 0; j < NPIX; ++j) {
    const int p = pg + j * npg;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float t = v[j][c] * sc[c] + sf[c];
      if (inner_relu) t = fmaxf(t, 0.f);
      if (res) t += r[j][c];
      if (outer_relu) t = fmaxf(t, 0.f);
      v[j][c] = t;
    }
    if (p < HW) zt_st8(y + (size_t)p * ldy + o * 8, v[j]);
  }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_bf16x8_kernel(const zt_bf16* __restrict__ dy, int lddy,
                                                                  const zt_bf16* __restrict__ z, int ldz,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ sums, float inv_n,
                                                                  zt_bf16* __restrict__ dz, int lddz, int HW, int C, int npg,
                                                                  int eval_mode) {
  const int Q8 = C >> 3;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int o = (int)(i % Q8), pg = (int)(i / Q8);
  if (pg >= npg) return;
  float sc[8], sf[8], mu[8], rs[8], a1[8], a2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    sc[c] = scale[o * 8 + c];
    sf[c] = shift[o * 8 + c];
    mu[c] = mean[o * 8 + c];
    rs[c] = rstd[o * 8 + c];
    a1[c] = sums[o * 8 + c] * inv_n;
    a2[c] = sums[C + o * 8 + c] * inv_n;
  }
  float g[NPIX][8], v[NPIX][8];
#pragma unroll
  for (int j = 0; j < NPIX; ++j) {
    const int p = min(pg + j * npg, HW - 1);
    zt_ld8(dy + (size_t)p * lddy + o * 8, g[j]);
    zt_ld8(z + (size_t)p * ldz + o * 8, v[j]);
  }
#pragma unroll
  for (int j = 0; j < NPIX; ++j) {
    const int p = pg + j * npg;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float gg = (v[j][c] * sc[c] + sf[c] > 0.f) ? g[j][c] : 0.f;
      const float zh = (v[j][c] - mu[c]) * rs[c];
      g[j][c] = eval_mode ? sc[c] * gg : sc[c] * (gg - a1[c] - zh * a2[c]);
    }
    if (p < HW) zt_st8(dz + (size_t)p * lddz + o * 8, g[j]);
  }
}

// statistics: block = 256 threads = (256 / Q8) pixel rows x Q8 octets; NPIX rows in flight per thread.  MODE 0: sum x, sum x^2
// (chan_stats); MODE 1: sum g, sum g * xhat with g = dy masked by the ReLU of the BN output (bn_bwd_reduce).
template <int MODE>
__global__ void __launch_bounds__(256) stats_bf16x8_kernel(const zt_bf16* __restrict__ x, int ldx, const zt_bf16* __restrict__ dy,
                                                           int lddy, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, int HW, int C, int nblk,
                                                           float* __restrict__ partial, NormTail tail) {
  __shared__ float sh[2][8][256];
  const int Q8 = C >> 3, R = 256 / Q8;
  const int tid = threadIdx.x;
  const int row = tid / Q8, o = tid - row * Q8;
  const int n = blockIdx.y, blk = blockIdx.x;
  const int chunk = (HW + nblk - 1) / nblk;
  const int p0 = blk * chunk, p1 = min(HW, p0 + chunk);
  float sa[8], sb[8], sc[8], sf[8], mu[8], rs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    sa[c] = 0.f;
    sb[c] = 0.f;
    if (MODE == 1) {
      sc[c] = scale[o * 8 + c];
      sf[c] = shift[o * 8 + c];
      mu[c] = mean[o * 8 + c];
      rs[c] = rstd[o * 8 + c];
    }
  }
  if (row < R) {
    const zt_bf16* xb = x + (size_t)n * HW * ldx + o * 8;
    for (int p = p0 + row; p < p1; p += NPIX * R) {
      float v[NPIX][8], g[NPIX][8];
#pragma unroll
      for (int j = 0; j < NPIX; ++j) {
        const int pp = min(p + j * R, p1 - 1);
        zt_ld8(xb + (size_t)pp * ldx, v[j]);
        if (MODE == 1) zt_ld8(dy + (size_t)pp * lddy + o * 8, g[j]);
      }
#pragma unroll
      for (int j = 0; j < NPIX; ++j) {
        const bool ok = p + j * R < p1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (MODE == 0) {
            const float t = ok ? v[j][c] : 0.f;
            sa[c] += t;
            sb[c] += t * t;
          } else {
            const float gg = (ok && v[j][c] * sc[c] + sf[c] > 0.f) ? g[j][c] : 0.f;
            sa[c] += gg;
            sb[c] += gg * ((v[j][c] - mu[c]) * rs[c]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    sh[0][c][tid] = sa[c];
    sh[1][c][tid] = sb[c];
  }
  __syncthreads();
  // rows folded as a tree (R = 256 / Q8 is a power of two): log2(R) steps of 16 LDS updates per thread; a serial fold by the
  // Q8 threads of row 0 was 2 * 8 * R dependent LDS reads -- most of the kernel's time on the small RAFT maps
  for (int st = R >> 1; st >= 1; st >>= 1) {
    if (row < st) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        sh[0][c][tid] += sh[0][c][tid + st * Q8];
        sh[1][c][tid] += sh[1][c][tid + st * Q8];
      }
    }
    __syncthreads();
  }
  if (row == 0) {
    float* ob = partial + ((size_t)(n * nblk + blk) * 2) * C + o * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      ob[c] = sh[0][c][tid];
      ob[C + c] = sh[1][c][tid];
    }
  }
  if (MODE == 0 && tail.tickets) instnorm_tail(tail, partial, nblk, C, n);
}

static inline bool zt_x8_ok(const void* p, int ld, int C) { return C % 8 == 0 && ld % 8 == 0 && ((uintptr_t)p & 15) == 0 && 256 % (C / 8) == 0; }

static int launch_chan_stats(const void* x, int dt, int ldx, int N, int HW, int C, int nblk, float* partial, const NormTail& tail,
                             hipStream_t stream) {
  if (dt == 0)
    hipLaunchKernelGGL(chan_stats_kernel<float>, dim3(nblk, N), dim3(256), 0, stream, (const float*)x, ldx, HW, C, nblk, partial, tail);
  else if (zt_x8_ok(x, ldx, C) && HW >= 4096)
    hipLaunchKernelGGL(stats_bf16x8_kernel<0>, dim3(nblk, N), dim3(256), 0, stream, (const zt_bf16*)x, ldx, (const zt_bf16*)nullptr, 0,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, HW, C, nblk, partial,
                       tail);
  else
    hipLaunchKernelGGL(chan_stats_kernel<zt_bf16>, dim3(nblk, N), dim3(256), 0, stream, (const zt_bf16*)x, ldx, HW, C, nblk, partial,
                       tail);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_chan_stats_nhwc(const void* x, int dt, int ldx, int N, int HW, int C, int nblk, float* partial,
                                  hipStream_t stream) {
  ZT_REQUIRE(x && partial && C % 4 == 0 && C >= 4 && C <= 1024 && ldx % 4 == 0 && nblk > 0 && ((uintptr_t)x & 7) == 0);
  NormTail tail = {};
  return launch_chan_stats(x, dt, ldx, N, HW, C, nblk, partial, tail, stream);
}

extern "C" int zt_instance_norm_stats(const void* x, int dt, int ldx, int N, int HW, int C, int nblk, float* partial, float eps,
                                      unsigned* tickets, float* scale, float* shift, hipStream_t stream) {
  ZT_REQUIRE(x && partial && C % 4 == 0 && C >= 4 && C <= 1024 && ldx % 4 == 0 && nblk > 0 && ((uintptr_t)x & 7) == 0);
  ZT_REQUIRE(tickets && scale && shift && HW > 0);
  NormTail tail = {tickets, scale, shift, (double)HW, eps};
  return launch_chan_stats(x, dt, ldx, N, HW, C, nblk, partial, tail, stream);
}

extern "C" int zt_norm_finalize_f32(const float* partial, int nblk, int N, int C, long long count, float eps, int mode,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float momentum, float* scale, float* shift,
                                    float* mean_out, float* rstd_out, hipStream_t stream) {
  ZT_REQUIRE(scale && shift && (mode == 2 || partial) && (mode == 0 || (gamma && beta)));
  ZT_REQUIRE(mode == 0 || (running_mean && running_var));
  ZT_REQUIRE(mode != 1 || N == 1);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(N), dim3(1024), 0, stream, partial, nblk, C, (double)count, eps, mode, gamma,
                     beta, running_mean, running_var, num_batches_tracked, momentum, scale, shift, mean_out, rstd_out);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_norm_apply_nhwc(const void* x, int dt, int ldx, const float* scale, const float* shift, const void* res,
                                  int ldres, void* y, int ldy, int N, int HW, int C, int inner_relu, int outer_relu,
                                  hipStream_t stream) {
  ZT_REQUIRE(x && y && scale && shift && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (!res || ldres % 4 == 0));
  long long total4 = (long long)N * HW * (C / 4);
  // elementwise: any C % 8 == 0 (the statistics kernels additionally need 256 % (C / 8) == 0), one grid row per sample
  auto x8 = [&](const void* p, int ld) { return C % 8 == 0 && ld % 8 == 0 && ((uintptr_t)p & 15) == 0; };
  if (dt != 0 && N <= 65535 && HW >= 4096 && x8(x, ldx) && x8(y, ldy) && (!res || x8(res, ldres))) {
    const int npg = zt_cdiv(HW, NPIX);
    hipLaunchKernelGGL(norm_apply_bf16x8_kernel, dim3((unsigned)zt_cdivl((long long)npg * (C / 8), 256), N), dim3(256), 0, stream,
                       (const zt_bf16*)x, ldx, scale, shift, (const zt_bf16*)res, ldres, (zt_bf16*)y, ldy, HW, C, npg, inner_relu, outer_relu);
  } else if (dt == 0)
    hipLaunchKernelGGL(norm_apply_kernel<float>, dim3((unsigned)zt_cdivl(total4, 256)), dim3(256), 0, stream, (const float*)x, ldx,
                       scale, shift, (const float*)res, ldres, (float*)y, ldy, HW, C, inner_relu, outer_relu, total4);
  else
    hipLaunchKernelGGL(norm_apply_kernel<zt_bf16>, dim3((unsigned)zt_cdivl(total4, 256)), dim3(256), 0, stream, (const zt_bf16*)x,
                       ldx, scale, shift, (const zt_bf16*)res, ldres, (zt_bf16*)y, ldy, HW, C, inner_relu, outer_relu, total4);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_bn_bwd_reduce(const void* dy, int dt, int lddy, const void* z, int ldz, const float* scale,
                                const float* shift, const float* mean, const float* rstd, int HW, int C, int nblk,
                                float* partial, hipStream_t stream) {
  ZT_REQUIRE(dy && z && partial && C % 4 == 0 && C <= 1024);
  if (dt != 0 && HW >= 4096 && zt_x8_ok(dy, lddy, C) && zt_x8_ok(z, ldz, C))
    hipLaunchKernelGGL(stats_bf16x8_kernel<1>, dim3(nblk, 1), dim3(256), 0, stream, (const zt_bf16*)z, ldz, (const zt_bf16*)dy, lddy, scale,
                       shift, mean, rstd, HW, C, nblk, partial, NormTail{});
  else if (dt == 0)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nblk), dim3(256), 0, stream, (const float*)dy, lddy, (const float*)z, ldz,
                       scale, shift, mean, rstd, HW, C, nblk, partial);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<zt_bf16>, dim3(nblk), dim3(256), 0, stream, (const zt_bf16*)dy, lddy, (const zt_bf16*)z,
                       ldz, scale, shift, mean, rstd, HW, C, nblk, partial);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_partial_reduce_f32(const float* partial, int nblk, int stride, int n, float* out, int accumulate,
                                     float* out2, hipStream_t stream) {
  ZT_REQUIRE(partial && n > 0 && (out || out2));
  hipLaunchKernelGGL(partial_reduce_kernel, dim3(n), dim3(256), 0, stream, partial, nblk, stride, n, out, accumulate, out2);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_bn_bwd_sums_f32(const float* partial, int nblk, int C, float* dbeta, float* dgamma, float* sums,
                                  hipStream_t stream) {
  ZT_REQUIRE(partial && nblk > 0 && C > 0 && dbeta && dgamma && sums);
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(2 * C), dim3(256), 0, stream, partial, nblk, C, dbeta, dgamma, sums, (const float*)nullptr);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_bn_bwd_sums_centered_f32(const float* partial, int nblk, int C, const float* rstd, float* dbeta, float* dgamma, float* sums,
                                           hipStream_t stream) {
  ZT_REQUIRE(partial && nblk > 0 && C > 0 && rstd && dbeta && dgamma && sums);
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(2 * C), dim3(256), 0, stream, partial, nblk, C, dbeta, dgamma, sums, rstd);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_bn_bwd_apply(const void* dy, int dt, int lddy, const void* z, int ldz, const float* scale,
                               const float* shift, const float* mean, const float* rstd, const float* sums,
                               void* dz, int lddz, int HW, int C, int eval_mode, hipStream_t stream) {
  ZT_REQUIRE(dy && z && dz && sums && C % 4 == 0);
  long long total4 = (long long)HW * (C / 4);
  if (dt != 0 && HW >= 4096 && zt_x8_ok(dy, lddy, C) && zt_x8_ok(z, ldz, C) && zt_x8_ok(dz, lddz, C)) {
    const int npg = zt_cdiv(HW, NPIX);
    hipLaunchKernelGGL(bn_bwd_apply_bf16x8_kernel, dim3((unsigned)zt_cdivl((long long)npg * (C / 8), 256)), dim3(256), 0, stream,
                       (const zt_bf16*)dy, lddy, (const zt_bf16*)z, ldz, scale, shift, mean, rstd, sums, 1.f / (float)HW, (zt_bf16*)dz, lddz,
                       HW, C, npg, eval_mode);
  } else if (dt == 0)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3((unsigned)zt_cdivl(total4, 256)), dim3(256), 0, stream, (const float*)dy,
                       lddy, (const float*)z, ldz, scale, shift, mean, rstd, sums, 1.f / (float)HW, (float*)dz, lddz, C, eval_mode, total4);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<zt_bf16>, dim3((unsigned)zt_cdivl(total4, 256)), dim3(256), 0, stream, (const zt_bf16*)dy,
                       lddy, (const zt_bf16*)z, ldz, scale, shift, mean, rstd, sums, 1.f / (float)HW, (zt_bf16*)dz, lddz, C, eval_mode, total4);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
