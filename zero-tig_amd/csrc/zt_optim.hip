// Optimizer step of the reference training loop (train.py:130-131): clip_grad_norm_(params, 5) followed by
// torch.optim.Adam(lr, betas, eps=1e-8, weight_decay (classic L2)) over ONE flat fp32 bucket holding all 20 trainable
// tensors (92 620 floats) -- the same bucket that is all-reduced across ranks in data-parallel runs.
#include "zt_common.h"

namespace {

__global__ void __launch_bounds__(256) sqnorm_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
  __shared__ float red[16];
  float v[1] = {0.f};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) v[0] += g[i] * g[i];
  zt_block_sum<1>(v, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = v[0];
}

__global__ void __launch_bounds__(256) clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, const float* __restrict__ partial,
                                                        int nblk, float gscale, float max_norm, float step_size, float b1,
                                                        float b2, float eps, float wd, float inv_sqrt_bc2,
                                                        float* __restrict__ gnorm_out) {
  __shared__ double tot;
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)partial[b];
    tot = s;
  }
  __syncthreads();
  float norm = (float)sqrt(tot) * gscale;
  float coef = fminf(max_norm / (norm + 1e-6f), 1.f);
  if (max_norm <= 0.f) coef = 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0 && gnorm_out) *gnorm_out = norm;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float pi = p[i];
  float gi = g[i] * gscale * coef + wd * pi;
  float mi = b1 * m[i] + (1.f - b1) * gi;
  float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = pi - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}

// y += (*alpha) * x: hands the step's gradients to the optimizer's bucket (loss.backward() with an upstream gradient on device)
__global__ void __launch_bounds__(256) axpy_dev_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ alpha,
                                                       long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = fmaf(alpha[0], x[i], y[i]);
}

}  // namespace

extern "C" int zt_axpy_dev_f32(float* y, const float* x, const float* alpha, long long n, hipStream_t stream) {
  ZT_REQUIRE(y && x && alpha && n > 0);
  hipLaunchKernelGGL(axpy_dev_kernel, dim3((unsigned)zt_cdivl(n, 256)), dim3(256), 0, stream, y, x, alpha, n);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}

extern "C" int zt_clip_adam_f32(float* p, const float* g, float* m, float* v, long long n, float* partial, int nblk, float gscale,
                                float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                long long step, float* gnorm_out, hipStream_t stream) {
  ZT_REQUIRE(p && g && m && v && partial && n > 0 && nblk > 0 && nblk <= 4096 && step >= 1);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblk), dim3(256), 0, stream, g, n, partial);
  double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)zt_cdivl(n, 256)), dim3(256), 0, stream, p, g, m, v, n, (const float*)partial,
                     nblk, gscale, max_norm, step_size, beta1, beta2, eps, weight_decay, inv_sqrt_bc2, gnorm_out);
  ZT_LAUNCH_CHECK();
  return ZT_OK;
}
