/* C ABI of libzerotig_hip.so -- the drop-in boundary for the Zero-TIG hot path on MI355X (gfx950).
 *
 * The reference has no FFI: its seam is the Python surface (model.model.Network / loss.LossFunction / utils.utils,
 * SURVEY 8(b)).  The one native hook it anticipates is `alt_cuda_corr.forward` (model/RAFT/corr.py:86).  Each entry
 * point below replaces the ATen call(s) of the cited reference lines.  Conventions:
 *   - plain pointers to DEVICE memory + sizes; no torch types; the caller owns every buffer (inputs, outputs,
 *     workspaces) and keeps it alive until the stream reaches the next op; nothing here allocates or synchronises;
 *   - every function enqueues on `stream` (hipStream_t passed as void*) and returns 0, a hipError_t, or 1001
 *     (invalid argument); the Python host raises RuntimeError on non-zero (zero-tig_amd/lib.py);
 *   - "planar" = [C][H][W] fp32 (the reference's NCHW with N == 1); "nhwc" = [N][H][W][ld] fp32 with an explicit
 *     channel stride `ld` so channel slices of wider buffers can be addressed without copies.
 */
#ifndef ZEROTIG_HIP_H
#define ZEROTIG_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* zt_stream_t;

/* utils/utils.py:203-230 warp_tensor x2 (model.py:249-250), fused.  flow: planar [2][Hf][Wf]; imgA/imgB/outA/outB: planar
 * [C][H][W] (imgB/outB may both be NULL); taps: optional int32 [H][W][2] = floor of the source coordinates (x0,y0). */
int zt_warp2_f32(const float* flow, int Hf, int Wf, const float* imgA, const float* imgB, float* outA, float* outB,
                 int* taps, int C, int H, int W, zt_stream_t stream);


/* ---- stencil primitives on planar tensors (zt_stencil.hip) ------------------------------------------------ */
/* utils/utils.py:15-24 pair_downsampler: o1 = (x[2y,2x+1]+x[2y+1,2x])/2, o2 = (x[2y,2x]+x[2y+1,2x+1])/2; outputs [C][H/2][W/2] */
int zt_pair_down_f32(const float* src, float* o1, float* o2, int C, int H, int W, zt_stream_t stream);
/* adjoint of the above: dst[C][H][W] (+)= pd^T(g1, g2) */
int zt_pair_down_adj_f32(const float* g1, const float* g2, float* dst, int C, int H, int W, int accumulate, zt_stream_t stream);
/* utils/utils.py:52-58 blur: reflect-pad 10 + 21x21 Gaussian, as two 21-tap passes; taps21_host: HOST pointer to the 1-D factor */
int zt_blur21_f32(const float* src, float* tmp, float* dst, const float* taps21_host, int C, int H, int W, zt_stream_t stream);
int zt_blur21_adj_f32(const float* g, float* tmp, float* dst, const float* taps21_host, int C, int H, int W, int accumulate, zt_stream_t stream);
/* utils/utils.py:41-50 LocalMean (reflect-pad 2, 5x5 mean) and its adjoint (dst (+)= scale * LM^T(src)) */
int zt_box5_reflect_f32(const float* src, float* dst, int C, int H, int W, zt_stream_t stream);
int zt_box5_reflect_adj_f32(const float* src, float* dst, int C, int H, int W, float scale, int accumulate, zt_stream_t stream);
/* utils/utils.py:60-79 calculate_local_variance of x = a - b (b may be NULL): D = x - box0(x)/25 (saved for backward, may be NULL), V = box0(D^2)/25 */
int zt_localvar_fwd_f32(const float* a, const float* b, float* D, float* V, int C, int H, int W, zt_stream_t stream);
/* backward: xbar (+)= sign * (E - box0(E)/25), E = 2 D box0(gV)/25 */
int zt_localvar_bwd_f32(const float* D, const float* gV, float* xbar, int C, int H, int W, float sign, int accumulate, zt_stream_t stream);
/* loss.py:99-136 TextureDifference: a, b planar [3][H][W] -> mask [H][W] in {0,1}; ratio (optional) = 2 s1 s2 / (s1^2+s2^2+1e-5) */
int zt_texture_mask_f32(const float* a, const float* b, float* mask, float* ratio, int H, int W, zt_stream_t stream);
/* loss.py:178-190 SmoothLoss.rgb2yCbCr over the flat memory (nelem = 3*H*W) */
int zt_ycc_flat_f32(const float* src, float* dst, long long nelem, zt_stream_t stream);


/* ---- MFMA implicit-GEMM convolution family (zt_conv.hip) -----------------------------------------------------
 * Replaces F.conv2d of model.py:20-27, 36-43, 55-80, every conv of model/RAFT/{extractor,update}.py, and (as a 1x1
 * conv whose "weights" are fmap2) the all-pairs matmul of corr.py:52-60.
 * x: nhwc [N][H][W][ldx] (first Cin channels used; channels >= csplit come from x2 [..][ldx2] when x2 != NULL);
 * w: device layout [KH*KW][Cin][ldw] (zt_repack_conv_weight_f32); y: nhwc [N][Ho][Wo][ldy], or planar
 * [N][Cout] planes of pitch ldy when out_planar; y = act(alpha * (conv + bias)) then epi: 0 none,
 * 1: *= LeakyReLU'(aux) (aux>0 ? 1 : 0.2), 2: *= (aux>0), 3: += aux  (aux nhwc, stride ldaux).
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.2), 3 sigmoid, 4 tanh, 5 clamp(sigmoid, 1e-4, 1).
 * Supported (KH,KW,stride): (3,3,1|2) (1,1,1|2) (1,5,1) (5,1,1) (7,7,1|2). */
int zt_conv2d_nhwc_f32(const float* x, const float* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin,
                       const float* w, int ldw, const float* bias, float* y, int ldy, int out_planar, int Cout, int KH,
                       int KW, int stride, int padH, int padW, int act, float alpha, const float* aux, int ldaux,
                       int epi, zt_stream_t stream);
/* weight gradient of a stride-1 "same" conv (autograd of the above): grad_w [Cout][Cin][KH][KW] (torch layout)
 * (+)= sum_p x[p+tap][ci] dz[p][co]; slab: workspace for per-workgroup partials (deterministic reduction). */
int zt_conv2d_wgrad_nhwc_f32(const float* x, int ldx, const float* dz, int lddz, int H, int W, int Cin, int Cout, int KH,
                             int KW, float* slab, size_t slab_bytes, float* grad_w, int accumulate, zt_stream_t stream);
/* torch weight [Cout][Cin][KH][KW] -> device layout [tap][Cin][ldw] at column offset co_off (transpose_flip = 0), or the
 * data-gradient operator [tap flipped][Cout][ldw] with in/out channels exchanged (transpose_flip = 1). */
int zt_repack_conv_weight_f32(const float* src, float* dst, int Cout, int Cin, int KH, int KW, int ldw, int co_off,
                              int transpose_flip, zt_stream_t stream);


/* ---- normalisation (zt_norm.hip): nn.BatchNorm2d of model.py:62 (train mode, shared 3x), eval BatchNorm /
 * InstanceNorm of model/RAFT/extractor.py:117-191.  All on nhwc buffers, C % 4 == 0. ------------------------------- */
/* per-(n,c) partial sums over HW pixels: partial[((n*nblk+blk)*2 + {0 sum, 1 sum of squares})*C + c] */
int zt_chan_stats_nhwc_f32(const float* x, int ldx, int N, int HW, int C, int nblk, float* partial, zt_stream_t stream);
/* mode 0 instance norm (scale = rstd, shift = -mean*rstd); 1 train BN (batch stats, running stats += momentum update with
 * unbiased variance, *num_batches_tracked += 1); 2 eval BN (running stats).  Outputs scale/shift (and mean/rstd) [N][C]. */
int zt_norm_finalize_f32(const float* partial, int nblk, int N, int C, long long count, float eps, int mode,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float momentum, float* scale, float* shift, float* mean_out,
                         float* rstd_out, zt_stream_t stream);
/* y = [outer_relu]([res +] [inner_relu](x*scale + shift)) */
int zt_norm_apply_nhwc_f32(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldres,
                           float* y, int ldy, int N, int HW, int C, int inner_relu, int outer_relu, zt_stream_t stream);
/* backward of y = ReLU(BN_train(z)): stage 1 partial sums of dyh = dy*[bn>0] and dyh*zhat; generic partial reduction;
 * stage 2 dz = gamma*rstd*(dyh - mean(dyh) - zhat*mean(dyh*zhat)) with sums = [sum dyh (C) | sum dyh*zhat (C)] */
int zt_bn_bwd_reduce_f32(const float* dy, int lddy, const float* z, int ldz, const float* scale, const float* shift,
                         const float* mean, const float* rstd, int HW, int C, int nblk, float* partial, zt_stream_t stream);
int zt_partial_reduce_f32(const float* partial, int nblk, int stride, int n, float* out, int accumulate, float* out2,
                          zt_stream_t stream);
int zt_bn_bwd_apply_f32(const float* dy, int lddy, const float* z, int ldz, const float* scale, const float* shift,
                        const float* mean, const float* rstd, const float* sums, float* dz, int lddz, int HW, int C,
                        zt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
