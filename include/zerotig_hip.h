/* C ABI of libzerotig_hip.so -- the drop-in boundary for the Zero-TIG hot path on MI355X (gfx950).
 *
 * The reference has no FFI: its seam is the Python surface (model.model.Network / loss.LossFunction / utils.utils,
 * SURVEY 8(b)).  The one native hook it anticipates is `alt_cuda_corr.forward` (model/RAFT/corr.py:86).  Each entry
 * point below replaces the ATen call(s) of the cited reference lines.  Conventions:
 *   - plain pointers to DEVICE memory + sizes; no torch types; the caller owns every buffer (inputs, outputs,
 *     workspaces) and keeps it alive until the stream reaches the next op; nothing here allocates or synchronises;
 *   - every function enqueues on `stream` (hipStream_t passed as void*) and returns 0, a hipError_t, or 1001
 *     (invalid argument); the Python host raises RuntimeError on non-zero (zero-tig_amd/lib.py);
 *   - `dt` arguments select the storage type of nhwc activation buffers: 0 = fp32, 1 = bf16 (raw 16-bit, round-to-nearest-even)
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
/* same with the fused SepConvGRU epilogues of model/RAFT/update.py:42-58 (nhwc output only):
 * epi 4: [z | r] = act(conv): channels < esplit are stored to y; channels >= esplit leave as r * h to y2 [..][ldy2] at channel
 *        co - esplit (aux = h, nhwc ldaux)                                    == `torch.cat`-free `r * h` of update.py:44/52
 * epi 5: q = act(conv): y (= h, read and written in place) <- (1 - z) * h + z * q, aux = z   == update.py:46/54 */
int zt_conv2d_nhwc_f32_ex(const float* x, const float* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin,
                          const float* w, int ldw, const float* bias, float* y, int ldy, int out_planar, int Cout, int KH,
                          int KW, int stride, int padH, int padW, int act, float alpha, const float* aux, int ldaux,
                          int epi, float* y2, int ldy2, int esplit, zt_stream_t stream);
/* weight gradient of a stride-1 "same" conv (autograd of the above): grad_w [Cout][Cin][KH][KW] (torch layout)
 * (+)= sum_p x[p+tap][ci] dz[p][co]; grad_b [Cout] (optional) (+)= sum_p dz[p][co] (bias gradient, folded into the same pass);
 * slab: workspace for per-workgroup partials (deterministic reduction). */
int zt_conv2d_wgrad_nhwc_f32(const float* x, int ldx, const float* dz, int lddz, int H, int W, int Cin, int Cout, int KH,
                             int KW, float* slab, size_t slab_bytes, float* grad_w, float* grad_b, int accumulate, zt_stream_t stream);
/* torch weight [Cout][Cin][KH][KW] -> device layout [tap][Cin][ldw] at column offset co_off (transpose_flip = 0), or the
 * data-gradient operator [tap flipped][Cout][ldw] with in/out channels exchanged (transpose_flip = 1). */
int zt_repack_conv_weight_f32(const float* src, float* dst, int Cout, int Cin, int KH, int KW, int ldw, int co_off,
                              int transpose_flip, zt_stream_t stream);


/* ---- normalisation (zt_norm.hip): nn.BatchNorm2d of model.py:62 (train mode, shared 3x), eval BatchNorm /
 * InstanceNorm of model/RAFT/extractor.py:117-191.  All on nhwc buffers, C % 4 == 0. ------------------------------- */
/* per-(n,c) partial sums over HW pixels: partial[((n*nblk+blk)*2 + {0 sum, 1 sum of squares})*C + c] */
int zt_chan_stats_nhwc(const void* x, int dt, int ldx, int N, int HW, int C, int nblk, float* partial, zt_stream_t stream);
/* mode 0 instance norm (scale = rstd, shift = -mean*rstd); 1 train BN (batch stats, running stats += momentum update with
 * unbiased variance, *num_batches_tracked += 1); 2 eval BN (running stats).  Outputs scale/shift (and mean/rstd) [N][C]. */
int zt_norm_finalize_f32(const float* partial, int nblk, int N, int C, long long count, float eps, int mode,
                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                         long long* num_batches_tracked, float momentum, float* scale, float* shift, float* mean_out,
                         float* rstd_out, zt_stream_t stream);
/* InstanceNorm statistics and scale/shift in one launch (RAFT feature encoder, reference model/RAFT/extractor.py:117-191,
   nn.InstanceNorm2d: biased variance, no affine): partial rows as zt_chan_stats_nhwc; the last workgroup of each sample writes
   scale[n*C+c] = rstd, shift = -mean*rstd.  tickets: N zero-initialised counters, left at zero. */
int zt_instance_norm_stats(const void* x, int dt, int ldx, int N, int HW, int C, int nblk, float* partial, float eps,
                           unsigned* tickets, float* scale, float* shift, zt_stream_t stream);
/* y = [outer_relu]([res +] [inner_relu](x*scale + shift)) */
int zt_norm_apply_nhwc(const void* x, int dt, int ldx, const float* scale, const float* shift, const void* res, int ldres,
                           void* y, int ldy, int N, int HW, int C, int inner_relu, int outer_relu, zt_stream_t stream);
/* backward of y = ReLU(BN_train(z)): stage 1 partial sums of dyh = dy*[bn>0] and dyh*zhat; generic partial reduction;
 * stage 2 dz = gamma*rstd*(dyh - mean(dyh) - zhat*mean(dyh*zhat)) with sums = [sum dyh (C) | sum dyh*zhat (C)];
 * eval_mode = 1: statistics are constants (running stats): dz = gamma*rstd*dyh (reference epochs >= 1, train.py:138) */
int zt_bn_bwd_reduce(const void* dy, int dt, int lddy, const void* z, int ldz, const float* scale, const float* shift,
                         const float* mean, const float* rstd, int HW, int C, int nblk, float* partial, zt_stream_t stream);
int zt_partial_reduce_f32(const float* partial, int nblk, int stride, int n, float* out, int accumulate, float* out2,
                          zt_stream_t stream);
/* one launch for a BatchNorm layer's backward sums: sums[0:2C] = column sums of partial [nblk][2][C]; dbeta += sums[0:C],
   dgamma += sums[C:2C] (torch BatchNorm2d backward, reference model.py:60-67 through autograd) */
int zt_bn_bwd_sums_f32(const float* partial, int nblk, int C, float* dbeta, float* dgamma, float* sums, zt_stream_t stream);
/* same for partials whose second half is sum g (z - mean) (zt_conv3x3_dgrad_bn_sums_bf16): multiplied by rstd on the way */
int zt_bn_bwd_sums_centered_f32(const float* partial, int nblk, int C, const float* rstd, float* dbeta, float* dgamma, float* sums,
                                zt_stream_t stream);
int zt_bn_bwd_apply(const void* dy, int dt, int lddy, const void* z, int ldz, const float* scale, const float* shift,
                        const float* mean, const float* rstd, const float* sums, void* dz, int lddz, int HW, int C,
                        int eval_mode, zt_stream_t stream);


/* ---- element-wise stages of Network.forward (model.py:144-203) and their backward (zt_glue.hip); planar [3][H][W], H,W even */
/* model.py:145-148, loss.py:25,51: x = inp+1e-4; (L11,L12) = pair_downsampler(x); (Lq11,Lq12) = pair_downsampler(inp+1e-9) */
int zt_prep_input_f32(const float* inp, float* x, float* L11, float* L12, float* Lq11, float* Lq12, int H, int W, zt_stream_t stream);
/* torch.cat([...],1) of up to four planar tensors into one nhwc buffer (remaining channels zero) (model.py:168,179,184,189) */
int zt_pack_nhwc(void* dst, int dt, int ld, long long HW, const float* s0, int c0, const float* s1, int c1, const float* s2, int c2,
                     const float* s3, int c3, zt_stream_t stream);
/* model.py:149-152 + loss.py:54: L2 = clamp(x-n,1e-4,1); L_pred1/2 = L11/12 - n11/12; (den1,den2) = pair_downsampler(L2) */
int zt_d1_tail_f32(const float* x, const float* n, const float* L11, const float* n11, const float* L12, const float* n12,
                   float* L2, float* Lp1, float* Lp2, float* den1, float* den2, int H, int W, zt_stream_t stream);
/* model.py:169-177,198-199 */
int zt_post_enh_f32(const float* x, const float* s2, const float* L2, const float* L11, const float* L12, float* s21, float* s22,
                    float* H2, float* H11, float* H12, float* H1, int H, int W, zt_stream_t stream);
/* model.py:179-192: (outA|outB) = clamp(cat[A,B] - r, 1e-4, 1), r planar 6ch */
int zt_clamp_sub6_f32(const float* A, const float* B, const float* r, float* outA, float* outB, long long HW, zt_stream_t stream);
/* its backward into the nhwc gradient of r */
int zt_clamp_sub6_bwd(const float* A, const float* B, const float* r, const float* gA, const float* gB, void* dr, int dt, int ld,
                          long long HW, zt_stream_t stream);
/* all gradient paths into s2 (H2, H11/H12, pair_downsampler, Denoise_2 inputs, direct loss terms) -> Enhancer output-layer gradient (nhwc) */
int zt_post_enh_bwd(const float* x, const float* s2, const float* L11, const float* L12, const float* s21, const float* s22,
                        const float* dIn5, const float* dH2x, const float* dIn3, const float* dIn4, const float* ds2_direct,
                        void* dO, int dt, int ld, float* ds2_total, int H, int W, zt_stream_t stream);
/* gradients entering the three Denoise_1 invocations (nhwc) */
int zt_d1_bwd_prep(const float* x, const float* n, const float* dLp1, const float* dLp2, const float* dden1,
                       const float* dden2, void* dn, void* dn11, void* dn12, int dt, int ld, int H, int W, zt_stream_t stream);
/* ReLU backward on nhwc buffers: out = g * [a > 0] */
int zt_relu_mask_nhwc(const void* g, int dt, int ldg, const void* a, int lda, void* out, int ldo, long long npix, int C, zt_stream_t stream);
/* element-wise helpers of Finetunemodel.forward (model.py:313-316,327-328): mode 0 a+p0; 1 clamp(a-b,p0,p1); 2 clamp(a/b,p0,p1) */
int zt_ew_f32(const float* a, const float* b, float* out, int mode, float p0, float p1, long long n, zt_stream_t stream);
/* utils.py:228 overlap_tensor = 0.5*warped + 0.5*img2: out = alpha*a + beta*b */
int zt_axpby_f32(const float* a, const float* b, float* out, float alpha, float beta, long long n, zt_stream_t stream);
int zt_add3_f32(const float* a, const float* b, const float* c, float* out, long long n, zt_stream_t stream);

/* ---- LossFunction.forward (loss.py:12-78) + gradient w.r.t. its inputs (zt_loss.hip) ------------------------------- */
int zt_plane_sums_f32(const float* x, int C, long long HW, int nblk, float* partial, zt_stream_t stream);
/* loss.py:26-37: scal[0..2] = clamp(enhancement_factor,1,25), scal[3..5] = 0.7^-ef / ef */
int zt_loss_scalars_f32(const float* partial, int nblk, long long HW, int is_WB, float* scal, zt_stream_t stream);
/* loss.py:46-49 (700 MSE, 1000 MSE, 5 SmoothLoss, 1600 L_TV): partial[block][4] + direct d/ds2.  fast_exp: 1 = hardware exponential for
 * the bilateral weights (bf16 throughput mode), 0 = libm expf (fp32 parity mode) */
int zt_loss_s2_f32(const float* L2, const float* s2, const float* Y, const float* scal, int H, int W, float* ds2, float* partial,
                   int fast_exp, zt_stream_t stream);
/* loss.py:51-62, 68-73: partial[block][10] = res1_a..d, res2_a..d, inter_a, inter_b; direct gradients; u1/u2 = (1-m)*de (to be LocalMean-adjointed) */
int zt_loss_half_f32(const float* Lq11, const float* Lq12, const float* Lp1, const float* Lp2, const float* den1, const float* den2,
                     const float* H3p, const float* H4p, const float* H11, const float* s21, const float* H12, const float* s22,
                     const float* H3d1, const float* H3d2, const float* mask, const float* LM1, const float* LM2, float* dLp1,
                     float* dLp2, float* dden1, float* dden2, float* dH3p, float* dH4p, float* dH3d1, float* dH3d2, float* u1,
                     float* u2, long long hw, float* partial, zt_stream_t stream);
/* loss.py:64, 66, 75-77: partial[block][3] = color, ill, var; dH3_blur, ds3, gV = d/dV(H2) (= -d/dV(H3-H2)) */
int zt_loss_full_f32(const float* H2b, const float* H3b, const float* s2, const float* s3, const float* VH2, const float* VN,
                     float* dH3b, float* ds3, float* gV, long long n, float* partial, zt_stream_t stream);


/* ---- RAFT / update_cache specific kernels (zt_raft.hip) ------------------------------------------------------------ */
/* F.interpolate(bilinear, align_corners=False) * mul on planar tensors (model.py:226-227,231), ATen CPU rounding sequence */
int zt_resize_bilinear_f32(const float* src, float* dst, int C, int H, int W, int h, int w, float mul, zt_stream_t stream);
/* (x).to(uint8) + per-channel histogram + torchvision==0.18.1 equalize LUT (model.py:234); q: uint8 [C][hw], hist/lut: int32 [C][256] */
int zt_equalize_prepare_u8(const float* src, unsigned char* q, int* hist, int* lut, int C, int hw, zt_stream_t stream);
/* raft.py:80-83,132-138: centred replicate pad to multiples of 8, 2*(x/255)-1; frame 1 float, frame 2 = lut[q2]; dst nhwc [2][Hp][Wp][ld] (dt: 0 fp32, 1 bf16) */
int zt_raft_pack_input(const float* img1, const unsigned char* q2, const int* lut, void* dst, int dt, int ld, int h, int w, int Hp, int Wp, zt_stream_t stream);
/* the same head for two float frames in [0,255] (RAFT.forward called directly, raft.py:77-83) */
int zt_raft_pack_pair(const float* img1, const float* img2, void* dst, int dt, int ld, int h, int w, int Hp, int Wp, zt_stream_t stream);
/* corr.py:25-27 one pyramid level: avg_pool2d(2,2) of [npx][hin][win] (row pitch ldin) -> [npx][hin/2][win/2] */
int zt_corr_pool_f32(const float* src, float* dst, int npx, int hin, int win, int ldin, zt_stream_t stream);
/* corr.py:13-27, 52-60 in one pass, bf16 feature maps: c0[m][y2*w + x2] = alpha * <f1[m], f2[y2*w + x2]> (fp32 [npx][ld0]) and the
 * three avg_pool2d(2,2) levels c1 [npx][h/2 * w/2], c2, c3 (floor sizes), each from the rounded level below.  f1 / f2: nhwc bf16
 * [npx][ld >= 256] (256 channels).  Replaces zt_conv2d_nhwc_bf16 (as a 1x1 GEMM) + 3 x zt_corr_pool_f32. */
int zt_corr_volume_pyramid_bf16(const void* f1, int ld1, const void* f2, int ld2, int h, int w, float alpha, float* c0, int ld0, float* c1,
                                float* c2, float* c3, zt_stream_t stream);
/* corr.py:29-50 (the seam of alt_cuda_corr.forward, corr.py:86): coords [npx][2] -> out nhwc [npx][ldo>=324], channel = lvl*81 + i*9 + j */
int zt_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3, int h, int w, int ld0, const float* coords,
                   void* out, int dt, int ldo, int npx, zt_stream_t stream);
/* The same lookup with the flow bookkeeping of the PREVIOUS refinement iteration folded in (raft.py:112-126: `coords1 = coords1 +
 * delta_flow` feeds the next iteration's `corr_fn(coords1)`): looks up at coords + delta (delta [npx][ldd] fp32, may be NULL) and
 * records coords_out = coords + delta (a different buffer than coords), flow = coords_out - grid into f4 (fp32 [npx][ldf4]) and the
 * nhwc destinations fhx / fin of storage type dt (each may be NULL) -- one launch less per iteration than lookup + zt_raft_flow_step,
 * bit-identical values. */
int zt_corr_lookup_step(const float* l0, const float* l1, const float* l2, const float* l3, int h, int w, int ld0, const float* coords,
                        void* out, int dt, int ldo, int npx, const float* delta, int ldd, float* coords_out, float* f4, int ldf4,
                        void* fhx, int ldfhx, void* fin, int ldfin, zt_stream_t stream);
/* update.py:42-45: rh = r*h with zr = [z|r]; update.py:47: h = (1-z)h + z q */
int zt_gru_rh(const void* zr, int dt, int ldzr, const void* hbuf, int ldh, void* rh, int ldrh, int C, int npx, zt_stream_t stream);
int zt_gru_update(const void* zr, int dt, int ldzr, const void* q, int ldq, void* hbuf, int ldh, int C, int npx, zt_stream_t stream);
/* raft.py:57-62,112-120: coords grid; coords1 += delta (may be NULL), flow = coords1 - coords0 written to f4 (fp32) and to up to two nhwc destinations of storage type dt */
int zt_raft_coords_init_f32(float* coords, int h, int w, zt_stream_t stream);
int zt_raft_flow_step(float* coords1, const float* delta, int ldd, int h, int w, float* f4, int ldf4, void* fhx, int ldfhx, void* fin, int ldfin, int dt, zt_stream_t stream);
/* raft.py:64-75 upsample_flow: flow nhwc (ldf), mask nhwc [npx][576] -> planar [2][8h][8w]; optional planar flow_low [2][h][w] */
int zt_convex_upsample_f32(const float* f4, int ldf, const float* mask, int ldm, float* up, float* flow_low, int h, int w, zt_stream_t stream);


/* ---- optimizer step (zt_optim.hip): nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step (train.py:130-131)
 * over one flat bucket.  g is multiplied by gscale first (1/world_size after the data-parallel all-reduce).  partial: nblk floats
 * of workspace.  max_norm <= 0 disables clipping.  gnorm_out (optional): the pre-clip global norm. */
int zt_clip_adam_f32(float* p, const float* g, float* m, float* v, long long n, float* partial, int nblk, float gscale,
                     float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, long long step,
                     float* gnorm_out, zt_stream_t stream);


/* y[i] += alpha[0] * x[i]: `loss.backward()` (train.py:129) hands the step's parameter gradients (already computed next to the
 * loss by the fused plan) to the optimizer's flat bucket, scaled by the upstream gradient that autograd holds on the device. */
int zt_axpy_dev_f32(float* y, const float* x, const float* alpha, long long n, zt_stream_t stream);


/* ---- RAFT encoder stem in bf16 mode (zt_stem.hip): conv 7x7 / stride 2 / pad 3, 3 -> 64 (extractor.py:120, 168-170).
 * x: nhwc [N][H][W][8] bf16, channels 0..2 valid and 3..7 ZERO (as zt_raft_pack_input / zt_raft_pack_pair write them);
 * w: zt_repack_stem_weight_bf16 of the torch weight [64][3][7][7] -> [7][64][64] bf16; y: nhwc [N][H/2][W/2][ldy >= 64] bf16
 * = conv + bias, then ReLU when relu != 0 (context encoder: the frozen BatchNorm `norm1` folded into w / bias by the caller, so
 * relu(norm1(conv1(x))) of extractor.py:168-170 is this one launch; the feature encoder's InstanceNorm stays a separate pass). */
int zt_repack_stem_weight_bf16(const float* src, void* dst, zt_stream_t stream);
int zt_raft_stem_conv_bf16(const void* x, int N, int H, int W, const void* w, const float* bias, void* y, int ldy, int relu,
                           zt_stream_t stream);


/* ---- output side (zt_io.hip): predict.py:57-61 save_images and evals.py:83-85 PSNR on the device -------------------------
 * zt_quantize_u8_hwc: planar fp32 [3][H][W] in [0,1] -> interleaved uint8 [H][W][3] (what PIL / cv2 write);
 *   mode 0 = clip(x*255, 0, 255) truncated (predict.py / train.py save_images), mode 1 = np.round(x*255) (evals.py:83-84).
 * zt_sqdiff_u8_f32: out[0] = sum_i (round(a_i*255) - round(b_i*255))^2, exact 64-bit integer; PSNR = 10 log10(255^2 n / out).
 *   partial: nblk x 8 bytes of workspace. */
int zt_quantize_u8_hwc(const float* src, unsigned char* dst, int H, int W, int mode, zt_stream_t stream);
int zt_sqdiff_u8_f32(const float* a, const float* b, long long n, unsigned long long* partial, int nblk, unsigned long long* out,
                     zt_stream_t stream);


/* ---- input side (zt_ingest.hip): dataloader/multi_read_data.py:127-132 on the device -------------------------------------
 * The loader workers decode to interleaved uint8 RGB [H][W][3]; `im.resize((1920, 1080))` (PIL default filter for RGB = BICUBIC,
 * Pillow's 8-bit two-pass resampler) and `transforms.ToTensor()` run here, bit-identical to the host libraries.
 * zt_resample_u8_hwc: ONE pass of the resampler along `axis` (1 = horizontal: [Hi][Wi][3] -> [Hi][Wo][3], Ho == Hi;
 *   0 = vertical: [Hi][Wi][3] -> [Ho][Wi][3], Wo == Wi, 3*Wi % 4 == 0): out = clip8((2^21 + sum_k src[min + k] * coef[o][k]) >> 22);
 *   coef: int32 [out][ksize] = round(w * 2^22), bounds: int32 [out][2] = (first source index, tap count) -- device arrays, computed
 *   by the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do.  Horizontal first, then vertical (Resample.c).
 * zt_u8hwc_to_planar_f32: uint8 [H][W][3] -> planar fp32 [3][H][W], value lut256[byte] (lut256[k] = float(k) / 255.f: ToTensor). */
int zt_resample_u8_hwc(const unsigned char* src, unsigned char* dst, int Hi, int Wi, int Ho, int Wo, int axis, const int* coef,
                       const int* bounds, int ksize, zt_stream_t stream);
int zt_u8hwc_to_planar_f32(const unsigned char* src, float* dst, int H, int W, const float* lut256, zt_stream_t stream);


/* ---- hardware self-test probes (zt_probe.hip): pin the test emulator's model of gfx950 instructions to the chip ----- */
/* ds_read_b64_tr_b16 on a [16][64] image of 16-bit codes: out[lane*4+q]; bf16 MFMA 16x16x32: D[16][16] = A[16][32] B[32][16] */
int zt_probe_tr16(const unsigned short* img, unsigned short* out, int col0, zt_stream_t stream);
int zt_probe_mfma_bf16(const unsigned short* A, const unsigned short* B, float* D, zt_stream_t stream);


/* ---- bf16 throughput mode of the convolution family: x / x2 / aux / dz are bf16 nhwc (channel strides multiples of 8), w is
 * bf16 [tap][CoutP][ldk] (input channel fastest; the all-pairs correlation passes fmap2 itself, nhwc, as w), accumulation fp32.
 * out_mode: 0 = bf16 nhwc, 1 = fp32 planar (plane pitch ldy), 2 = fp32 nhwc.  Same geometries / act / alpha / epi as the fp32 entry. */
int zt_conv2d_nhwc_bf16(const void* x, const void* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin, const void* w,
                        int CoutP, int ldk, const float* bias, void* y, int ldy, int out_mode, int Cout, int KH, int KW, int stride,
                        int padH, int padW, int act, float alpha, const void* aux, int ldaux, int epi, zt_stream_t stream);
/* TWO independent stride-1 convolutions over the same map (square kernels KA / KB in {1, 3, 7}, pad K / 2; bf16 nhwc in and out,
 * y = act(conv + bias)) in ONE launch: RAFT's motion encoder runs convc1 (1x1, 324 -> 256) next to convf1 (7x7, 2 -> 128) and convc2
 * (3x3, 256 -> 192) next to convf2 (3x3, 128 -> 64) (update.py:89-94).  Same results as two zt_conv2d_nhwc_bf16 calls (which is what
 * it falls back to when the two problems do not share a launch shape). */
int zt_conv2d_pair_nhwc_bf16(const void* xA, int ldxA, int CinA, const void* wA, int CoutPA, int ldkA, const float* biasA, void* yA, int ldyA,
                             int CoutA, int KA, const void* xB, int ldxB, int CinB, const void* wB, int CoutPB, int ldkB, const float* biasB,
                             void* yB, int ldyB, int CoutB, int KB, int N, int H, int W, int act, zt_stream_t stream);
/* same, with the kernel variant pinned (tests / tuning): 0 auto, 1 persistent weight-stationary (stride 1, K in {1,3}, Cin <= 64, N == 1),
 * 2 tiled, 3 register-stationary persistent (3x3, bf16 nhwc output, act in {none, ReLU, LeakyReLU}, exactly 48 or 64 couts) */
int zt_conv2d_nhwc_bf16_variant(const void* x, const void* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin, const void* w,
                                int CoutP, int ldk, const float* bias, void* y, int ldy, int out_mode, int Cout, int KH, int KW, int stride,
                                int padH, int padW, int act, float alpha, const void* aux, int ldaux, int epi, int variant, zt_stream_t stream);
/* bf16 twin of zt_conv2d_nhwc_f32_ex (epi 4 / 5: bf16 nhwc outputs, tiled kernel), plus
 * epi 6: y = relu(act(conv + bias) + aux) -- the tail of a RAFT ResidualBlock whose eval-mode BatchNorm is folded into the
 *        conv's weights and bias (model/RAFT/extractor.py:45-56: `y = relu(norm2(conv2(y))); return relu(x + y)`) */
int zt_conv2d_nhwc_bf16_ex(const void* x, const void* x2, int csplit, int ldx, int ldx2, int N, int H, int W, int Cin, const void* w,
                           int CoutP, int ldk, const float* bias, void* y, int ldy, int out_mode, int Cout, int KH, int KW, int stride,
                           int padH, int padW, int act, float alpha, const void* aux, int ldaux, int epi, void* y2, int ldy2, int esplit,
                           zt_stream_t stream);
/* Enhancer block head (model.py:60-62): y = conv3x3(x) + bias (stride 1, pad 1, N == 1, bf16 nhwc) AND the train-mode
 * BatchNorm statistics of y in the same pass: stats [stats_blocks = 512][2][Cout] per-workgroup partial (sum, sum of squares) of
 * the stored values, zero-filled beyond the launch's workgroups -- feed to zt_norm_finalize_f32(partial = stats, nblk = 512).
 * The 64 -> 64 full-resolution layer accumulates them in the register-stationary kernel's store phase; any other geometry runs
 * the convolution followed by the statistics kernel (same output contract). */
int zt_conv3x3_bn_stats_bf16(const void* x, int ldx, int H, int W, int Cin, const void* w, int CoutP, int ldk, const float* bias,
                             void* y, int ldy, int Cout, float* stats, int stats_blocks, zt_stream_t stream);
/* Enhancer block backward (autograd of model.py:60-67): df = conv3x3^T(dz) + res (the data gradient with the residual add, wT = the
 * flipped / transposed weights, all bf16 nhwc with 64 channels, N == 1) AND, in the same pass, the BatchNorm-backward sums of the
 * block BELOW, whose output gradient df is: g = df * [bn_scale * zprev + bn_shift > 0] (its ReLU), stats [512][2][64] = per-workgroup
 * (sum g, sum g (zprev - bn_mean)), zero beyond the launch's workgroups -- feed to zt_bn_bwd_sums_centered_f32(nblk = 512).
 * Replaces zt_conv2d_nhwc_bf16 (epi 3) + zt_bn_bwd_reduce. */
int zt_conv3x3_dgrad_bn_sums_bf16(const void* dz, int lddz, int H, int W, const void* wT, int CoutP, int ldk, void* df, int lddf,
                                  const void* res, int ldres, const void* zprev, int ldz, const float* bn_scale, const float* bn_shift,
                                  const float* bn_mean, float* stats, int stats_blocks, zt_stream_t stream);
/* relu_mask (optional, thin-input 3x3 layer with 64 couts): dz is taken as dz * [relu_mask > 0], i.e. the backward of the ReLU
 * that follows the layer (model.py:55-56) is folded into the staging of dz instead of a separate masking pass */
int zt_conv2d_wgrad_nhwc_bf16(const void* x, int ldx, const void* dz, int lddz, int H, int W, int Cin, int Cout, int KH, int KW,
                              float* slab, size_t slab_bytes, float* grad_w, float* grad_b, int accumulate, const void* relu_mask,
                              int ldmask, zt_stream_t stream);
/* Deferred, batched form of the weight gradient for a whole backward pass (bf16 mode): every call of a layer APPENDS its
 * per-workgroup slabs (KH*KW*Cin16*Cout16 + Cout16 floats each, Cin16 / Cout16 = channel counts rounded up to 16) to that layer's slab region and reports how many it wrote
 * (*nslab_out, host int); zt_wgrad_reduce_multi_f32 then reduces up to 16 layers in ONE launch: layer i sums nslab[i] slabs at
 * slab[i] into grad_w[i] ([Cout][Cin][K][K]) and grad_b[i] (may be NULL), (+)= when accumulate.  All array arguments are host arrays. */
int zt_conv2d_wgrad_partial_bf16(const void* x, int ldx, const void* dz, int lddz, int H, int W, int Cin, int Cout, int KH, int KW,
                                 float* slab, size_t slab_bytes, const void* relu_mask, int ldmask, int* nslab_out, zt_stream_t stream);
/* Backward of Denoise_1/2's 1x1 output layer (model.py:27, 43) in one pass over its operands: dz[p][48] = (W^T dr[p]) * LeakyReLU'(a2[p])
 * (what zt_conv2d_nhwc_bf16 with the transposed weights wT [48][8] and epi 1 computes) AND the layer's weight / bias gradient slabs
 * ([48][16] + [16] floats each, *nslab_out of them appended at `slab`: feed to zt_wgrad_reduce_multi_f32 with Cin 48, Cout Cdr, K 1).
 * dr: nhwc bf16 [HW][8] with Cdr = 3 or 6 valid channels; a2 / dz: nhwc bf16, channel strides lda / lddz >= 48. */
int zt_thin1x1_bwd_bf16(const void* dr, int Cdr, const void* wT, const void* a2, int lda, void* dz, int lddz, int HW, float* slab,
                        size_t slab_bytes, int* nslab_out, zt_stream_t stream);
int zt_wgrad_reduce_multi_f32(int nseg, const void* const* slab, const int* nslab, const int* Cin, const int* Cout, const int* K,
                              void* const* grad_w, void* const* grad_b, int accumulate, zt_stream_t stream);
/* all weight repacks of a step in one launch (up to 24 entries; host arrays; square kernels K x K; same layouts as the single form) */
int zt_repack_conv_weights_bf16_multi(int n, const void* const* src, void* const* dst, const int* Cout, const int* Cin, const int* K,
                                      const int* CoutP, const int* ldk, const int* transpose_flip, zt_stream_t stream);
int zt_repack_conv_weight_bf16(const float* src, void* dst, int Cout, int Cin, int KH, int KW, int CoutP, int ldk, int co_off,
                               int transpose_flip, zt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
