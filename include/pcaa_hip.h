/*
 * pcaa_hip.h -- C ABI of the MI355X (gfx950) PCAA hot path.
 *
 * The reference (rmazzier/OpenSetGaitRecognition_PCAA) has no FFI: its operator
 * boundary is Python nn.Module.forward + autograd, below which sits PyTorch
 * ATen.  Each entry point here replaces the ATen work behind one reference call
 * site (cited per function as reference file:line).  The Python side
 * (opensetgaitrecognition_pcaa_amd/functional.py) binds them with ctypes and
 * wraps them in torch.autograd.Functions; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; nothing is
 *     allocated, freed or synchronised inside a call (graph-capture safe);
 *   - `stream` is a hipStream_t passed as void*;
 *   - return value: PCAA_OK, or an error code with a message retrievable from
 *     pcaa_last_error() (thread-local);
 *   - activations are ROW-MAJOR [rows, channels] ("point-major": one row per
 *     point / per (batch,time) step, channels fastest);
 *   - dtype arguments: PCAA_F32 or PCAA_BF16 (storage type of an activation
 *     tensor); parameters, statistics, losses and optimizer state are fp32
 *     (BatchNorm statistics accumulate in fp64).
 */
#ifndef PCAA_HIP_H
#define PCAA_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCAA_ABI_VERSION 16 /* pcaa_abi_version() of a library built from this header */

#define PCAA_OK 0
#define PCAA_ERR_INVALID_ARG 1
#define PCAA_ERR_LAUNCH 2

#define PCAA_F32 0
#define PCAA_BF16 1
#define PCAA_SPLIT_F16 2 /* the [hi | lo] 16-bit image of an fp32 tensor (pcaa_gemm_split3): accepted where a call says so */

/* operand storage order for pcaa_gemm */
#define PCAA_LAYOUT_KC 0 /* contraction index contiguous: A[m*ld + k], B[n*ld + k] */
#define PCAA_LAYOUT_RC 1 /* row index contiguous:         A[k*ld + m], B[k*ld + n] */

#define PCAA_ACT_NONE 0
#define PCAA_ACT_ELU 1

const char* pcaa_last_error(void);
int pcaa_abi_version(void);

/* ------------------------------------------------------------------ GEMM
 * C[M,N] (=|+=) A(M,K) . B(K,N) (+ bias[N]) on the MFMA pipe.
 *   math = PCAA_F32 : v_mfma_f32_32x32x2_f32 (exact fp32), any operand dtype/layout
 *   math = PCAA_BF16: v_mfma_f32_32x32x16_bf16, fp32 accumulate; operands KC
 * colstats != NULL: per-column sum and sum-of-squares of the bias-free
 *   accumulator over the M rows are added (fp64 atomics) into
 *   colstats[(tile_m % nrep)][2][N] -- the BatchNorm batch statistics of the
 *   layer this GEMM produces.
 * split_k > 1 or accumulate != 0: fp32 atomic accumulation into C (C must be
 *   fp32 and pre-initialised); bias is added by split 0 only.
 * Replaces: Conv2d(1x1) models.py:20-27, Conv1d via im2col models.py:59-68,
 *   Linear models.py:252-277, 346-371, 409-416, and their autograd backward.
 */
int pcaa_gemm(int math,
              const void* A, int a_dtype, int a_layout, long lda,
              const void* B, int b_dtype, int b_layout, long ldb,
              void* C, int c_dtype, long ldc,
              int M, int N, int K,
              const float* bias, double* colstats, int nrep,
              int split_k, int accumulate, void* stream);

/* Split-K without atomics (the long-K weight-gradient products): split s writes its partial
 * M x N product to slabs + s * slab_stride (fp32), pcaa_splitk_reduce sums the slabs into
 * out (=|+=).  pcaa_gemm_num_splits tells how many splits pcaa_gemm / pcaa_gemm_slabs will
 * actually run for a requested split_k (K is cut into multiples of the kernel's K step). */
int pcaa_gemm_num_splits(int math, int K, int split_k);
/* ABI 14: n <= 8 small products C_i[M_i, N_i] += A_i^T . B_i in ONE launch (the temporal block's six weight gradients
 * dW_l = dy_l^T . col_l, reference models.py:108-160 through autograd): A_i [K_i, M_i], B_i [K_i, N_i], C_i [M_i, N_i], all
 * fp32 with contiguous rows, 16-B aligned, M_i and N_i multiples of 4; exact-fp32 MFMA; C_i is accumulated into (atomics
 * over split_k[i] ranges of the contraction): the caller zeroes it. */
int pcaa_gemm_group_rc_f32(int n, const void* const* A, const void* const* B, void* const* C, const int* M,
                           const int* N, const int* K, const int* split_k, void* stream);
/* Which tile loop serves the bf16 / split-fp16 KC x KC products without K splits (whole 256 x 256 tiles, contraction
 * >= 320 deep): 1 (default; environment PCAA_GEMM_V2=0 to start with 0) = the 4-wave loop of round 4 (csrc/gemm_v2.h:
 * 128 x 128 wave tiles, both operands requested two K steps ahead, the request stream continuous across tiles), 0 = the
 * 8-wave loop of rounds 1-3.  Same operands, same epilogues, same results up to the summation order inside a
 * 64-deep step (none: both accumulate k in the same order); kept switchable for same-process A/B (tools/gemm_lab.py). */
int pcaa_gemm_v2_enable(int on);
int pcaa_gemm_slabs(int math,
                    const void* A, int a_dtype, int a_layout, long lda,
                    const void* B, int b_dtype, int b_layout, long ldb,
                    float* slabs, long slab_stride, int M, int N, int K, int split_k, void* stream);
/* dgrad of a PointNet layer fused with the BatchNorm+ELU backward of the layer BELOW it
 * (models.py:20-29 backward): da = dy[M,K] . Wt[N,K]^T never reaches memory; the epilogue reads that
 * layer's stored pre-activation y[M,N] and writes  dz = da * ELU'(y*scale+shift)  (bf16, same ld as y)
 * while adding {sum dz, sum dz*(y-mean)*rstd} per column into stats (as pcaa_bn_act_bwd_dz does in a
 * separate pass).  All operands bf16, M and N multiples of 256, K of 64.
 * x, xc, W1: must be NULL / 0 / NULL -- the variant of rounds 1-4 that rebuilt y of the first PointNet layer from the
 * points in the epilogue (never faster than the separate statistics pass) left with the 8-wave kernel in round 5; y is
 * required. */
/* Eval-mode PointNet layer in one launch (models.py:6-34 with BatchNorm2d in eval mode: a per-channel affine map
 * known before the product): out[M,N] bf16 = ELU(scale[n] * (A[M,K] . W[N,K]^T) + shift[n]), A and W bf16,
 * contraction contiguous; same shape rule as pcaa_gemm_dgrad_bn_supported.  pool_rows in {32, 64, 128}: the
 * activation is additionally averaged over groups of pool_rows consecutive rows (AvgPool2d((1,N)) over the
 * points of a frame, models.py:242-243) and out is fp32 [M/pool_rows, N] -- the [M,N] activation never
 * exists.  The inference path (inference_PCAA.py:196-231, BASELINE config[4]) uses both forms. */
int pcaa_gemm_affine_elu(const void* A, long lda, const void* W, long ldw, void* out, long ldo,
                         const float* scale, const float* shift, int M, int N, int K, int pool_rows,
                         void* stream);
int pcaa_gemm_dgrad_bn_supported(int M, int N, int K);
int pcaa_gemm_dgrad_bn(const void* dy, long lddy, const void* Wt, long ldw, const void* y, void* dz, long ld,
                       const float* scale, const float* shift, const float* mean, const float* rstd,
                       double* stats, int nrep, int M, int N, int K,
                       const float* x, int xc, const float* W1, void* stream);
/* dy = coef0*dz + coef1*y + coef2 (fp32 dz, y) written as its [hi | lo] fp16 image [rows, 2 ch]: the second half of the
 * BatchNorm backward behind pcaa_gemm_dgrad_bn_split3 */
int pcaa_bn_bwd_dy_split(const float* dz, const float* y, void* dy_img, const float* coef, long rows, int ch,
                         float img_scale, void* stream);
/* pcaa_gemm_dgrad_bn for the split-fp16 parity mode: dy_img [M, 2K], Wt_img [N, 2K] are [hi | lo] fp16 images, y and dz
 * fp32 [M, ld]; out_scale = 1 / (image scales). */
int pcaa_gemm_dgrad_bn_split3(const void* dy_img, long lddy, const void* Wt_img, long ldw, const float* y, float* dz,
                              long ld, const float* scale, const float* shift, const float* mean, const float* rstd,
                              double* stats, int nrep, int M, int N, int K, float out_scale, void* stream);
/* Kernel-exact timing of one LDS-DMA GEMM launch (bench.py's roofline figure): events made by
 * pcaa_timing_events_create and armed with pcaa_time_next_gemm ride on the NEXT such launch of the
 * calling thread (hipExtLaunchKernelGGL start/stop events: the timestamps of the kernel's own dispatch
 * packet, what rocprofv3 reports).  pcaa_timing_pending: 1 while armed events have not been consumed
 * (the call did not reach that kernel); pcaa_time_next_gemm(NULL, NULL) disarms.  pcaa_timing_elapsed_ms
 * after the stream has been synchronised. */
int pcaa_timing_events_create(void** start, void** stop);
int pcaa_timing_events_destroy(void* start, void* stop);
int pcaa_time_next_gemm(void* start, void* stop);
int pcaa_timing_pending(void);
int pcaa_timing_elapsed_ms(void* start, void* stop, float* ms);
int pcaa_splitk_reduce(const float* slabs, int nsplit, long slab_stride, long n, float* out,
                       int accumulate, void* stream);
/* same, for out[rows, ch], plus the BatchNorm column statistics of out (stats as in pcaa_gemm) */
int pcaa_splitk_reduce_stats(const float* slabs, int nsplit, long slab_stride, float* out,
                             double* stats, int nrep, long rows, int ch, void* stream);

/* First PointNet layer, Conv2d(C -> cout, 1x1) on the raw points x[P,C] (models.py:87-89):
 * y[P,cout] = x . W[cout,C]^T + bias, with the same BatchNorm statistics as pcaa_gemm;
 * and its weight gradient dW[cout,C] += dy[P,cout]^T . x (dW must be pre-initialised).
 * C <= 8; the contraction is too narrow for MFMA, these stream y / dy at HBM rate. */
int pcaa_pointnet_in_fwd(const float* x, int C, const float* W, const float* bias, void* y, int y_dtype,
                         long P, int cout, double* stats, int nrep, void* stream);
int pcaa_pointnet_in_wgrad(const void* dy, int dy_dtype, const float* x, int C, float* dW, long P,
                           int cout, void* stream);
/* Recompute path of the same layer (+ its BatchNorm2d + ELU, models.py:20-29): y costs C FMAs per
 * element, so it is never stored.  Forward: pcaa_pointnet_in_fwd with y == NULL (statistics only),
 * pcaa_bn_finalize, then  a = ELU((x.W^T)*scale + shift).  Backward, two passes over the incoming
 * gradient da only: statistics {sum dz, sum dz*yhat} with dz = da*ELU'(z) (then
 * pcaa_bn_bwd_finalize), and  dW += dy^T.x  with dy = coef0*dz + coef1*y + coef2 formed in registers. */
int pcaa_pointnet_in_apply(const float* x, int C, const float* W, const float* scale, const float* shift,
                           void* a, int a_dtype, long P, int cout, void* stream);
int pcaa_pointnet_in_bwd_stats(const void* da, int dtype, const float* x, int C, const float* W,
                               const float* scale, const float* shift, const float* mean, const float* rstd,
                               double* stats, int nrep, long P, int cout, void* stream);
/* One pass instead of the two (round 3): the weight gradient is linear in dy and y = x.W^T, so
 *   dW = c0 (.) (dz^T x) + c1 (.) (W . x^T x) + c2 (x) sum_p x.
 * pcaa_pointnet_in_bwd_onepass reduces the statistics AND G[nrep][cout,C] += dz^T.x (zero-initialised; workgroup b adds
 * into replica b % nrep: one shared image ran at the contended atomic rate) from one read of
 * da; pcaa_points_moments accumulates mom[pcaa_points_moments_size()] (zero-initialised fp64: x^T x at [k*8 + c], the
 * sums at [64 + c]); pcaa_pointnet_in_bwd_combine forms dW (=) from G, the moments and pcaa_bn_bwd_finalize's coef. */
int pcaa_pointnet_in_bwd_onepass(const void* da, int dtype, const float* x, int C, const float* W,
                                 const float* scale, const float* shift, const float* mean, const float* rstd,
                                 double* stats, int nrep, float* G, long P, int cout, const double* pivot_mom,
                                 double pivot_inv_count, void* stream);
int pcaa_points_moments_size(void);
int pcaa_points_moments(const float* x, int C, long P, double* mom, void* stream);
/* Round 4: G is accumulated against the points centred on a pivot, pivot[c] = (float)(pivot_mom[64 + c] *
 * pivot_inv_count) (pivot_mom NULL: no centring) -- normally the points' own moments and 1 / P; under SyncBN the
 * all-reduced moments and 1 / (global count), so that every rank uses the same pivot.  The combine takes the LOCAL
 * moments and point count P plus the same pivot source, and drops pivot (x) sum_p dy (zero per channel). */
int pcaa_pointnet_in_bwd_combine(const float* G, int nrep, const float* W, const double* mom, const float* coef,
                                 float* dW, int cout, int C, long P, const double* pivot_mom, double pivot_inv_count,
                                 void* stream);
/* The same moments give the layer's forward BatchNorm statistics without a pass over the points (sum y = W.sum x,
 * sum y^2 = W^T (x^T x) W per channel): arm the forward finalize on `mom` (pcaa_bn_tail_arm_fwd with stats = mom,
 * count = P, any non-null counter), then this call writes scale / shift / mean / rstd and the running statistics. */
int pcaa_pointnet_in_moment_stats(const double* mom, const float* W, int C, int cout, void* stream);
int pcaa_pointnet_in_bwd_wgrad(const void* da, int dtype, const float* x, int C, const float* W,
                               const float* scale, const float* shift, const float* coef, float* dW,
                               long P, int cout, int dz_is_pre /* da already is dz (pcaa_gemm_dgrad_bn) */,
                               void* stream);

/* bf16 shadow of an fp32 weight matrix src[R,C]: dst[R,C] and/or its transpose dst_t[C,R]
 * (either may be NULL).  Lets the bf16 GEMM stream both operands by LDS-DMA. */
int pcaa_cast_bf16(const float* src, void* dst, void* dst_t, int R, int C, void* stream);

/* ------------------------------------------------------------------ BatchNorm (+ELU) pieces
 * Training-mode BatchNorm2d/1d + ELU of PointNetModule (models.py:28-29) and
 * DilTempConv1d (models.py:71,77-78), split around the grid-wide statistics
 * dependency.  stats = [nrep][2][ch] fp64 (sum, sumsq of the bias-free linear
 * output); count = rows reduced over.
 */
int pcaa_bn_finalize(const double* stats, int nrep, long count, const float* lin_bias,
                     const float* gamma, const float* beta,
                     float* running_mean, float* running_var, long long* num_batches_tracked,
                     float momentum, float eps,
                     float* scale, float* shift, float* mean, float* rstd, int ch, void* stream);
/* eval mode: scale/shift from the running statistics.  NOTE: the pre-BN tensor y of these
 * layers is stored WITHOUT the linear layer's bias (it cancels in train-mode BatchNorm; here
 * it is folded into shift), hence lin_bias in both finalisers. */
int pcaa_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, const float* lin_bias, float eps, float* scale,
                        float* shift, int ch, void* stream);
/* a = ELU(y*scale + shift) */
int pcaa_bn_act_fwd(const void* y, void* a, int dtype, const float* scale, const float* shift,
                    long rows, int ch, void* stream);
/* pooled[g][c] = mean_{r in group g} ELU(y[g*group_rows + r][c]*scale[c] + shift[c])
 * (AvgPool2d over the N points, models.py:242-243,282; AvgPool1d over T, :249,284).
 * Training (e1, e2, mean, rstd non-NULL; all NULL in eval): also
 *   e1[g][c] = sum_r ELU'(z),  e2[g][c] = sum_r ELU'(z) * (y - mean[c]) * rstd[c]
 * from which pcaa_bn_pool_bwd_stats forms this layer's BatchNorm-backward statistics
 *   stats += { sum_g dpool*pool_scale*e1, sum_g dpool*pool_scale*e2 }
 * without re-reading y (the incoming gradient is constant over a group). */
int pcaa_bn_act_meanpool_fwd(const void* y, int dtype, const float* scale, const float* shift,
                             const float* mean, const float* rstd, float* pooled, float* e1, float* e2,
                             long groups, int group_rows, int ch, void* stream);
int pcaa_bn_pool_bwd_stats(const float* dpool, const float* e1, const float* e2, float pool_scale,
                           double* stats, int nrep, long groups, int ch, void* stream);
/* backward through ELU (+ the mean-pool broadcast when dpool != NULL):
 *   da = (dpool ? dpool[row / group_rows][c] * pool_scale : da[row][c])
 *   dz = da * ELU'(y*scale + shift);   stats += { sum dz, sum dz * yhat }  (fp64) */
int pcaa_bn_act_bwd_dz(const void* da, const float* dpool, int group_rows, float pool_scale,
                       const void* y, void* dz, int dtype,
                       const float* scale, const float* shift, const float* mean, const float* rstd,
                       double* stats, int nrep, long rows, int ch, void* stream);
/* coef[3][ch] such that dy = coef0*dz + coef1*y + coef2; dgamma, dbeta */
int pcaa_bn_bwd_finalize(const double* stats, int nrep, long count, const float* gamma,
                         const float* mean, const float* rstd,
                         float* coef, float* dgamma, float* dbeta, int ch, void* stream);
int pcaa_bn_bwd_dy(const void* dz, const void* y, void* dy, int dtype, const float* coef,
                   long rows, int ch, void* stream);
/* The finalize CARRIED BY THE PRODUCER of the statistics (round 3; csrc/bn_tail.h).  pcaa_bn_tail_arm_fwd / _bwd
 * note the arguments of pcaa_bn_finalize / pcaa_bn_bwd_finalize on the calling thread; the NEXT launch on that thread
 * that accumulates into exactly this `stats` buffer and can carry a tail (pcaa_gemm on the LDS-DMA path with colstats,
 * pcaa_gemm_dgrad_bn, pcaa_pointnet_in_fwd (statistics only), pcaa_pointnet_in_bwd_stats / _onepass, pcaa_bn_pool_bwd_stats,
 * pcaa_dtc_conv_fwd / _dgrad without K split) takes it: its last workgroup to finish (agent-scope arrival counter,
 * `counter`: one zero-initialised word, left at zero) writes the coefficients, and the separate finalize launch
 * -- 5-8 us on the critical path, 20 per train step -- is gone.  pcaa_bn_tail_pending() == 1 after the producer's
 * call means it did not take the tail: call pcaa_bn_tail_disarm() and the stand-alone finalize. */
int pcaa_bn_tail_arm_fwd(const double* stats, int nrep, long count, const float* lin_bias,
                         const float* gamma, const float* beta,
                         float* running_mean, float* running_var, long long* num_batches_tracked,
                         float momentum, float eps,
                         float* scale, float* shift, float* mean, float* rstd, int ch, unsigned* counter);
int pcaa_bn_tail_arm_bwd(const double* stats, int nrep, long count, const float* gamma,
                         const float* mean, const float* rstd,
                         float* coef, float* dgamma, float* dbeta, int ch, unsigned* counter);
int pcaa_bn_tail_pending(void);
int pcaa_bn_tail_disarm(void);
/* Two-pass form that never materialises dz: first pcaa_bn_act_bwd_dz with dz == NULL
 * (statistics only), then dy = coef0 * (da * ELU'(y*scale+shift)) + coef1 * y + coef2 here
 * (da / dpool as in pcaa_bn_act_bwd_dz; dy may alias da). */
int pcaa_bn_bwd_dy_fused(const void* da, const float* dpool, int group_rows, float pool_scale,
                         const void* y, void* dy, int dtype, const float* scale, const float* shift,
                         const float* coef, long rows, int ch, void* stream);

/* ------------------------------------------------------------------ small fp32 helpers */
/* y = act(y + bias[col])  (Linear bias + ELU of the decoder / MLP heads, models.py:373-382) */
int pcaa_bias_act(float* y, const float* bias, int act, long rows, int cols, void* stream);
/* dz = da * ELU'(z) computed from the saved output a = ELU(z) */
int pcaa_elu_bwd_from_out(const float* da, const float* a, float* dz, long n, void* stream);
/* out[c] = sum_r x[r][c]   (bias gradients) */
int pcaa_colsum(const float* x, float* out, long rows, int cols, void* stream);
/* out[0] = scale * sum(x[0..n)) -- deterministic single-block reduction */
int pcaa_sum(const float* x, long n, float scale, float* out, void* stream);
/* out[r] = scale * sum_c x[r][c] */
int pcaa_rowsum(const float* x, float* out, long rows, int cols, float scale, void* stream);
/* out[i] = x[i] * s[0]   (s is a DEVICE scalar: chain-rule scaling without a host sync) */
int pcaa_scale_by_device_scalar(const float* x, const float* s, float* out, long n, void* stream);
/* out[r][c] = x[r][c] * s[r] */
int pcaa_scale_rows(const float* x, const float* s, float* out, long rows, int cols, void* stream);
/* Prior sampling of the D-step (PCAA_ablation.py:904-931): onehot[b][k] = (k == gt[b]),
 * z[b] = z0[b] + means[gt[b]] */
int pcaa_prior_sample(const float* z0, const float* means, const long long* gt, int B, int K, int D,
                      float* z, float* onehot, void* stream);
/* dst[b,t,n,c] (contiguous) = src[b,c,t,n] (given element strides) */
int pcaa_pack_points(const float* src, long sb, long sc, long st, long sn,
                     float* dst, int B, int C, int T, int N, void* stream);

/* Batch assembly from a packed crop store resident in HBM (replaces the per-sample np.load + default collate
 * of MSRadarDataset.__getitem__ / DataLoader, datasets.py:466-479, PCAA_ablation.py:794-800):
 * dst[r] = src[idx[r]], rows of row_bytes bytes (multiple of 16).  An index outside [0, n_src_rows)
 * zero-fills its row and sets *err_flag (device int, may be NULL) to 1. */
int pcaa_gather_rows(const void* src, long n_src_rows, long row_bytes, const long long* idx, void* dst,
                     long n_rows, int* err_flag, void* stream);

/* causal dilated Conv1d (k=3) as a GEMM: col[(b,t)][ci*3+tap] = a[b][t-(2-tap)*d][ci] or 0
 * (models.py:59-68,75-76) and its adjoint */
int pcaa_dtc_im2col(const float* a, float* col, int B, int T, int Cin, int dilation, void* stream);
int pcaa_dtc_col2im(const float* dcol, float* da, int B, int T, int Cin, int dilation, void* stream);

/* ------------------------------------------------------------------ losses
 * SeqChamferLoss (utils.py:88-132) forward + analytic backward w.r.t. preds.
 * preds/gts/dpreds are logical [B,C,T,N] tensors with element strides.
 * frame_loss[b*T+t] = sum_j min_i P + sum_i min_j P;
 * dpreds (nullable) = d/dpreds of sum_{b,t} w_b * frame_loss, w_b = grad_scale
 *   * (grad_per_b ? grad_per_b[b] : 1). */
int pcaa_chamfer_fwd_bwd(const float* preds, long p_sb, long p_sc, long p_st, long p_sn,
                         const float* gts, long g_sb, long g_sc, long g_st, long g_sn,
                         int B, int T, int N, int C, float* frame_loss,
                         float* dpreds, long d_sb, long d_sc, long d_st, long d_sn,
                         float grad_scale, const float* grad_per_b, void* stream);
/* CrossEntropyLoss(mean) (PCAA_ablation.py:1008) + argmax(softmax) (:891-893).
 * loss, dlogits, preds are each nullable. dlogits = grad_scale*(softmax - onehot)/B.
 * err_flag (nullable, device int32[1]): set to 1 if a target is outside [0,K) -- torch raises there; the
 * row is then scored against class 0 so that nothing is read out of bounds. */
int pcaa_cross_entropy(const float* logits, const long long* target, int B, int K,
                       float* loss, float* dlogits, float grad_scale, long long* preds,
                       int* err_flag, void* stream);

/* ------------------------------------------------------------------ CGDiscriminator (models.py:405-421)
 * u = [x(32) ; label(K)] -> 64 ELU -> 32 ELU -> 1.  Parameters in PyTorch layout. */
int pcaa_disc_forward(const float* x, const float* label, int B, int K,
                      const float* W1, const float* b1, const float* W2, const float* b2,
                      const float* W3, const float* b3, float* out, void* stream);
/* first-order backward: gout[B] -> dx[B,32], dlabel[B,K], parameter grads (each nullable;
 * parameter grads are OVERWRITTEN) */
int pcaa_disc_backward(const float* x, const float* label, int B, int K,
                       const float* W1, const float* b1, const float* W2, const float* b2,
                       const float* W3, const float* b3, const float* gout,
                       float* dx, float* dlabel,
                       float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3,
                       float* workspace, size_t workspace_bytes, void* stream);
/* backward OF that input gradient (what torch.autograd.grad(D(interp), interp, create_graph=True) followed by
 * .backward() needs, PCAA_ablation.py:955-976): with g[B,32] = the dx of pcaa_disc_backward for the same
 * (x, label, gout) and gbar[B,32] the incoming gradient w.r.t. g, the gradients of sum_b <gbar_b, g_b> w.r.t.
 * x (dx2), label (dlabel2), gout (dgout [B]) and the parameters (db3 = 0).  Outputs nullable, OVERWRITTEN;
 * workspace as pcaa_disc_workspace_bytes. */
int pcaa_disc_backward_backward(const float* x, const float* label, int B, int K,
                                const float* W1, const float* b1, const float* W2, const float* b2,
                                const float* W3, const float* b3, const float* gout, const float* gbar,
                                float* dx2, float* dlabel2, float* dgout,
                                float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3,
                                float* workspace, size_t workspace_bytes, void* stream);
/* WGAN-GP critic step (PCAA_ablation.py:939-976): d_loss = mean D(fv) - mean D(z)
 * + gp_weight * mean (|dD/dx(z + alpha (fv - z))| - 1)^2 with the closed-form
 * second-order gradient.  losses[0]=d_loss, losses[1]=gp.
 * Parameter grads are OVERWRITTEN.  dz (or NULL): d(d_loss)/dz [B,32] -- ablation variant 1
 * (PCAA_ablation.py:28-378) learns the prior centroids, z = z0 + GaussianMeanLearner(onehot) (:170-186),
 * so the critic loss is differentiated w.r.t. z too (real pass + the penalty through the interpolates). */
size_t pcaa_disc_workspace_bytes(int B, int K);
int pcaa_disc_wgan_gp(const float* z, const float* fv, const float* label, const float* alphas,
                      int B, int K,
                      const float* W1, const float* b1, const float* W2, const float* b2,
                      const float* W3, const float* b3, float gp_weight, float* losses,
                      float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3,
                      float* dz, float* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ open-set scoring
 * joint_likelihood (inference_PCAA.py:129-136): lik[b] = (1/K) sum_k N(x_b; mu_k, I_D), float64,
 * evaluated as exp(log-pdf) like scipy.stats.multivariate_normal.pdf.
 * k-window vote (inference_PCAA.py:263-271): window w covers crops [w*k,(w+1)*k); known iff
 * #(lik > threshold) > k/2 -> most frequent predicted label over the encoder's n_classes outputs
 * (np.argmax(np.bincount(preds)): lowest label on ties), else n_labels (the "unknown" id = number of
 * distinct labels of the known test split, which may be smaller than n_classes). */
int pcaa_joint_likelihood(const float* x, const float* means, int B, int K, int D, double* lik,
                          void* stream);
int pcaa_kvote(const double* lik, const long long* preds, double threshold, int k, int n_labels,
               int n_classes, int n_windows, long long* out, void* stream);

/* ------------------------------------------------------------------ batch-skinny Linear layers
 * The CGDecoder's Linear stack (models.py CGDecoder: nn.Linear(32+K, S/16) ... nn.Linear(S/2, S),
 * called from PCAA_ablation.py train_variant4) with M = batch <= 64 rows: each product is one
 * pass over the fp32 weight matrix W[N][K] (N = out_features, K = in_features, ldw elements
 * between rows), streamed from HBM straight into bf16 MFMA fragments (fp32 accumulate).
 *   fwd  : y[M][N]  = act(x[M][K] . W^T + bias)
 *   dgrad: dx[M][K] (=|+=) (dz[M][N] . W) * ELU'(a_prev)   (a_prev = ELU output of the layer
 *          below, NULL: no factor) -- i.e. the gradient w.r.t. that layer's pre-activation
 *   wgrad: dW[N][K] = dz^T . x
 * fwd/dgrad split the contraction into `nsplit` slabs in `ws` (>= nsplit*M*N resp. nsplit*M*K
 * floats) and reduce them deterministically; nsplit must come from pcaa_skinny_splits (kind 0
 * fwd, 1 dgrad; 2 / 3: the _exact forms) -- the deepest split whose grid still fits the chip's resident workgroups of
 * that kernel in ONE round.  pcaa_skinny_supported: M <= 64, N and K multiples of 64 and >= 128. */
int pcaa_skinny_supported(int M, int N, int K);
int pcaa_skinny_splits(int kind, int M, int N, int K);
int pcaa_skinny_linear_fwd(const float* x, long ldx, const float* W, long ldw, const float* bias, int act,
                           float* y, float* ws, long ws_floats, int M, int N, int K, int nsplit,
                           void* stream);
int pcaa_skinny_linear_dgrad(const float* dz, long lddz, const float* W, long ldw, float* dx,
                             const float* a_prev, int accumulate, float* ws, long ws_floats, int M, int N,
                             int K, int nsplit, void* stream);
int pcaa_skinny_linear_wgrad(const float* dz, long lddz, const float* x, long ldx, float* dW, long lddw,
                             int M, int N, int K, void* stream);
/* The same three passes with fp32 products (v_mfma_f32_32x32x2_f32 on the unrounded operands): the decoder of the parity
 * modes ("fp32", "fp16x3"), which rounds 1-2 ran on the 128x128-tile fp32 GEMM.  Same arguments; the dgrad needs W 8-B
 * aligned with an even leading dimension. */
int pcaa_skinny_linear_fwd_exact(const float* x, long ldx, const float* W, long ldw, const float* bias, int act,
                                 float* y, float* ws, long ws_floats, int M, int N, int K, int nsplit, void* stream);
int pcaa_skinny_linear_dgrad_exact(const float* dz, long lddz, const float* W, long ldw, float* dx,
                                   const float* a_prev, int accumulate, float* ws, long ws_floats, int M, int N,
                                   int K, int nsplit, void* stream);
int pcaa_skinny_linear_wgrad_exact(const float* dz, long lddz, const float* x, long ldx, float* dW, long lddw,
                                   int M, int N, int K, void* stream);
int pcaa_skinny_linear_wgrad_adam_exact(const float* dz, long lddz, const float* x, long ldx, float* W,
                                        float* exp_avg, float* exp_avg_sq, long ldw, int M, int N, int K,
                                        float beta1, float beta2, float eps, float grad_scale,
                                        const float* coef_dev, void* stream);   /* = pcaa_skinny_linear_wgrad_adam, fp32 products */
/* The same product written as bf16 (dW_bf16 [N, lddw] bf16, lddw even): the data-parallel step with bf16 gradient
 * buckets produces the gradient in the form it crosses the wire in (no fp32 copy, no cast pass). */
int pcaa_skinny_linear_wgrad_bf16(const float* dz, long lddz, const float* x, long ldx, void* dW_bf16, long lddw,
                                  int M, int N, int K, void* stream);
/* The same product fused with optimizer_G's Adam update of that weight (PCAA_ablation.py:1018-1021: backward,
 * then optimizer_G.step()): dW[N,K] = dz^T x stays in registers, W / exp_avg / exp_avg_sq [N,ldw] are read and
 * written in place -- torch.optim.Adam's update with the step-dependent scalars from coef_dev (pcaa_adam_advance),
 * bit-identical to pcaa_skinny_linear_wgrad followed by pcaa_adam_step_dev on the same buffers.  W must not be
 * read by anything still in flight (the layer's dgrad).  Single-process training only. */
int pcaa_skinny_linear_wgrad_adam(const float* dz, long lddz, const float* x, long ldx, float* W,
                                  float* exp_avg, float* exp_avg_sq, long ldw, int M, int N, int K,
                                  float beta1, float beta2, float eps, float grad_scale,
                                  const float* coef_dev, void* stream);
/* ABI 15 (round 5): the same fused update from GATHERED rows -- the data-parallel step's exchange for the batch-skinny
 * decoder layers.  dz [M, lddz], x [M, ldx] hold the rows of ALL ranks stacked (M = world * B <= 512, all-gathered: 4 M
 * (N + K) bytes instead of the gradient's 4 N K); every rank forms the global gradient dz^T x in registers and applies
 * the identical Adam update (grad_scale = 1 / world: the reference's mean over the global batch, PCAA_ablation.py:1008-
 * 1021 at BATCH_SIZE = world * B).  bf16 products, fp32 accumulation over the rows in ascending order; M <= 64 gives the
 * bits of pcaa_skinny_linear_wgrad_adam. */
int pcaa_skinny_linear_wgrad_adam_rows(const float* dz, long lddz, const float* x, long ldx, float* W,
                                       float* exp_avg, float* exp_avg_sq, long ldw, int M, int N, int K,
                                       float beta1, float beta2, float eps, float grad_scale,
                                       const float* coef_dev, int rows_alloc, void* stream);
/* (x is read in whole 64-row chunks: rows_alloc = the rows x is allocated for, >= the next of 64 / 128 / 256 / 512 above
 * M; the rows past M must hold finite values -- they meet zeros) */
/* ABI 16 (round 6): the gathered update from PACKED operands -- what the data-parallel step exchanges since round 6.
 * The reference has no multi-GPU path; this stands where a torch DDP port of PCAA_ablation.py:1018-1021 would all-reduce
 * the decoder's gradient.  Every rank packs the two weight-gradient operands of a batch-skinny layer -- dz [rows, N] and
 * x [rows, K], rows <= 64 -- into ONE chunk P[(N + K)][64] bf16: transposed (one 128-B row per column of dz, then per
 * column of x), rounded to nearest even (the rounding the weight-gradient kernels apply in registers), zero behind `rows`.
 * The ranks' chunks concatenate (one all-gather per layer, 2 (N + K) 64 bytes per rank); pcaa_skinny_linear_wgrad_adam_t16
 * contracts over `chunks` <= 8 of them (chunk c at packed + c * chunk_stride elements, chunk_stride >=
 * pcaa_packed_chunk_elems(N, K)) and applies Adam in place: W <- Adam(W, grad_scale * sum_c dz_c^T x_c), 24 B per
 * parameter, bf16 products, fp32 accumulation.  Supersedes pcaa_skinny_linear_wgrad_adam_rows in the trainer (that entry
 * takes fp32 row-major operands and needs no pack; it stays for callers that hold such operands). */
long pcaa_packed_chunk_elems(int N, int K);
int pcaa_pack_rows_t16(const float* dz, long lddz, int N, const float* x, long ldx, int K, int rows, void* chunk_bf16,
                       void* stream);
int pcaa_skinny_linear_wgrad_adam_t16(const void* packed_bf16, long chunk_stride, int chunks, float* W, float* exp_avg,
                                      float* exp_avg_sq, long ldw, int N, int K, float beta1, float beta2, float eps,
                                      float grad_scale, const float* coef_dev, void* stream);
/* ABI 14: the bf16 IMAGE of a decoder weight (W16 [N, ldw] bf16: every element = the weight rounded to nearest even, the
 * conversion pcaa_skinny_linear_fwd / _dgrad apply in registers -- results are bit-identical).  _fwd_w16 / _dgrad_w16
 * stream the image instead of the fp32 matrix: half the bytes of the two passes that sit on the step's critical path.
 * The caller owns the image: train.PCAATrainer rebuilds it with pcaa_cast_bf16 behind each fused update (side stream) and
 * whenever the weight's version counter moved.  (Writing the image from inside pcaa_skinny_linear_wgrad_adam was
 * measured and dropped: 64-B partial lines from different CUs doubled that kernel's time.) */
int pcaa_skinny_linear_fwd_w16(const float* x, long ldx, const void* W16, long ldw, const float* bias, int act,
                               float* y, float* ws, long ws_floats, int M, int N, int K, int nsplit, void* stream);
int pcaa_skinny_linear_dgrad_w16(const float* dz, long lddz, const void* W16, long ldw, float* dx,
                                 const float* a_prev, int accumulate, float* ws, long ws_floats, int M, int N,
                                 int K, int nsplit, void* stream);

/* ------------------------------------------------------------------ temporal block, fused forward
 * One DilTempConv1d layer (models.py:37-79) in one launch: implicit im2col of the causal dilated
 * convolution, the previous layer's BatchNorm+ELU applied while its bias-free output `src` is staged
 * (scale/shift null: src is the block input, used as is), y[B*T,cout] = bias-free pre-BN output,
 * BatchNorm statistics of y (fp64 sums into stats[nrep][2][cout], as pcaa_gemm's colstats) and,
 * if col != null, the im2col matrix col[B*T, cin*3] (layout of pcaa_dtc_im2col) that the weight
 * gradient contracts with.  W = conv1d.weight viewed [cout, cin*3].  Exact-fp32 MFMA.
 * ksplit > 1 cuts the input channels over workgroups: split z writes its partial product to
 * y + z*slab_stride (y then holds ksplit slabs, stats must be null) and pcaa_splitk_reduce_stats
 * finishes the sum and the statistics.
 * pcaa_dtc_conv_supported: T <= 32, cin % 4 == 0, cout % 16 == 0. */
int pcaa_dtc_conv_supported(int T, int cin, int cout);
int pcaa_dtc_conv_ksplit(int B, int cin, int cout);   /* the ksplit to pass: >= cin/256, 8 for few long tiles */
int pcaa_dtc_conv_fwd(const float* src, const float* scale, const float* shift, const float* W, float* y,
                      float* col, double* stats, int nrep, int B, int T, int cin, int cout, int dilation,
                      int ksplit, long slab_stride, void* stream);
/* the bf16 throughput mode's variant (round 4): same arguments, same results up to bf16 rounding of the operands -- the
 * contraction on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), 128 output channels per workgroup; y, col, statistics fp32 */
int pcaa_dtc_conv_fwd_bf16(const float* src, const float* scale, const float* shift, const float* W, float* y,
                      float* col, double* stats, int nrep, int B, int T, int cin, int cout, int dilation,
                      int ksplit, long slab_stride, void* stream);

/* The adjoint w.r.t. the layer input in one launch (replaces dcol = dy . W on the im2col layout followed by
 * pcaa_dtc_col2im): da[(b,t)][ci] = sum_{co,tap} dy[b][t+(2-tap)*d][co] * W[co][ci][tap].
 *  - dy given, or formed on load from this layer's dz, y and the coefficients of pcaa_bn_bwd_finalize
 *    (dy = coef0*dz + coef1*y + coef2: the second half of the BatchNorm backward, no separate pass); dy_out
 *    (optional) receives the staged dy for the weight gradient;
 *  - ep_stats given: `out` is dz of the layer BELOW, da * ELU'(ep_y*ep_scale + ep_shift), and its BatchNorm-
 *    backward statistics {sum dz, sum dz*(ep_y-ep_mean)*ep_rstd} are accumulated into ep_stats[nrep][2][cin]
 *    (the first half of that layer's backward, no separate pass); needs ksplit == 1.
 * ksplit > 1 (cout > 512) writes slabs for pcaa_splitk_reduce, as the forward. */
int pcaa_dtc_conv_dgrad_ksplit(int B, int cin, int cout);
int pcaa_dtc_conv_dgrad(const float* dy, const float* dz, const float* y, const float* coef, float* dy_out,
                        const float* W, float* out, const float* ep_y, const float* ep_scale,
                        const float* ep_shift, const float* ep_mean, const float* ep_rstd, double* ep_stats,
                        int nrep, int B, int T, int cin, int cout, int dilation, int ksplit, long slab_stride,
                        void* stream);
int pcaa_dtc_conv_dgrad_bf16(const float* dy, const float* dz, const float* y, const float* coef, float* dy_out,
                        const float* W, float* out, const float* ep_y, const float* ep_scale,
                        const float* ep_shift, const float* ep_mean, const float* ep_rstd, double* ep_stats,
                        int nrep, int B, int T, int cin, int cout, int dilation, int ksplit, long slab_stride,
                        void* stream);      /* likewise, the adjoint */

/* ------------------------------------------------------------------ MLP heads, fused
 * CGEncoder's MLP_sup1 / MLP_head / MLP_sup2 (models.py:252-277, applied at :285-292) and the
 * decoder projection head Sequential(Linear(32,64), ELU) (PCAA_ablation.py:778-781) as one forward
 * and one backward launch (fp32 FMAs, exact in both precision modes):
 *   sup_fv = ELU(W1 x4 + b1)  [B,32];   h = ELU(Wh sup_fv + bh)  [B,16]   (Wh null: no projection head)
 *   logits = ELU(W2 (h | sup_fv) + b2)  [B,K];   hproj = ELU(Wg sup_fv + bg)  [B,64]   (Wg null: skipped)
 * Backward: d_logits / d_hproj are gradients w.r.t. those (post-ELU) outputs, d_sup is whatever else
 * arrives at sup_fv (critic, decoder without head); any of the three may be null.  All weight / bias
 * gradients and dx4 [B,512] are WRITTEN (not accumulated).  pcaa_heads_supported: widths
 * 512/32/(16|0)/(64|0); the backward keeps all rows in LDS: B <= 64, K <= 8. */
int pcaa_heads_supported(int B, int K, int d_in, int d_sup, int d_head, int d_proj, int backward);
int pcaa_heads_fwd(const float* x4, const float* W1, const float* b1, const float* Wh, const float* bh,
                   const float* W2, const float* b2, const float* Wg, const float* bg, float* sup_fv,
                   float* h, float* logits, float* hproj, int B, int K, void* stream);
int pcaa_heads_bwd(const float* x4, const float* sup_fv, const float* h, const float* logits,
                   const float* hproj, const float* W1, const float* Wh, const float* W2,
                   const float* Wg, const float* d_logits, const float* d_sup, const float* d_hproj,
                   float* dW1, float* db1, float* dWh, float* dbh, float* dW2, float* db2, float* dWg,
                   float* dbg, float* dx4, int B, int K, void* stream);

/* ------------------------------------------------------------------ split-operand parity mode (round 3, "fp16x3")
 * A parity-grade product without the 1/16-rate fp32 MFMA: an fp32 operand e is kept as hi = fp16(s e),
 * lo = fp16(s e - hi) -- its "[hi | lo] image": for a row-major [rows, ch] tensor a 16-bit [rows, 2 ch] with lo at
 * column ch + c (the same bytes as the fp32 tensor); s (img_scale) is a power of two that keeps the tensor in fp16's
 * normal range (activations 1, weights 2^8, gradients 2^16) -- and the product is hi.hi + lo.hi + hi.lo on the f16
 * MFMA pipe (the bf16 rate), fp32 accumulate, times out_scale = 1 / (s_A s_B): 22 mantissa bits per operand.
 * pcaa_gemm_split3: C[M,N] fp32 = A.B^T from images; layout KC: A [M, 2K], B [N, 2K]; RC: A [K, 2M], B [K, 2N]
 * (contraction over the rows: the weight gradient); colstats as pcaa_gemm.  pcaa_gemm_slabs_split3: slab split-K
 * (pcaa_gemm_split3_num_splits slabs, reduce with pcaa_splitk_reduce).  Whole 256x256 tiles, K % 64 == 0.
 * Producers of images: pcaa_split_f16 (any fp32 matrix, optionally transposed), pcaa_bn_act_fwd_split,
 * pcaa_bn_bwd_dy_fused_split, pcaa_pointnet_in_apply with a_dtype = PCAA_SPLIT_F16 (img_scale 1). */
int pcaa_gemm_split3_supported(int M, int N, int K);
int pcaa_gemm_split3(const void* A, const void* B, int layout, long lda, long ldb, float* C, long ldc, int M,
                     int N, int K, double* colstats, int nrep, float out_scale, void* stream);
int pcaa_gemm_split3_num_splits(int K, int split_k);
int pcaa_gemm_slabs_split3(const void* A, const void* B, int layout, long lda, long ldb, float* slabs,
                           long slab_stride, int M, int N, int K, int split_k, float out_scale, void* stream);
int pcaa_split_f16(const float* src, void* dst_img, long rows, int ch, int transpose, float img_scale, void* stream);
/* Range guard of the images (round 4): fp16 holds |x| <= 65504 while the fp32 tensors of the reference have no limit.
 * Every image producer saturates a scaled value that leaves that range (finite hi, lo = 0) and sets *dev_flag
 * (device int32, registered per calling thread; NULL = no flag) to 1; NaN inputs set it too and stay NaN in the image.  The caller reads the
 * flag when it next synchronises (PCAATrainer.check()) and must not trust that step's products. */
int pcaa_set_range_flag(int* dev_flag);
int pcaa_bn_act_fwd_split(const float* y, void* a_img, const float* scale, const float* shift, long rows,
                          int ch, float img_scale, void* stream);
int pcaa_bn_bwd_dy_fused_split(const float* da, const float* dpool, int group_rows, float pool_scale,
                               const float* y, void* dy_img, const float* scale, const float* shift,
                               const float* coef, long rows, int ch, float img_scale, void* stream);

/* ------------------------------------------------------------------ OR-CED baseline heads (round 3)
 * ORCEDEncoder's three Linear heads and the reparametrisation (reference models.py:489-505):
 *   mu = x4.Wmu^T + bmu, logvar = x4.Wlv^T + blv, sup_fv = mu + eps*exp(0.5*logvar), logits = sup_fv.Wc^T + bc
 * (x4 [B,d_in], eps [B,d_lat]: the caller's randn draw), their backward (any of d_logits / d_sup / d_mu / d_logvar may
 * be NULL = zero; ws: 2*B*d_lat floats of scratch; dx4 may be NULL), and CG_kl_divergence (utils.py:72-85):
 *   loss = mean_b( -0.5 sum_j (1 + logvar - (mu - mu_k)^2 - exp(logvar)) ),  gradients scaled by gscale / B. */
int pcaa_orced_heads_supported(int B, int K, int d_in, int d_lat);
int pcaa_orced_heads_fwd(const float* x4, const float* Wmu, const float* bmu, const float* Wlv, const float* blv,
                         const float* eps, const float* Wc, const float* bc, float* mu, float* logvar, float* sup_fv,
                         float* logits, int B, int K, int d_in, int d_lat, void* stream);
int pcaa_orced_heads_bwd(const float* x4, const float* eps, const float* logvar, const float* sup_fv,
                         const float* Wmu, const float* Wlv, const float* Wc, const float* d_logits,
                         const float* d_sup, const float* d_mu, const float* d_logvar, float* ws, float* dWmu,
                         float* dbmu, float* dWlv, float* dblv, float* dWc, float* dbc, float* dx4, int B, int K,
                         int d_in, int d_lat, void* stream);
int pcaa_orced_kl(const float* mu, const float* logvar, const float* mu_k, float* loss, float* d_mu,
                  float* d_logvar, float* d_muk, float gscale, int B, int d_lat, void* stream);

/* ------------------------------------------------------------------ optimizer
 * torch.optim.Adam (no weight decay, no amsgrad; PCAA_ablation.py:820-833) on a
 * flat fp32 buffer. `step` is the 1-based step count after this update.  max_blocks (0 =
 * default, fill the device) caps the grid: <= 1024 selects the 4-quads-per-thread kernel that
 * saturates HBM from 1-2 workgroups per CU, for updates that run beside other kernels. */
int pcaa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                   float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                   int max_blocks, void* stream);
/* The same update with the step count kept ON THE DEVICE, so that a captured hipGraph of the
 * train step replays without per-step host arguments: pcaa_adam_advance does
 * `*step_dev += 1; coef_dev[0] = lr / (1 - beta1^step); coef_dev[1] = 1 / sqrt(1 - beta2^step)`
 * (fp64, the arithmetic of pcaa_adam_step) once per optimizer step; pcaa_adam_step_dev applies the
 * update to any sub-range of the flat buffer with those two scalars read from coef_dev. */
int pcaa_adam_advance(int* step_dev, float* coef_dev, float lr, float beta1, float beta2, void* stream);
int pcaa_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                       float beta1, float beta2, float eps, const float* coef_dev, float grad_scale,
                       int max_blocks, void* stream);
/* pcaa_adam_step_dev with the gradient given as bf16 (n elements, 8-B aligned): the data-parallel step with bf16
 * gradient buckets hands the reduced bucket to the optimizer as it came off the wire, without widening it into the
 * fp32 gradient buffer first (one pass over the decoder's 157 M gradients less, and half the bytes of Adam's read). */
int pcaa_adam_step_dev_g16(float* param, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, long n,
                           float beta1, float beta2, float eps, const float* coef_dev, float grad_scale,
                           int max_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCAA_HIP_H */
