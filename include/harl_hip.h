/*
 * harl_hip.h -- C ABI of libharl_hip.so: gfx950 (MI355X) kernels for HARL's on-policy
 * sequential-update path (HAPPO / V-critic).
 *
 * The reference (PKU-MARL/HARL) is 100 % Python and has no FFI for this path; every entry
 * point below replaces a Python/NumPy/ATen *op sequence* of the reference, cited per function
 * (paths relative to the reference root).  The Python classes in harl_amd/ mirror the
 * reference's Runner / Algorithm / Buffer API and are the only callers (ctypes, see
 * INTEGRATION.md for the binding a HARL maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (memory owned by the caller, normally a torch tensor);
 *   - no allocation, no host synchronisation inside; work is enqueued on `stream`
 *     (a hipStream_t passed as void*), so calls are safe under hipGraph capture;
 *   - return value: 0 on success, negative on error (harl_last_error() gives the text);
 *   - all floating data is fp32; "ATL" = activation tile layout: for a width-H
 *     activation, slab g (32 consecutive samples) is stored as [H/8][64 lanes][4] floats;
 *   - `idx` arguments are optional int64 row-gather arrays (NULL = identity): sample j of the
 *     minibatch is row idx[j] of the flattened [T*N, .] buffer -- exactly the arrays the
 *     reference draws with torch.randperm (on_policy_actor_buffer.py:131-135).
 */
#ifndef HARL_HIP_H
#define HARL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HARL_PS_STRIDE 48 /* floats per row of the per-workgroup partial-scalar tables */
#define HARL_DHEAD_LD 32  /* row stride of the head-gradient matrix */
#define HARL_MD_MAX_HEADS 8  /* MultiDiscrete: entries of nvec */
#define HARL_MD_MAX_GROUPS 4 /* MultiDiscrete: logits images (<= 128 logits each) */
/* caller-owned device scratch of the two entry points that combine per-workgroup partial sums across the grid (8-byte aligned;
 * ONE block per stream that may have such a launch in flight; the library itself never allocates device memory -- the only
 * exception is harl_comm_create, which must export its buffer through hipIpc): */
#define HARL_MM_SCRATCH_BYTES 24640 /* harl_masked_moments: 1024 x 3 doubles + a ticket word */
#define HARL_CG_SCRATCH_BYTES 1088  /* harl_trpo_cg_step: barrier words + 2 x 64 doubles; zero-filled ONCE by the caller */

int harl_version(void);
const char *harl_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * GAE / returns reverse scan + fused advantages.
 * Replaces OnPolicyCriticBufferEP.compute_returns (common/buffers/on_policy_critic_buffer_ep.py:97-200,
 * all 8 branches; FP twin on_policy_critic_buffer_fp.py:107-210) and the advantage subtraction of
 * OnPolicyHARunner.train (runners/on_policy_ha_runner.py:26-33).
 *   rewards[T,ncols] value_preds[T+1,ncols] masks/bad_masks[T+1,ncols] next_value[ncols]
 *   vn_stats: {running_mean, running_mean_sq, debiasing_term} or NULL (no ValueNorm)
 *   returns[T+1,ncols] (out)   advantages[T,ncols] (out, may be NULL)
 * Same fp32 operation order as the reference (no FMA contraction): bit-identical results.
 * fp_order=1 selects the FP buffer's `gamma*lambda*gae*mask` product order (_fp.py:130).
 */
int harl_gae_returns(const float *rewards, float *value_preds, const float *masks, const float *bad_masks,
                     const float *next_value, const float *vn_stats, float *returns, float *advantages,
                     int T, int ncols, float gamma, float gamma_lambda, int use_gae,
                     int use_proper_time_limits, int fp_order, void *stream);

/* Masked moments of the advantages: {sum x, sum x^2, count} over entries with active != 0, fp64.
 * Replaces the NaN trick + np.nanmean/np.nanstd of HAPPO.train (algorithms/actors/happo.py:122-127).
 * out3 (double[3]) is ACCUMULATED into (zero it first); all-reduce it across ranks when sharded.
 * scratch: HARL_MM_SCRATCH_BYTES of caller-owned device memory (see the defines above). */
int harl_masked_moments(const float *x, const float *active, long n, double *out3, void *scratch, void *stream);

/* Rollout-side row arithmetic on the head outputs: everything StochasticPolicy.forward / evaluate_actions do around the random
 * draw (harl/models/base/act.py:45-157, distributions.py:31-103; the draw itself is torch's device generator, as in the
 * reference).  One thread per row; every output pointer may be NULL.
 *   kind 0 (DiagGaussian, head = mean [M, act_dim]): actions = mean + sigma * noise (noise NULL: the mode), logp [M, act_dim]
 *          = log N(a; mean, sigma), ent_rows [M] = sum_d (0.5 + 0.5 log 2pi + log sigma_d), sigma_out [act_dim]
 *          (sigma = sigmoid(log_std / std_x_coef) * std_y_coef)
 *   kind 1 (Categorical; n_heads heads side by side in head [M, act_dim], head_off = n_heads + 1 offsets or NULL for one head;
 *          head = normalised, availability-masked logits): probs = exp(head), argmax_out [M, n_heads] = first largest logit per
 *          head, logp [M, n_heads] (sum_heads: [M, 1], act.py:56-73) = logit of `actions` [M, n_heads] (float indices; NULL: of
 *          the argmax), ent_rows = -sum clamp(logit) p over all heads */
int harl_dist_rows(const float *head, long M, int act_dim, int kind, const float *log_std, float std_x_coef, float std_y_coef,
                   const float *noise, float *actions, const int *head_off, int n_heads, int sum_heads, float *logp,
                   float *probs, float *argmax_out, float *ent_rows, float *sigma_out, void *stream);
/* mean_out[0] = (float)(moments3[0] / moments3[2]) of a harl_masked_moments triple: the (active-mask-weighted) mean of the
 * entropy rows that evaluate_actions returns (act.py:104-157) */
int harl_moments_mean(const double *moments3, float *mean_out, void *stream);
/* Row tables of a recurrent minibatch in ONE launch (the chunk slicing of on_policy_actor_buffer.py:255-322 /
 * on_policy_critic_buffer_ep.py:285-381 and the naive whole-column sampling :180-221): sequence j < m covers source rows
 * first[j] + l * stride, l < L, of the t-major flattened buffers; sequences m .. m_pad-1 (padding to the 32-sample slab) replay
 * sequence 0.  idx[l * m_pad + j] = that row (int64), valid_idx[l * m + j] the same without the padding (may be NULL),
 * mask_rows = masks_src[idx], h0[j] = h0_src[first[j]] (H floats per row). */
int harl_build_seq(const int64_t *first, int m, int m_pad, int L, long stride, const float *masks_src, const float *h0_src,
                   int H, int64_t *idx, int64_t *valid_idx, float *mask_rows, float *h0, void *stream);
/* Measurement aid (bench.py; nothing in the reference corresponds to it): one lane waits `ticks` periods of the constant
 * 100 MHz counter and writes {shader cycles elapsed, ticks elapsed} to cycles_ticks[0..1] (int64, device) -- the shader clock
 * under whatever runs next to it on other streams.  ticks in (0, 1e8]. */
int harl_clock_probe(long long *cycles_ticks, long ticks, void *stream);
/* adv_out = (adv - mean) / (std + 1e-5) with mean/std from `moments3` (happo.py:127). */
int harl_adv_normalize(const float *adv, const double *moments3, float *adv_out, long n, void *stream);

/* factor[i] *= agg_d exp(new_logp[i,d] - old_logp[i,d]), agg = prod (0) | mean (1).
 * Replaces runners/on_policy_ha_runner.py:116-124. */
int harl_factor_update(float *factor, const float *new_logp, const float *old_logp, long n, int act_dim,
                       int agg_mean, void *stream);

/* ValueNorm (common/valuenorm.py:47-64): sums2 (double[2]) += {sum R, sum R^2} over the minibatch
 * (rows idx[0..m) of `returns`, or 0..m); then harl_valuenorm_apply does the debiased EMA update of
 * vn_stats = {running_mean, running_mean_sq, debiasing_term} with the (global) count. */
int harl_sum_sumsq(const float *x, const int64_t *idx, long m, double *sums2, void *stream);
int harl_valuenorm_apply(float *vn_stats, const double *sums2, double count, double beta, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused grad-norm + clip + Adam over one flat parameter arena.
 * Replaces nn.utils.clip_grad_norm_ / get_grad_norm + torch.optim.Adam.step
 * (algorithms/actors/happo.py:93-100, algorithms/critics/v_critic.py:148-155).
 *   grad is first multiplied by *grad_scale (device scalar, e.g. 1/sum(active); NULL = 1);
 *   norm = ||grad||_2 ; if use_clip: grad *= min(1, max_norm/(norm+1e-6));
 *   m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
 *   lr, beta1, beta2 are DOUBLES as in torch.optim.Adam: 1-beta and lr/bc1 are formed in double and then rounded
 *   (1.f - 0.999f is 1.3e-5 away from float(0.001): enough to bias every step of the run).
 *   info_out[0] += norm (pre-clip) if info_out != NULL (double accumulator: the reference sums Python floats).
 */
int harl_gradnorm_clip_adam(float *param, float *grad, float *exp_avg, float *exp_avg_sq, long n,
                            const float *grad_scale, int use_clip, float max_norm, double lr, double beta1,
                            double beta2, float eps, float weight_decay, double bias_correction1,
                            double bias_correction2, double *info_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * MLP (harl/models/base/mlp.py:7-70) on the matrix pipes (H x H GEMMs: bf16 MFMA with an exact three-way fp32 operand
 * split, fp32 accumulation -- csrc/split_mfma.h; narrow GEMMs: fp32 MFMA).  LayerNorm affine terms are folded into the
 * following Linear (W' = W diag(gamma), b' = b + W beta); the kernels work on pure normalisations
 * and harl_mlp_unfold_grads maps the folded gradients back to the reference parameters.
 */
/* Wp[out,in] = W[out,in]*gamma[in]; bp[out] = b[out] + sum_k W[out,k]*beta[k]  (gamma/beta NULL = identity) */
int harl_fold_linear(const float *W, const float *b, const float *gamma, const float *beta, float *Wp,
                     float *bp, int out_dim, int in_dim, void *stream);
/* dW = dWp*gamma + dbp (x) beta ; db = dbp ; dgamma[k] = sum_o W[o,k] dWp[o,k] ; dbeta[k] = sum_o W[o,k] dbp[o]
 * dWp has row stride ldp (>= in_dim).  dgamma/dbeta may be NULL; accumulate != 0 adds into them (Linears sharing one
 * LayerNorm, e.g. the three GRU gate blocks). */
int harl_unfold_linear_grads(const float *dWp, const float *dbp, int ldp, const float *W, const float *gamma,
                             const float *beta, float *dW, float *db, float *dgamma, float *dbeta,
                             int out_dim, int in_dim, int accumulate, void *stream);

/* first layer: x_hat1 = norm(relu(Wp * norm0(X[idx]) + bp))
 *   X[rows, ldx] row-major, D features; use_ln0: feature LayerNorm on the input (mlp.py:57-58,65-66)
 *   outputs: xout ATL(H), relu_mask [n_slabs][H/64][64] u32, rstd[M_pad], mu0/rstd0[M_pad] (input LN stats)
 *   x0n (optional, honoured for D <= 64 only): ATL(32*ceil(D/32)) image of the normalised inputs, zero-padded -- the
 *   B operand of harl_mlp_dw_partials(b_kind = 0) for this layer's weight gradient (no second gather of raw rows) */
int harl_mlp_fwd_input(const float *X, long ldx, const int64_t *idx, long M, int D, const float *Wp,
                       const float *bp, int use_ln0, int H, float *xout, uint32_t *relu_mask, float *rstd,
                       float *mu0, float *rstd0, float *x0n, void *stream);
/* observations wider than 32 (32 < D <= 512; harl_amd/csrc/wide.hip; harl_mlp_x0n_wide itself accepts any D >= 1).  The
 * first layer is split in two streaming kernels:
 *   harl_mlp_x0n_wide: x0n = ATL(KP) image of norm0(X[idx]) (KP = D rounded up to 32, zero padded; use_ln0 = 0: the raw
 *     rows), mu0 / rstd0 by minibatch position.  x0n depends on the inputs only: it is also the operand of
 *     harl_mlp_tangent_wide and of harl_mlp_dw_partials(b_kind = 0, K = KP) for this layer.
 *   harl_mlp_fwd_wide: xout = norm(relu(Wp x0n + bp)) with Wp [H][D]; w_img = scratch of 3 * H * KP * 2 bytes (the three
 *     bf16 images of Wp, rebuilt on every call). */
int harl_mlp_x0n_wide(const float *X, long ldx, const int64_t *idx, long M, int D, int use_ln0, float *x0n, float *mu0,
                      float *rstd0, void *stream);
int harl_mlp_fwd_wide(const float *x0n, long M, int KP, const float *Wp, int D, const float *bp, int H, void *w_img,
                      float *xout, uint32_t *relu_mask, float *rstd, void *stream);
/* Hidden width 256 (csrc/panel.hip; the reference's dexhands HAPPO configurations, mlp.py:7-70 with hidden_sizes
 * [256, 256, 256]).  Three bf16 images of a 256-row matrix do not fit the LDS, so K is walked in 32-column panels that the
 * workgroup restages from L2; everything else (ATL images, ReLU masks, LayerNorm statistics) is as in the 64/128-wide kernels.
 *   harl_mlp_panel_fwd: xout = norm(relu(Wp xin + bp)), xin = ATL(KP) image (x0n of harl_mlp_x0n_wide, or x_hat of the layer
 *     before with KP = D = 256), Wp [256][D] row-major.
 *   harl_mlp_panel_bwd: dz_prev = relu' . LNbwd(Wp^T dz) for a 256 -> 256 layer (the role of harl_mlp_bwd_dx).
 * Weight gradients: harl_mlp_dw_partials(a_kind = 0, HO = 256). */
int harl_mlp_panel_fwd(const float *xin, long M, int KP, const float *Wp, int D, const float *bp, int HO, float *xout,
                       uint32_t *relu_mask, float *rstd, void *stream);
/* Head weight gradient of a 256-wide trunk from the row-major head gradients [M_pad][32] (HATRPO's surrogate gradient and
 * Fisher-vector product, hatrpo.py:95-140; HAPPO's loss kernel fuses it): n_wg partial rows dWp[32][256] | dbp[32] over
 * contiguous sample ranges, the layout harl_mlp_dw_partials leaves for narrower trunks (combined by the same reduce). */
int harl_head_dw_rows256(const float *dhead, long M, int act_dim, const float *xhat, float *part, int n_wg, void *stream);
/* Forward-mode tangent of a 256-wide layer (HATRPO's Fisher-vector product, trpo_util.py:132-158, on the dexhands-shaped
 * networks): x_out_dot = LNjac(mask . (Wdp x_in + bdp [+ Wp x_in_dot])) with the PRIMAL x_hat / mask / rstd of this layer.
 * First layer: x_in = the x0n image (KP wide, D valid columns), x_in_dot = NULL (the inputs carry no tangent); hidden layers:
 * KP = D = 256, x_in = x_hat of the layer before, x_in_dot its tangent, Wp the folded weights. */
int harl_mlp_panel_tangent(const float *xin_dot, const float *xin, long M, int KP, const float *Wp, const float *Wdp, int D,
                           const float *bdp, const float *xprimal, const uint32_t *mask_in, const float *rstd_in,
                           float *xout_dot, void *stream);
int harl_mlp_panel_bwd(const float *dz, const float *xprev, const uint32_t *relu_mask_prev, const float *rstd_prev, long M,
                       int HO, int HI, const float *Wp, float *dz_prev, void *stream);
/* the same two layers FROM the x0n ATL(32 | 64) image of harl_mlp_x0n_wide (D <= 64; identity row order; the image is
 * built once per buffer): no gather, no input-LayerNorm work per call */
int harl_mlp_fwd_fused2x(const float *x0n, long M, const float *W1p, int D, const float *b1p, const float *W2p,
                         const float *b2p, int H, int store1, float *x1out, uint32_t *mask1, float *rstd1, float *x2out,
                         uint32_t *mask2, float *rstd2, void *stream);
/* fused layers 1+2 for narrow inputs (D <= 32) and equal widths H: x_hat_1 stays in registers between the two GEMMs;
 * store1 != 0 also writes x_hat_1 / mask1 / rstd1 / mu0 / rstd0 and (if non-NULL) x0n as in harl_mlp_fwd_input
 * (needed only when a backward pass follows). */
int harl_mlp_fwd_fused2(const float *X, long ldx, const int64_t *idx, long M, int D, const float *W1p,
                        const float *b1p, int use_ln0, const float *W2p, const float *b2p, int H, int store1,
                        float *x1out, uint32_t *mask1, float *rstd1, float *mu0, float *rstd0, float *x2out,
                        uint32_t *mask2, float *rstd2, float *x0n, void *stream);
/* hidden layer: xout = norm(relu(Wp * xin + bp)), xin ATL(HI) -> xout ATL(HO) */
int harl_mlp_fwd_hidden(const float *xin, long M, int HI, int HO, const float *Wp, const float *bp,
                        float *xout, uint32_t *relu_mask, float *rstd, void *stream);
/* backward through Linear(HI->HO) then the preceding relu+norm:
 *   dz_prev = relu_mask_prev ? LNbwd(Wp^T dz ; xprev, rstd_prev) : 0     (dz ATL(HO) -> dz_prev ATL(HI))
 * dw_part != NULL: FIRST-layer variant -- dz_prev is dz_1, whose only consumer is dW_1' = dz_1^T x0n; that GEMM is fused
 * in (x0n = the ATL(kp0) image written by the forward pass, kp0 in {32, 64}, in_dim < kp0 so that its last pad column is
 * the column of ones that yields db_1').  Launched with n_wg workgroups, each writing one partial [HI*kp0 + HI] in the
 * layout of harl_mlp_dw_partials; dz_prev may then be NULL (dz_1 is not needed in HBM). */
int harl_mlp_bwd_dx(const float *dz, const float *xprev, const uint32_t *relu_mask_prev, const float *rstd_prev,
                    long M, int HO, int HI, const float *Wp, float *dz_prev, const float *x0n, int kp0, float *dw_part,
                    int n_wg, void *stream);
/* The WHOLE backward of one hidden Linear(128 -> 128) and the relu + norm in front of it in ONE persistent launch (round 5;
 * autograd through MLPLayer, harl/models/base/mlp.py:25-38, happo.py:93-100):
 *   dz_prev = relu_mask_prev ? LNbwd(Wp^T dz ; xprev, rstd_prev) : 0,   dW'[o][k] = sum_s dz[s][o] xprev[s][k],   db'[o] = sum_s dz[s][o]
 * i.e. harl_mlp_dw_partials(dz, xprev) + harl_mlp_bwd_dx(...) with dz and xprev read from HBM ONCE.  dw2_part: n_wg partial rows
 * [HO*HI + HO] in the layout of harl_mlp_dw_partials (the launch uses min(n_wg, 256) of them and clears the rest).
 * dw1_part != NULL: FIRST-layer variant as in harl_mlp_bwd_dx (x0n = ATL(32) image, kp0 = 32; dz_prev may be NULL);
 * dw1_part == NULL: dz_prev (ATL(HI)) is written.  fill: 1 = the operand splits are interleaved with the MFMAs of the
 * weight-gradient rounds (default), 0 = the same work in separate phases (A/B measurements).  HO = HI = 128 only. */
int harl_mlp_bwd_dx_dw(const float *dz, const float *xprev, const uint32_t *relu_mask_prev, const float *rstd_prev,
                       long M, int HO, int HI, const float *Wp, float *dz_prev, const float *x0n, int kp0, float *dw1_part,
                       float *dw2_part, int n_wg, int fill, void *stream);
/* weight-gradient partials: part[wg] = { dWp[HO_pad32, KP] , dbp[HO_pad32] } summed over the samples the
 * workgroup processed; a_kind: 0 = ATL(HO) dz, 1 = row-major [M, lda] (head gradients, HO <= 32);
 * b_kind: 0 = ATL(K) x_hat (K = 32, 64, 128, or a wide x0n image: any multiple of 32 up to 512), 1 = raw X[idx] rows
 * (ldx, D=K) normalised with mu0/rstd0 (NULL = no LN0).
 * n_wg workgroups (= number of partial slabs) ; KP = K rounded up to 32. */
int harl_mlp_dw_partials(const float *a, int a_kind, int lda, int HO, const float *b, int b_kind, long ldx,
                         const int64_t *idx, const float *mu0, const float *rstd0, int K, long M, float *part,
                         int n_wg, void *stream);
/* n <= 8 independent weight-gradient problems of ONE shape (square 64 / 128 blocks, both operands ATL images of M rows) in a
 * single launch: a[k], b[k], part[k] are HOST arrays of device pointers; problem k writes n_wg partial rows of
 * dWp[HO][K] | dbp[HO] at part[k] exactly as harl_mlp_dw_partials would.  Used for the six gate blocks of a GRU
 * (autograd of nn.GRU's weight_ih / weight_hh, harl/models/base/rnn.py:14-27). */
int harl_mlp_dw_partials_multi(int n, const float *const *a, const float *const *b, float *const *part, int HO, int K,
                               long M, int n_wg, void *stream);
/* ... and of DIFFERENT widths in one launch (round 6): problem k multiplies the ATL(64) image a[k] with column tiles
 * tile0[k] .. tile0[k] + nt[k] - 1 (nt <= 4) of the ATL(K[k]) image b[k] and writes n_wg partial rows dWp[64][K[k]] | dbp[64] at
 * part[k] (db' by the tile0 = 0 group) exactly as harl_mlp_dw_partials does for that group.  n <= 12; a / b / part and K / tile0 /
 * nt are HOST arrays.  All weight gradients of a 64-wide recurrent network (three MLP layers, six gate blocks) in one launch:
 * autograd through harl/models/base/mlp.py:25-38 and nn.GRU (harl/models/base/rnn.py:14-27). */
int harl_mlp_dw_partials_multi_v(int n, const float *const *a, const float *const *b, float *const *part, int HO, const int *K,
                                 const int *tile0, const int *nt, long M, int n_wg, void *stream);
/* The six gate blocks of a 64-wide GRU as ONE weight-gradient problem (autograd of nn.GRU's weight_ih_l0 / weight_hh_l0 and both
 * biases, harl/models/base/rnn.py:14-27): part[0..2] = d gi_g^T xhat, part[3..5] = d gh_g^T hpm with d gi = [dr, dz, dn],
 * d gh = [dr, dz, dhn]; all six operands ATL(64) images of M rows, each read once; part: HOST array of six device pointers, each
 * receiving n_wg partial rows dWp[64][64] | dbp[64] exactly as harl_mlp_dw_partials_multi(HO = K = 64) writes them. */
int harl_gru_dw6(const float *dr, const float *dz, const float *dn, const float *dhn, const float *xhat, const float *hpm,
                 float *const *part, long M, int n_wg, void *stream);
/* The whole 64-wide trunk in one launch per direction (csrc/trunk.hip; replaces MLPBase.forward / autograd through it,
 * harl/models/base/mlp.py:41-70, and the input half of nn.GRU's gates, harl/models/base/rnn.py:23-81).
 * harl_mlp_fwd_trunk: layer 1 as harl_mlp_fwd_wide (x0n ATL(KP), W1p [64][D], w_img scratch), then n_hidden (1 or 2) layers
 *   as harl_mlp_fwd_hidden (Wp[l] [64][64], bp[l]), then -- gi_ws != NULL -- the gate product of harl_gru_fwd's first phase
 *   (Wih [192][64] folded, b_ih + b_hh for r and z) into gi_ws (3 * M_pad * 64 floats; pass save | 2 to harl_gru_fwd).
 *   xout / relu_mask / rstd: HOST arrays of n_hidden + 1 device pointers; a NULL xout[l] skips that layer's activation record
 *   (forward-only passes).  Results are bit-identical to the layer-by-layer launches.
 * harl_mlp_bwd_trunk: Wih != NULL: d x_hat_top = Wih^T [dr, dzg, dn] and the LayerNorm / ReLU backward of the top MLP layer
 *   (the dx launch of harl_gru_bwd; call that with dz_mlp = NULL) -> dz_out[0]; else dz of the top layer is read from dz_in.
 *   Then for k = 0 .. n_hidden - 1: dz_out[k + 1] = harl_mlp_bwd_dx(dz_out[k] (or dz_in), xh[k + 1], relu_mask[k + 1],
 *   rstd[k + 1], Wp[k]) with Wp[k] the folded weights of the k-th hidden Linear FROM THE TOP; xh[0] .. rstd[0] describe the top
 *   layer (used by the gate stage only).  HOST arrays of device pointers; bit-identical to the separate launches. */
int harl_mlp_fwd_trunk(const float *x0n, long M, int KP, const float *W1p, int D, const float *b1p, int H, void *w_img,
                       int n_hidden, const float *const *Wp, const float *const *bp, float *const *xout,
                       uint32_t *const *relu_mask, float *const *rstd, const float *Wih, const float *bih, const float *bhh,
                       float *gi_ws, void *stream);
int harl_mlp_bwd_trunk(long M, int H, int n_hidden, const float *Wih, const float *dr, const float *dzg, const float *dn,
                       const float *dz_in, const float *const *Wp, const float *const *xh, const uint32_t *const *relu_mask,
                       const float *const *rstd, float *const *dz_out, void *stream);
/* out[e] = sum_w part[w][e] in fixed order (deterministic), e < elems */
int harl_reduce_partials(const float *part, int n_wg, long elems, float *out, void *stream);

/* Layer table used by the two fused kernels below: int32[HARL_TABLE_STRIDE * n_layers] in device memory, per Linear
 * (hidden layers in order, head last):
 *   0 w_off 1 b_off 2 gamma_off(-1) 3 beta_off(-1)  -- offsets into the flat parameter/gradient arena (floats)
 *   4 out   5 in    6 pack_w_off    7 pack_b_off     -- offsets into the folded-weight arena
 *   8 dwp_off (dense folded gradient dWp[op][kp] then dbp[op])  9 kp  10 op  11 part_off (per-workgroup partials arena,
 *   row stride op*kp+op floats per workgroup) */
#define HARL_TABLE_STRIDE 12
/* dwp[dwp_off_l + e] = sum_w part[part_off_l + w*elems_l + e] for all layers in ONE launch (fixed order, deterministic) */
int harl_reduce_partials_multi(const float *part, const int *table, int n_layers, int n_wg, long total_elems,
                               float *dwp, void *stream);
/* harl_fold_linear for EVERY entry of the device layer table (HARL_TABLE_STRIDE ints per entry, as harl_adam_fold reads it) in
 * one launch: packs[pack_w] = W * gamma, packs[pack_b] = b + W . beta, one block per output row, total_rows = sum of the
 * entries' out_dim.  The same arithmetic per row as harl_fold_linear (the fold in front of every update: models/base/mlp.py:25-38
 * evaluated with the LayerNorm affine folded into the next Linear). */
int harl_fold_table(const float *param, float *packs, const int *table, int n_layers, int total_rows, void *stream);
/* harl_unfold_linear_grads / harl_fold_linear_tangent for EVERY entry of the layer table in one launch each (same arithmetic
 * and summation order; Linears sharing a LayerNorm accumulate in table order): grad (reference parameter layout) from the
 * dense folded gradients `dwp`; pack_d (laid out like the folded-weight arena) = tangent of the folded weights in direction
 * `vec`.  total_cols = sum of the entries' input widths, total_rows = sum of their output widths.  Replaces autograd through
 * the LayerNorm affine terms (harl/models/base/mlp.py:25-38) in HATRPO's Fisher-vector product (trpo_util.py:132-158). */
int harl_unfold_table(const float *param, float *grad, const float *dwp, const int *table, int n_layers, int total_cols,
                      void *stream);
int harl_fold_tangent_table(const float *param, const float *vec, float *pack_d, const int *table, int n_layers,
                            int total_rows, void *stream);
/* hilo[k*PS + t], k = 0..3: the fp64 scalar t split into four fp32 pieces on a fixed exponent grid (quanta 2^24, 2^4,
 * 2^-16, 2^-36, each piece an integer multiple of its quantum below 2^20): the loss scalars ride behind the folded
 * gradients in the single fp32 SUM all-reduce of the data-parallel path and every piece sums EXACTLY over <= 16 ranks. */
int harl_pack_scalars_hilo(const double *scalars, float *hilo, void *stream);
/* harl_reduce_scalars (overwriting `scalars`, not accumulating) + harl_pack_scalars_hilo in ONE launch: what a data-parallel
 * optimiser step does in front of its collective (harl_amd/happo.py _optimizer_step, v_critic.py; reference: the loss / entropy
 * / ratio means of happo.py:77-91 and v_critic.py:112 become global means).  n_blocks = 0: zeros (no local row). */
int harl_reduce_pack_scalars(const float *part_scalars, int n_blocks, double *scalars, float *hilo, void *stream);
/* Fused optimiser epilogue (64 co-resident workgroups, ONE software grid barrier): loss scalars -> gradient scale +
 * statistics (info), unfold the folded gradients of every table entry into `grad` (reference parameter layout; Linears
 * sharing one LayerNorm accumulate its gradients), ||grad||, clip, Adam, re-fold the updated weights into `packs` -- Adam and
 * the re-fold run row by row of the table entries without a barrier in between, the LayerNorm parameters are written back by
 * the last workgroup to finish.
 * mode 0 (actor): scale = 1/scalars[1] (sum active), info += {loss, entropy, grad_norm, ratio};
 * mode 1 (critic): scale = const_scale (= value_loss_coef / m), info += {value_loss, grad_norm}.  `info` is double: every
 * update's fp32 figure is added the way the reference adds `.item()` values to a Python float (happo.py:145-150).
 * part_scalars != NULL: `scalars` (double[HARL_PS_STRIDE]) is first computed here as the fixed-order sum of the loss
 * kernel's n_scalar_blocks partial rows (single-GPU path); else scalars_hilo != NULL: `scalars` = sum of the four
 * all-reduced fp32 pieces written by harl_pack_scalars_hilo (data-parallel path); else `scalars` already holds the sums.
 * logstd_off >= 0: grad[logstd_off + d] = scalars[8 + d].  ws: >= 32 KiB device workspace, zero-initialised ONCE by the
 * caller (barrier words are reset by the kernel).  Every parameter must belong to a table entry (weight, bias, or the
 * LayerNorm in front of one) or be the log_std block.  Replaces clip_grad_norm_ + Adam.step + the LayerNorm-affine adjoint
 * (algorithms/actors/happo.py:89-100, algorithms/critics/v_critic.py:144-155, utils/models_tools.py:110-117). */
int harl_adam_fold(float *param, float *grad, float *exp_avg, float *exp_avg_sq, long n, const float *dwp,
                   const int *table, int n_layers, float *packs, double *scalars, const float *part_scalars,
                   int n_scalar_blocks, const float *scalars_hilo, int mode, float const_scale, int logstd_off,
                   int act_dim, double *info,
                   int use_clip, float max_norm, double lr, double beta1, double beta2, float eps, float weight_decay,
                   double bias_correction1, double bias_correction2, void *ws, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Heads and losses.
 * Gaussian head: DiagGaussian (models/base/distributions.py:58-89) + FixedNormal.log_probs/entropy,
 * Categorical head: distributions.py:7-21,37-55; ACTLayer.evaluate_actions (models/base/act.py:104-157).
 *   xL ATL(H); Whp[act_dim, H], bhp[act_dim] folded head weights; log_std[act_dim] (Gaussian only)
 *   discrete: actions[rows,1] hold the action index as fp32, avail[rows, act_dim] or NULL.
 */
/* log-probs only (the old/new passes of on_policy_ha_runner.py:66-83,96-113):
 *   logp_out[M, act_w] (act_w = act_dim for Box, 1 for Discrete), natural row order (no gather).
 *   If factor != NULL: factor[i] *= agg_d exp(logp - old_logp[i,d]) (fused on_policy_ha_runner.py:116-124)
 *   and logp_out may be NULL.  head_out (nullable) [M, act_dim] receives the head outputs themselves: the Gaussian
 *   mean, or the normalised Categorical logits (used for rollout sampling and for HATRPO's KL). */
int harl_actor_head_logp(const float *xL, long M, int H, const float *Whp, const float *bhp,
                         const float *log_std, float std_x_coef, float std_y_coef, int discrete, int act_dim,
                         const float *actions, const float *avail, float *logp_out, const float *old_logp,
                         float *factor, int agg_mean, float *head_out, long m_valid, long m_pad, void *stream);
/* HAPPO.update loss forward + backward (algorithms/actors/happo.py:28-102), everything up to dz_L:
 *   inputs gathered by idx: actions, old_logp[rows,act_w], adv[rows] (raw), adv_moments (double[3], NULL = adv
 *   already normalised), factor[rows], active[rows] (NULL = ones / use_policy_active_masks False)
 *   outputs: dzL ATL(H) (grad wrt last hidden pre-activation, UNSCALED by 1/sum(active)),
 *            dhead[M_pad, 32] (grad wrt head outputs; dw input), part_scalars[harl_head_blocks(M)][HARL_PS_STRIDE]:
 *            {0: sum loss*active, 1: sum active, 2: sum ent*active, 3: sum ratio, 4: count, 8..8+act_dim: dlogstd}
 *   mask/rstd: relu mask and rstd of the last hidden layer.
 *   trpo: 0 = HAPPO; 1 = HATRPO surrogate  sum_s +ratio*f*adv*active  (no clip, no entropy term;
 *   algorithms/actors/hatrpo.py:77-95); 2 = HAA2C  sum_s -ratio*f*adv*active  (no clip; algorithms/actors/haa2c.py:70-80).
 *   Scalar 0 is that sum.
 *   dw_part (optional) + n_wg: fuse the head weight gradient dW_head' = dhead^T x_hat_L into this kernel: launched with
 *   n_wg workgroups, each writing one partial [32*H + 32] in the layout of harl_mlp_dw_partials (a_kind = 1); dhead is
 *   then not written and the separate harl_mlp_dw_partials launch for the head is not needed; part_scalars has n_wg rows.
 *   logp_out (optional): log pi(a|o) of every row under the current parameters, [M, act_w] by batch position -- with a
 *   single full-buffer minibatch the first epoch's forward IS the runner's pre-update log-prob pass
 *   (on_policy_ha_runner.py:66-83), so the runner takes it from here instead of running that pass separately.
 */
int harl_actor_head_loss(const float *xL, const uint32_t *relu_mask, const float *rstd, long M, int H,
                         const float *Whp, const float *bhp, const float *log_std, float std_x_coef,
                         float std_y_coef, int discrete, int act_dim, const int64_t *idx, const float *actions,
                         const float *avail, const float *old_logp, const float *adv, const double *adv_moments,
                         const float *factor, const float *active, double clip_param, float entropy_coef,
                         int agg_mean, int trpo, long m_valid, long m_pad, float *logp_out, float *dzL, float *dhead,
                         float *part_scalars, float *dw_part, int n_wg, void *stream);
/* V head forward: values[M] = Whp . xL + bhp   (v_net.py:64) */
int harl_critic_head_values(const float *xL, long M, int H, const float *Whp, const float *bhp, float *values,
                            void *stream);
/* Row validity for recurrent batches (all head kernels): with m_pad > 0, row j counts only if (j % m_pad) < m_valid
 * (padding sequences inside every time step); m_pad = 0 means plain j < M. */
/* VCritic.cal_value_loss forward + backward (algorithms/critics/v_critic.py:75-114), up to dz_L (UNSCALED by
 * value_loss_coef / m).  vn_stats NULL = no ValueNorm (it must already contain this step's update).
 * part_scalars[harl_head_blocks(M)][HARL_PS_STRIDE]: {0: sum loss, 1: count} */
int harl_critic_head_loss(const float *xL, const uint32_t *relu_mask, const float *rstd, long M, int H,
                          const float *Whp, const float *bhp, const int64_t *idx, const float *value_preds,
                          const float *returns, const float *vn_stats, float clip_param, int use_clipped,
                          int use_huber, float huber_delta, long m_valid, long m_pad, float *dzL, float *dhead,
                          float *part_scalars, float *dw_part, int n_wg, void *stream);  /* dw_part/n_wg: see actor */
/* ---------------------------------------------------------------------------------------------
 * HATRPO (algorithms/actors/hatrpo.py:37-194, utils/trpo_util.py:47-158).  The Fisher-vector product
 * F v = grad((grad KL) . v) is evaluated as J^T M (J v) (exact at theta_new == theta_old, where KL's first-order terms
 * vanish): a forward-mode tangent pass (harl_fold_linear_tangent, harl_mlp_tangent_input/_hidden), M and the head
 * backward (harl_actor_head_fvp), then the ordinary backward kernels above.
 */
/* Wpd = Wd*gamma + W*gammad ; bpd = bd + Wd.beta + W.betad   (gamma/beta NULL: Wpd = Wd, bpd = bd) */
int harl_fold_linear_tangent(const float *W, const float *gamma, const float *beta, const float *Wd, const float *bd,
                             const float *gammad, const float *betad, float *Wpd, float *bpd, int out_dim, int in_dim,
                             void *stream);
/* x1dot = LNjac(mask1 * (Wdp norm0(X[idx]) + bdp)) given the primal x1 / mask1 / rstd1 (ATL(H)) */
int harl_mlp_tangent_input(const float *X, long ldx, const int64_t *idx, long M, int D, const float *Wdp,
                           const float *bdp, int use_ln0, int H, const float *x1, const uint32_t *mask1,
                           const float *rstd1, float *x1dot, void *stream);
/* z = W' x0n + b' as an ATL(H) image, no epilogue (the first Linear of an MLP whose activation is not ReLU, mlp.py:25-30);
 * arguments as harl_mlp_fwd_wide. */
int harl_mlp_linear_wide(const float *x0n, long M, int KP, const float *Wp, int D, const float *bp, int H, void *w_img,
                         float *zout, void *stream);
/* Activation functions other than ReLU (models_tools.py:28-50; act: 1 leaky_relu, 2 tanh, 3 sigmoid, 4 selu), element-wise
 * over ATL(H) images, H = 64 / 128.  harl_act_ln_fwd: x_hat = LayerNorm(act(z)) with mean(act(z)) and rstd per sample
 * (mlp.py:25-38).  harl_act_bwd: dz <- dz * act'(z) in place, act' from the activation value x_hat / rstd + mean; dz holds
 * the LayerNorm backward as harl_mlp_bwd_dx / the loss kernels leave it when they are given an all-ones ReLU mask. */
int harl_act_ln_fwd(const float *z, long M, int H, int act, float *xhat, float *mean, float *rstd, void *stream);
int harl_act_bwd(float *dz, const float *xhat, const float *mean, const float *rstd, long M, int H, int act, void *stream);
/* Forward-mode tangent of [activation, LayerNorm] for HATRPO's Fisher-vector product (harl/utils/trpo_util.py:132-158 on
 * networks built with activation_func != relu, mlp.py:25-38):  a_dot = act'(z) (zd1 + zd2),
 * x_hat_dot = rstd (a_dot - mean_f(a_dot) - x_hat mean_f(x_hat a_dot)).  zd1, zd2 (zd2 may be NULL): ATL images of the two
 * halves of the pre-activation's tangent (raw GEMMs: harl_mlp_linear / harl_mlp_linear_wide); xhat / mean / rstd as
 * harl_act_ln_fwd left them; act' is taken from the activation value like harl_act_bwd.  act = 0: no activation (mean may be
 * NULL) -- the LayerNorm tangent alone, used for rnn.norm of the composed GRU. */
int harl_act_ln_tangent(const float *zd1, const float *zd2, const float *xhat, const float *mean, const float *rstd, long M,
                        int H, int act, float *xhat_dot, void *stream);
/* the same for wide inputs, from the x0n image of harl_mlp_x0n_wide (w_img: scratch as in harl_mlp_fwd_wide) */
int harl_mlp_tangent_wide(const float *x0n, long M, int KP, const float *Wdp, int D, const float *bdp, int H, void *w_img,
                          const float *x1, const uint32_t *mask1, const float *rstd1, float *x1dot, void *stream);
/* GRU policies: forward-mode tangent through the recurrence (csrc/gru.hip).
 * harl_gru_gates: out_g (+)= W_g xin for the three gate blocks of W [3H][H] (xin ATL(H), n_slabs slabs; bit g of acc_mask:
 *   accumulate into out_g instead of overwriting) -- the recurrence-free parts of the gate tangents, all steps at once.
 * harl_gru_tangent: y' for L steps x m_pad sequences from  g_r = W_ir x' + W_ir' x + W_hr' h~ (same for g_z),
 *   g_nx = W_in x' + W_in' x,  g_nh = W_hn' h~,  the primal tensors saved by harl_gru_fwd(save = 1) and its outputs
 *   y / rstd_y, the primal W_hh and the bias tangents bihd / bhhd [3H]; the initial hidden state has no tangent. */
int harl_gru_gates(const float *xin, const float *W, int H, long n_slabs, float *out_r, float *out_z, float *out_n,
                   int acc_mask, void *stream);
int harl_gru_tangent(const float *g_r, const float *g_z, const float *g_nx, const float *g_nh, const float *mask_rows,
                     const float *Whh, const float *bihd, const float *bhhd, const float *hpm, const float *r,
                     const float *z, const float *n, const float *hn, const float *y, const float *rstd_y, int H, int L,
                     long m_pad, float *ydot, void *stream);
/* xout_dot = LNjac(mask * (Wp xin_dot + Wdp xin + bdp)) given the primal xprimal / mask / rstd of this layer */
int harl_mlp_tangent_hidden(const float *xin_dot, const float *xin, long M, int HI, int HO, const float *Wp,
                            const float *Wdp, const float *bdp, const float *xprimal, const uint32_t *mask_in,
                            const float *rstd_in, float *xout_dot, void *stream);
/* The same tangent in ONE launch (csrc/wide.hip): a K = 2 HI GEMM  [W' | W'_dot] [x_in_dot ; x_hat_in]  over the two input
 * images with the weight images (built per call into w_img: 3 * HO * 2 HI bf16 = 6 HO HI bytes of device scratch) streamed
 * from L2, then the LayerNorm Jacobian -- no pass through the output image (trpo_util.py:132-158, the J v half of F v). */
int harl_mlp_tangent_hidden2(const float *xin_dot, const float *xin, long M, int HI, int HO, const float *Wp,
                             const float *Wdp, const float *bdp, void *w_img, const float *xprimal,
                             const uint32_t *mask_in, const float *rstd_in, float *xout_dot, void *stream);
/* head tangent, M (Gaussian 1/sigma^2 on the mean; identity on the normalised logits incl. masked entries), head +
 * LayerNorm/ReLU backward -> dzL ATL(H), dhead[M_pad,32]; NOT yet divided by the batch size.  m_valid / m_pad as in
 * harl_actor_head_loss (padding sequences of a recurrent batch get zero gradients; 0, 0 = no padding) */
int harl_actor_head_fvp(const float *xL, const float *xLdot, const uint32_t *relu_mask, const float *rstd, long M, int H,
                        const float *Whp, const float *bhp, const float *Whdp, const float *bhdp, const float *log_std,
                        float std_x_coef, float std_y_coef, int discrete, int act_dim, const float *avail, long m_valid,
                        long m_pad, float *dzL, float *dhead, void *stream);
/* HATRPO's P-sized vector algebra, one launch each instead of a dozen torch ops (trpo_util.py:96-158):
 *   harl_trpo_fvp_finish: out = grad / m_global + damping * vec, with the log_std block (offset logstd_off, act_dim entries;
 *     log_std NULL for Categorical policies) replaced by 2 (dsigma/dlog_std)^2 / sigma^2 * vec  -- the epilogue of F v + 0.1 v.
 *   harl_trpo_cg_step: one conjugate-gradient iteration on (x, r, p) given avp = F p; state[0] = r.r, state[1] = done flag
 *     (the reference's `if rdotr < 1e-10: break`, kept on the device: a finished solve freezes x, r and p);
 *     scratch: HARL_CG_SCRATCH_BYTES of caller-owned, zero-filled device memory (see the defines at the top). */
int harl_trpo_fvp_finish(const float *grad, const float *vec, const float *log_std, float *out, long n, float m_global,
                         float damping, long logstd_off, int act_dim, float std_x_coef, float std_y_coef, void *stream);
int harl_trpo_cg_step(float *x, float *r, float *p, const float *avp, long n, float *state, void *scratch, void *stream);
/* HATRPO's scalar glue on the device (hatrpo.py:92-192; round 6): the set-up of the conjugate-gradient solve, the step size, the
 * expected improvement and the line search's accept test -- torch.dot / sqrt / .item() round trips in the reference and, until
 * round 5, here.  st = double[HARL_TRPO_STATE], the update's record, read by the host ONCE per line-search step:
 *   [0] surrogate at theta_old  [1] shs = 1/2 x.Fx  [2] step size  [3] expected improvement (x backtrack_coeff per backtrack)
 *   [4] fraction  [5] accepted (0/1)  [6] backtracks  [7] kl  [8] loss_improve  [9] surrogate at the candidate
 *   [10] dist_entropy  [11] ratio  [12] expected improvement at fraction 1
 * harl_trpo_begin : g = grad_sum * (float)(1 / scalars[1]) (hatrpo.py:92-95 on the unscaled sums of harl_actor_head_loss; the
 *                   log_std block logstd_off .. + act_dim of a Gaussian policy from scalars[8 ..], logstd_off < 0: none),
 *                   x = 0, r = p = g, cg_state = {r.r, 0} (trpo_util.py:101-105), st cleared, st[0] = scalars[0] / scalars[1].
 * harl_trpo_step  : st[1], st[2] (hatrpo.py:123-124), full_step = step x, theta_save = theta, st[3] = st[12] = g.full_step,
 *                   st[4] = 1, st[5] = st[6] = 0.
 * harl_trpo_ls_candidate : theta = theta_save + (float)st[4] * full_step (hatrpo.py:139-140).
 * harl_trpo_ls_test : new surrogate = scalars[0] / scalars[1], kl = kl_sum[0] / m_global (kl_sum is reset to 0), the accept test
 *                   `kl < delta and improve / expected > accept_ratio and improve > 0` (hatrpo.py:171-178); on reject st[3], st[4]
 *                   are multiplied by backtrack_coeff and st[6] incremented (hatrpo.py:184-185).
 * scratch (begin, step): HARL_CG_SCRATCH_BYTES, the block harl_trpo_cg_step uses.  Dot products: fp64 over fixed-order partials. */
#define HARL_TRPO_STATE 16
int harl_trpo_begin(const float *grad_sum, const double *scalars, long logstd_off, int act_dim, float *g, float *x, float *r,
                    float *p, long n, float *cg_state, double *st, void *scratch, void *stream);
int harl_trpo_step(const float *x, const float *fx, const float *g, const float *theta, float *theta_save, float *full_step,
                   long n, float kl_threshold, double *st, void *scratch, void *stream);
int harl_trpo_ls_candidate(const float *theta_save, const float *full_step, const double *st, float *theta, long n, void *stream);
int harl_trpo_ls_test(const double *scalars, double *kl_sum, double m_global, double kl_threshold, double accept_ratio,
                      double backtrack_coeff, double *st, void *stream);
/* scalars[0..HARL_PS_STRIDE) = column sums of the loss kernels' per-block partial rows: harl_reduce_scalars ACCUMULATES (zero
 * `scalars` first), harl_reduce_scalars_set overwrites.  harl_zero_bytes: hipMemsetAsync on `stream`. */
int harl_reduce_scalars_set(const float *part_scalars, int n_blocks, double *scalars, void *stream);
int harl_zero_bytes(void *p, long bytes, void *stream);
/* out_sum (double, accumulated) += sum_s KL(old || new)_s from the head outputs of harl_actor_head_logp:
 * Gaussian analytic KL in fp64 (trpo_util.py:54-62), Categorical kl_approx on normalised logits (trpo_util.py:47-51) */
int harl_trpo_kl_sum(const float *head_old, const float *head_new, const float *log_std_old, const float *log_std_new,
                     float std_x_coef, float std_y_coef, long M, int act_dim, int discrete, double *out_sum,
                     void *stream);

/* ---------------------------------------------------------------------------------------------
 * MultiDiscrete action spaces (harl/models/base/act.py:35-43,56-73,117-141: one Categorical per entry of nvec on the same
 * trunk output; csrc/multihead.hip).  The concatenated heads are ordinary Linears packed into n_groups GROUPS of at most 128
 * logits; group g is one weight matrix Wp_g [sp_g][H] (sp_g in {64, 128}, zero rows past its last head) with
 *   logits        z_g = harl_mlp_linear(x_hat_L, Wp_g, bp_g)                      ATL(sp_g) image
 *   weight grads  harl_mlp_dw_partials(dz_g, a_kind 0, HO = sp_g, x_hat_L)
 *   trunk grad    dz_L = sum_g harl_mlp_bwd_dx(dz_g, x_hat_L, mask_L, rstd_L, HO = sp_g, HI = H, Wp_g)
 * z / dz: HOST arrays of n_groups device pointers; nvec[n_heads] = logits per head, head_group[n_heads] = its image (heads
 * fill an image in order).  actions [rows, n_heads] hold the indices as fp32.
 * harl_mlp_linear: xout = Wp xin + bp, ATL(HI) -> ATL(HO), no epilogue (HI, HO in {64, 128}).
 * harl_md_head_logp: logp_out[M] = sum_heads log p_k(a_k) (act.py:124-137); if factor != NULL:
 *   factor[i] *= agg_c exp(logp - old_logp[i, c]), c < old_w (on_policy_ha_runner.py:116-124); head_out (nullable)
 *   [M, sum(nvec)] = the normalised logits of every head, concatenated (rollout sampling).
 * harl_md_head_loss: HAPPO / MAPPO (mode 0) or HAA2C (mode 2) loss and d(unscaled loss)/d(logits) -> dz_g (may alias
 *   z_g).  The reference compares the summed log-prob [m, 1] with EVERY column of the buffer's [rows, old_w = n_heads]
 *   log-prob array (happo.py:66-70), so `prod` raises the ratio to the old_w-th power; its entropy bonus is
 *   (1/m) sum_rows sum_heads H (act.py:126-139: never active-mask weighted), while the surrogate is divided by
 *   sum(active): ent_scale = device scalar sum(active) / m of the (global) minibatch (NULL = 1) folds that into the
 *   unscaled sums.  part_scalars[n_blocks][HARL_PS_STRIDE] as harl_actor_head_loss
 *   ({0: sum loss*active, 1: sum active, 2: ent_scale * sum ent, 3: sum ratio, 4: count}); launched with n_blocks workgroups. */
int harl_mlp_linear(const float *xin, long M, int HI, int HO, const float *Wp, const float *bp, float *xout, void *stream);
/* Three independent raw products z_g = W_g x_g + b_g in ONE launch (x_g ATL(HI) images of M rows, W_g [HO][HI], outputs ATL(HO)):
 * the three gate products of one time step of the composed 128-wide / stacked GRU (harl_amd/gru_wide.py; harl/models/base/rnn.py:8-81
 * through nn.GRU's gate blocks).  Widths 64 x 64 or 128 x 128.  Same arithmetic as three harl_mlp_linear calls: bit-identical. */
int harl_mlp_linear3(const float *x0, const float *x1, const float *x2, long M, int HI, int HO, const float *W0, const float *W1,
                     const float *W2, const float *b0, const float *b1, const float *b2, float *o0, float *o1, float *o2,
                     void *stream);
int harl_md_head_logp(const float *const *z, int n_groups, const int *sp, int n_heads, const int *nvec,
                      const int *head_group, long M, const float *actions, float *logp_out, const float *old_logp,
                      int old_w, float *factor, int agg_mean, float *head_out, long m_valid, long m_pad, void *stream);
int harl_md_head_loss(const float *const *z, float *const *dz, int n_groups, const int *sp, int n_heads, const int *nvec,
                      const int *head_group, long M, const int64_t *idx, const float *actions, const float *old_logp,
                      int old_w, const float *adv, const double *adv_moments, const float *factor, const float *active,
                      const float *ent_scale, double clip_param, float entropy_coef, int agg_mean, int mode, long m_valid,
                      long m_pad, float *logp_out, float *part_scalars, int n_blocks, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused optimiser-step kernels for two equal-width hidden layers (H in {64, 128}) and inputs up to 64 wide, identity
 * row order (csrc/update.hip).  They take the cached normalised-input image x0n (harl_mlp_x0n_wide) and keep every
 * activation on chip: x_hat_1, x_hat_2, the ReLU masks and LayerNorm statistics are never written; x_hat_1 is recomputed
 * from x0n in the backward launches.  Together they replace harl_mlp_fwd_fused2x + harl_actor_head_loss |
 * harl_critic_head_loss + harl_mlp_dw_partials(hidden) + harl_mlp_bwd_dx, i.e. autograd through MLPBase + ACTLayer | v_out
 * for one minibatch (algorithms/actors/happo.py:28-102, algorithms/critics/v_critic.py:116-157), and move ~1.9 KB per
 * sample instead of ~4.5 KB.  Loss arguments as in harl_actor_head_loss / harl_critic_head_loss (idx = NULL, no padding).
 *   harl_update_fwd_*  : forward, head, loss, head weight gradient, backward to dz2 (ATL(H), the only activation written);
 *                        part_scalars rows [n_part_rows][HARL_PS_STRIDE], dw_part_head rows [n_part_rows][32*H + 32]
 *                        (rows beyond the launch grid are cleared), logp_out as in harl_actor_head_loss.
 *   harl_update_bwd    : dW_2' | db_2' partial rows into dw_part2 ([n_part_rows][H*H + H]) and dW_1' | db_1' into dw_part1
 *                        ([n_part_rows][H*KP0 + H], KP0 = 32 or 64) from x0n and dz2.
 *   harl_update_logp   : forward + log-probs (+ factor *= agg(exp(new - old)), head_out) -- harl_mlp_fwd_fused2x +
 *                        harl_actor_head_logp without the x_hat_2 round trip (on_policy_ha_runner.py:66-124).
 *   harl_update_values : forward + values (v_critic.py:54-73).
 * xh1 / rmask1 / rstd1 (harl_update_fwd_*; all three or none): the HYBRID step -- the launch also leaves layer 1's activation
 *   record (x_hat_1 as an ATL image, ReLU mask words, 1/sigma) in HBM, in the format of harl_mlp_fwd_fused2x, so that the
 *   layer-by-layer backward (harl_mlp_bwd_dx + harl_mlp_dw_partials) runs behind it instead of harl_update_bwd: x_hat_2,
 *   its mask and statistic still never leave the chip, and nothing is recomputed.
 * harl_update_supported returns 1 when (D, H, act_dim) is inside the instantiated range (D <= 64, H in {64,128}, act_dim <= 8;
 *   D = 0: the last-layer entry points) AND the launch fits the 160 KiB of LDS a workgroup can have.  `kind`: 0 forward-only
 *   pass, 1 actor optimiser step, 2 critic optimiser step (an optimiser step also keeps wave-private head-gradient tiles in
 *   LDS: the ACTOR step of a 128-wide network with 33..64 inputs does not fit). */
int harl_update_supported(int D, int H, int act_dim, int kind);
int harl_update_fwd_actor(const float *x0n, long M, int D, int H, const float *W1p, const float *b1p, const float *W2p,
                          const float *b2p, const float *Whp, const float *bhp, const float *log_std, float std_x_coef,
                          float std_y_coef, int discrete, int act_dim, const float *actions, const float *avail,
                          const float *old_logp, const float *adv, const double *adv_moments, const float *factor,
                          const float *active, double clip_param, float entropy_coef, int agg_mean, int trpo,
                          float *logp_out, float *dz2, float *part_scalars, float *dw_part_head, int n_part_rows,
                          float *xh1, uint32_t *rmask1, float *rstd1, void *stream);
int harl_update_logp(const float *x0n, long M, int D, int H, const float *W1p, const float *b1p, const float *W2p,
                     const float *b2p, const float *Whp, const float *bhp, const float *log_std, float std_x_coef,
                     float std_y_coef, int discrete, int act_dim, const float *actions, const float *avail,
                     float *logp_out, const float *old_logp, float *factor, int agg_mean, float *head_out, void *stream);
int harl_update_fwd_critic(const float *x0n, long M, int D, int H, const float *W1p, const float *b1p, const float *W2p,
                           const float *b2p, const float *Whp, const float *bhp, const float *value_preds,
                           const float *returns, const float *vn_stats, float clip_param, int use_clipped, int use_huber,
                           float huber_delta, float *dz2, float *part_scalars, float *dw_part_head, int n_part_rows,
                           float *xh1, uint32_t *rmask1, float *rstd1, void *stream);
/* Last hidden layer + head + loss of a network with MORE than two hidden layers (or a wide first layer), as one launch
 * (k_upd_fwd with the previous layer's x_hat image as input): replaces harl_mlp_fwd_hidden(layer L) + harl_actor_head_loss /
 * harl_critic_head_loss in the optimiser steps -- x_hat_L, its ReLU mask and statistic never cross HBM; dz_L, the head's
 * weight-gradient partials and the loss partial sums come out exactly as the loss kernels write them, and the layer-by-layer
 * backward runs behind it unchanged (reference: the same autograd graph, happo.py:28-102 / v_critic.py:116-157).
 * xin = ATL(H) image of x_hat_{L-1} in batch order; idx (optional) gathers the per-row loss inputs as in the loss kernels. */
int harl_update_last_actor(const float *xin, long M, int H, const float *Wp, const float *bp, const float *Whp,
                           const float *bhp, const float *log_std, float std_x_coef, float std_y_coef, int discrete,
                           int act_dim, const int64_t *idx, const float *actions, const float *avail,
                           const float *old_logp, const float *adv, const double *adv_moments, const float *factor,
                           const float *active, double clip_param, float entropy_coef, int agg_mean, int trpo,
                           float *logp_out, float *dz, float *part_scalars, float *dw_part_head, int n_part_rows,
                           void *stream);
int harl_update_last_critic(const float *xin, long M, int H, const float *Wp, const float *bp, const float *Whp,
                            const float *bhp, const int64_t *idx, const float *value_preds, const float *returns,
                            const float *vn_stats, float clip_param, int use_clipped, int use_huber, float huber_delta,
                            float *dz, float *part_scalars, float *dw_part_head, int n_part_rows, void *stream);
int harl_update_values(const float *x0n, long M, int D, int H, const float *W1p, const float *b1p, const float *W2p,
                       const float *b2p, const float *Whp, const float *bhp, float *values, void *stream);
int harl_update_bwd(const float *x0n, const float *dz2, long M, int D, int H, const float *W1p, const float *b1p,
                    const float *W2p, float *dw_part1, float *dw_part2, int n_part_rows, void *stream);

/* ---------------------------------------------------------------------------------------------
 * GRU recurrent layer (models/base/rnn.py:8-81, recurrent_n = 1, H = 64), forward and BPTT.  A recurrent batch is L steps x m
 * sequences with row (l, j) at index l*m_pad + j (m_pad = m rounded up to 32); a wave owns 32 sequences for the whole chunk and
 * keeps the hidden state in registers.  xin / y / saved tensors are ATL(H) over L*m_pad rows; mask_rows[L*m_pad] are the
 * reset masks in that row order; h0 / h_last are row-major [m_pad, H].  Wih is the FOLDED input matrix (LayerNorm affine of the
 * last MLP layer folded in), gate order r, z, n.  y = normalised h (rnn.norm's affine part is folded into the head).
 * save != 0 stores h~ = h*mask, r, z, n, hn for the backward pass.
 * gi_ws (optional, 3*L*m_pad*H floats): when given, the input half of the gates (W_i' x + b) of ALL steps is computed
 * first by a fully parallel kernel and the recurrence only carries the W_hh products (bit-identical results).
 * save bit 1 (value 2): gi_ws ALREADY holds that product (harl_mlp_fwd_trunk wrote it) -- the gate launch is skipped and xin is
 * not read.  harl_gru_bwd with dz_mlp = NULL leaves the input side (W_ih'^T d gates and the MLP's LayerNorm backward) to
 * harl_mlp_bwd_trunk.
 */
int harl_gru_fwd(const float *xin, const float *mask_rows, const float *h0, const float *Wih, const float *bih,
                 const float *Whh, const float *bhh, int H, int L, long m_pad, float *y, float *rstd_y, float *hpm, float *r,
                 float *z, float *n, float *hn, float *h_last, int save, float *gi_ws, void *stream);
/* GRU steps composed from layer GEMMs, for hidden widths whose W_hh images do not fit the LDS of the fused kernels above
 * (H = 128; csrc/gru_cell.hip, host composition in harl_amd/gru_wide.py; H = 64 is instantiated too, as a cross-check of the
 * composition against harl_gru_fwd / harl_gru_bwd).  Per step l (m_pad rows, ATL(H) images):
 *   gh_g = harl_mlp_linear(h~_l, W_hg, b_hg), g = r, z, n ; gi_g (all steps) = harl_mlp_linear(x_hat, W_ig', b_ig')
 *   harl_gru_cell_init: hpm0 = h0 (row-major [m_pad, H]) * mask_0
 *   harl_gru_cell_fwd : r, z, n, hn (= gh_n; all four NULL when nothing is saved), h_l, h~_{l+1} = h_l * mask_next (NULL at the
 *     last step), h_last (row-major, nullable)
 *   harl_gru_cell_bwd : G_l = dh_out_l + mask_next * (gz + t_r + t_z + t_n) with gz = G_{l+1} z_{l+1} (in place: leaves G_l z_l)
 *     and t_g = W_hg^T dgh_{g,l+1} (all NULL at the last step); outputs dr, dz, dn (= d gi) and dhn (d gh = [dr, dz, dhn])
 *   harl_rownorm      : y = (x - mean) rstd over M rows (rnn.norm without its affine part), rstd[M_pad] */
int harl_gru_cell_init(const float *h0, const float *mask_rows, int H, long m_pad, float *hpm0, void *stream);
int harl_gru_cell_fwd(const float *gi_r, const float *gi_z, const float *gi_n, const float *gh_r, const float *gh_z,
                      const float *gh_n, const float *hpm, const float *mask_next, int H, long m_pad, float *r, float *z,
                      float *n, float *hn, float *h, float *hpm_next, float *h_last, void *stream);
int harl_gru_cell_bwd(const float *dh_out, const float *t_r, const float *t_z, const float *t_n, const float *mask_next,
                      const float *r, const float *z, const float *n, const float *hn, const float *hpm, int H, long m_pad,
                      float *gz, float *dr, float *dz, float *dn, float *dhn, void *stream);
int harl_rownorm(const float *x, long M, int H, float *y, float *rstd, void *stream);
/* Forward-mode tangent of one cell step (HATRPO's Fisher-vector product, harl/utils/trpo_util.py:132-158, through the composed
 * GRU of harl/models/base/rnn.py:8-81):  r_dot = r (1 - r) sum(g_r), z_dot = z (1 - z) sum(g_z),
 * n_dot = (1 - n^2) (sum(gi_n) + r_dot hn + r sum(gh_n)),  h_dot = (1 - z) n_dot + z_dot (h~ - n) + z h~_dot.
 * Gate tangents arrive as the ATL images of the raw GEMMs that form them: gia = W_i x_dot, gib = W_i_dot x + b_i_dot,
 * gha = W_h h~_dot (all three NULL together with hpm_dot at the first step), ghb = W_h_dot h~ + b_h_dot; r, z, n, hn, hpm as
 * harl_gru_cell_fwd saved them.  Emits h_dot of the step and (unless NULL) the next step's h~_dot = h_dot * mask_next. */
int harl_gru_cell_tangent(const float *gia_r, const float *gia_z, const float *gia_n, const float *gib_r, const float *gib_z,
                          const float *gib_n, const float *gha_r, const float *gha_z, const float *gha_n, const float *ghb_r,
                          const float *ghb_z, const float *ghb_n, const float *r, const float *z, const float *n,
                          const float *hn, const float *hpm, const float *hpm_dot, const float *mask_next, int H, long m_pad,
                          float *h_dot, float *hpm_dot_next, void *stream);
/* dhout = d(loss)/d(h_l) through the output path (after the rnn.norm backward, done by the head kernels with an all-ones relu
 * mask).  Outputs the gate gradients dr, dz, dn (d gi = [dr,dz,dn]) and dhn (d gh = [dr,dz,dhn]) as ATL(H) for
 * harl_mlp_dw_partials, and dz_mlp = LayerNorm/ReLU backward of W_ih'^T dgi for the last MLP layer (xmlp / mask_mlp / rstd_mlp). */
int harl_gru_bwd(const float *dhout, const float *mask_rows, const float *Wih, const float *Whh, const float *hpm,
                 const float *r, const float *z, const float *n, const float *hn, int H, int L, long m_pad,
                 const float *xmlp, const uint32_t *mask_mlp, const float *rstd_mlp, float *dr, float *dz, float *dn,
                 float *dhn, float *dz_mlp, void *stream);

/* number of workgroups (= rows of part_scalars) the head-loss kernels use for M samples */
int harl_head_blocks(long M);
/* scalars[j] += sum_b part_scalars[b][j], j < HARL_PS_STRIDE (fixed order, fp64) */
int harl_reduce_scalars(const float *part_scalars, int n_blocks, double *scalars, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Host helper (no GPU work): the permutation torch.randperm(n) WOULD return from the CPU generator whose state bytes
 * (torch.get_rng_state()) are given, and the state it would leave behind -- ATen's randperm_cpu algorithm replayed from
 * a copy of the mt19937 state without ATen's thread-pool fan-out (27 ms -> ~5 ms for n = 819200 on the GPU box's host).
 * out: int32[n], scratch: uint32[n] (caller-owned, reusable), state_out: state_bytes bytes for torch.set_rng_state().
 * Replaces the draw in on_policy_actor_buffer.py:131 / on_policy_critic_buffer_ep.py:223 bit for bit. */
int harl_randperm_replay(const uint8_t *state_in, long state_bytes, long n, int32_t *out, uint32_t *scratch,
                         uint8_t *state_out);
/* state_out = state_in advanced by n_draws 32-bit draws (host code; the generator advance of torch.randperm(n_draws + 1)
 * without materialising the permutation).  Returns 0 or -2 on an unexpected state layout. */
int harl_rng_advance(const uint8_t *state_in, long state_bytes, long n_draws, uint8_t *state_out);
/* The same advance with the bulk done by a GF(2) polynomial jump (x^J mod the characteristic polynomial of mt19937, applied to
 * 34 generated state blocks): ~0.1 ms whatever n_draws is, against 0.8 ms of block refreshes for the 6.5 M draws per sampler
 * call of an 8-GPU run (every rank replays the GLOBAL torch.randperm, on_policy_actor_buffer.py:131).  harl_rng_advance
 * switches to it by itself above HARL_RNG_JUMP_MIN_BLOCKS (default 2500) state blocks; this entry point always jumps
 * (tests).  Bit-identical final state.  The first call for a given block count builds its jump polynomial (~50 ms, cached). */
int harl_rng_jump(const uint8_t *state_in, long state_bytes, long n_draws, uint8_t *state_out);

/* ---------------------------------------------------------------------------------------------
 * One-shot SUM all-reduce of the data-parallel update's messages (csrc/comm.hip; opt-in, HARL_ALLREDUCE=oneshot).
 * Replaces nothing in the reference (it is single-process); it is the exchange step SURVEY.md section 8(e) prescribes for the
 * sharded update: < 100 KiB per optimiser step ([folded gradients | scalar pieces], harl_amd/dist.py) pushed by every rank
 * straight into a slot of every peer's hipIpc-mapped buffer, flags behind the data, the P local slots summed in rank order
 * (one hop over the point-to-point fabric instead of a ring's 2 (P - 1); the same bits on every rank).
 *   harl_comm_create  : HOST pointers.  Allocates this rank's buffer for messages up to cap_bytes (uncached / fine-grained
 *                       device memory where exportable), writes its 64-byte hipIpc handle to handle_out and the context to
 *                       *(void **)ctx_out.  Returns the allocation kind (2 uncached, 1 fine-grained, 0 plain) or < 0.
 *   harl_comm_connect : all_handles = world x 64 bytes (host), rank-major, gathered by the caller (torch.distributed object
 *                       gather); maps every peer's buffer.  Collective in the sense that every rank must get here.
 *   harl_comm_allreduce: in-place SUM of msg[0..n) (device; fp32, or fp64 when is_f64) on `stream`; same n and call order on
 *                       every rank.  No host state: safe under hipGraph capture.  A peer that does not show up within the
 *                       communicator's time-out (default 600 s; harl_comm_set_timeout, 0 = wait for ever like RCCL) turns
 *                       the result into NaN and sets the status word instead of hanging the device.  A time-out is FATAL
 *                       for the communicator (flags and epochs are out of step afterwards): the caller must check
 *                       harl_comm_status at its next host synchronisation point and stop.
 *   harl_comm_set_timeout: flag-wait limit of later launches in seconds, 0 = none (host state only).
 *   harl_comm_status  : 0, or q + 1 after a wait for rank q timed out (synchronises the device). */
int harl_comm_create(int world, int rank, long cap_bytes, int n_blocks, void *handle_out, void *ctx_out);
int harl_comm_connect(void *ctx, const void *all_handles);
int harl_comm_allreduce(void *ctx, void *msg, long n, int is_f64, void *stream);
int harl_comm_set_timeout(void *ctx, double seconds);
int harl_comm_status(void *ctx);
int harl_comm_destroy(void *ctx);

#ifdef __cplusplus
}
#endif
#endif /* HARL_HIP_H */
