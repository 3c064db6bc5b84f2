// heads.hip -- action / value heads with fused losses and their backward up to dz_L (gfx950).
//
//   actor : DiagGaussian / Categorical head  -> log-probs  (+ fused factor product)            [logp pass]
//           ... -> importance ratio, clipped surrogate x factor, entropy, d(loss)/d(head),
//           d(loss)/d(x_hat_L) -> LayerNorm/relu backward -> dz_L                                [train pass]
//   critic: v_out head -> values ; clipped + ValueNorm'd + Huber loss and backward to dz_L
//
// Reference: harl/models/base/act.py:104-157, harl/models/base/distributions.py:7-89,
// harl/algorithms/actors/happo.py:56-91, harl/algorithms/critics/v_critic.py:75-114,
// harl/runners/on_policy_ha_runner.py:66-124.
//
// The head is narrow (act_dim <= 32, value = 1), so it runs on the VALU rather than padding an MFMA
// tile: lane (s,h) holds 64 of the 128 features of sample s (ATL register image); the folded head
// weights sit in LDS as whl[h][R][DAP] so that every lane of a half reads the same address
// (LDS broadcast, ds_read_b128).  The two halves' partial dot products are combined with one
// lane^32 exchange, after which both lanes of a sample redundantly evaluate the (tiny) loss.
// These kernels are HBM-bound: they read x_hat_L once and write dz_L once (1 KiB per sample).
#include "common.h"
#include "../../include/harl_hip.h"

using namespace harl;

#include "heads_common.h"

namespace {

// =============================================================================================
// actor head
// =============================================================================================
template <int H, int DAP, bool DISCRETE, bool TRAIN, bool FUSE = false>
// (min waves per SIMD: 2 for the fused kernels with <= 8 head outputs; wider heads need more than 256 registers -- at
// 256 hipcc spills hundreds of them -- and run one wave per SIMD)
__global__ __launch_bounds__(WG_THREADS, (FUSE && DAP <= 8) ? 2 : 1) void k_actor_head(ActorArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  PHASE_BEGIN();
  float *whl = lds;                    // [2][H/2][DAP]
  float *cst = whl + 2 * (H / 2) * DAP;  // bias, sigma, logsigma, dsigma_dlogstd, rowsum, 1/sigma, 1/sigma^2  [DAP each]
  float *red = cst + 7 * DAP;          // [4][PS_STRIDE]
  float *dwl = red + 4 * PS_STRIDE;    // FUSE: [4 waves][HeadDw::WAVE_FLOATS] staging tiles (16-byte aligned)
  // FUSE with <= 8 head outputs: the head weight gradient accumulates in a wave-private LDS tile (head_dw_step_lds), and
  // the registers that frees hold the NEXT slab's x_hat_L while this slab is being worked on
  constexpr bool LDSACC = FUSE && DAP <= 8;
  constexpr int HROWS = DAP <= 8 ? DAP : 8;
  float *hacc = dwl + WAVES_PER_WG * HeadDw<H>::WAVE_FLOATS;  // LDSACC: [4 waves][HROWS][H]
  if (FUSE)
    for (int e = threadIdx.x; e < WAVES_PER_WG * HeadDw<H>::WAVE_FLOATS; e += WG_THREADS) dwl[e] = 0.f;
  if (LDSACC)
    for (int e = threadIdx.x; e < WAVES_PER_WG * HROWS * H; e += WG_THREADS) hacc[e] = 0.f;
  stage_head<H, DAP>(whl, cst, A.Whp, A.bhp, A.act_dim);
  if (!DISCRETE) {
    for (int e = threadIdx.x; e < DAP; e += WG_THREADS) {
      float sg = 0.5f, sig = 1.f, lsig = 0.f, dsd = 0.f;
      if (e < A.act_dim) {  // distributions.py:86-89: std = sigmoid(log_std / x_coef) * y_coef
        sg = 1.0f / (1.0f + expf(-A.log_std[e] / A.std_x_coef));
        sig = sg * A.std_y_coef;
        lsig = logf(sig);
        dsd = A.std_y_coef * sg * (1.f - sg) / A.std_x_coef;
      }
      cst[DAP + e] = sig;
      cst[2 * DAP + e] = lsig;
      cst[3 * DAP + e] = dsd;
      cst[5 * DAP + e] = 1.0f / sig;
      cst[6 * DAP + e] = 1.0f / (sig * sig);
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const float *whl_h = whl + h * (H / 2) * DAP;
  const int D = A.act_dim;
  const int act_w = DISCRETE ? 1 : D;

  float adv_mean = 0.f, adv_den = 1.f;
  if (TRAIN && A.adv_moments) {  // happo.py:122-127
    const double cnt = A.adv_moments[2];
    const double m = A.adv_moments[0] / cnt;
    const double var = A.adv_moments[1] / cnt - m * m;
    adv_mean = (float)m;
    adv_den = 1.0f / ((float)sqrt(var > 0 ? var : 0.0) + 1e-5f);  // reciprocal (actor_sample multiplies)
  }

  // per-lane partial sums: 0 loss*active, 1 active, 2 ent*active, 3 ratio, 4 count, [8..8+DAP) dlogstd
  float sc[8 + DAP];
#pragma unroll
  for (int k = 0; k < 8 + DAP; ++k) sc[k] = 0.f;
  f32x16 dwacc[(FUSE && !LDSACC) ? H / 32 : 1];
  float dbacc[FUSE ? DAP : 1];
  float *tx = dwl + wave_id() * HeadDw<H>::WAVE_FLOATS, *td = tx + SLAB * HeadDw<H>::HX;
  float *hw = hacc + wave_id() * (HROWS * H);
  if (FUSE) {
    if constexpr (!LDSACC) {
#pragma unroll
      for (int n = 0; n < H / 32; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dwacc[n][r] = 0.f;
    }
#pragma unroll
    for (int d = 0; d < DAP; ++d) dbacc[d] = 0.f;
  }

  // PREF: the NEXT slab's x_hat_L (16 float4 per lane) in flight while this slab's head / loss / backward math runs.
  // Measured on MI355X: with the 64 extra registers the Gaussian / Categorical kernels spill 37-56 registers at the
  // 256-register budget of two waves per SIMD and run SLOWER (0.40 vs 0.31 ms at 819 200 samples), so the actor kernels
  // only take the LDS accumulators (no spills at all); the critic kernel (one head output) has room and prefetches.
  constexpr bool PREF = false;
  const long slab_first = (long)blockIdx.x * WAVES_PER_WG + wave, slab_step = (long)gridDim.x * WAVES_PER_WG;
  f32x4 xnext[PREF ? H / 8 : 1];
  if constexpr (PREF)
    if (slab_first < A.n_slabs) head_load_regs<H>(A.xL, slab_first, lane, xnext);
  // the per-row loss inputs run ONE slab ahead of the arithmetic (a dozen registers; see ActorRow)
  ActorRow<DAP> rnext;
  constexpr int NWM = (H / 2 + 31) / 32;
  uint32_t mb0n = 0u, mb1n = 0u, mbm1n = 0u, mbm2n = 0u;  // ... and so do the ReLU-mask words and the LayerNorm statistic of the backward
  float rstdn = 0.f;
  if (slab_first < A.n_slabs) {
    actor_row_load<DAP, DISCRETE, TRAIN>(A, slab_first, lane, rnext);
    if constexpr (TRAIN) {
      mb0n = A.relu_mask[(slab_first * NWM + 0) * WAVE + lane];
      mb1n = NWM > 1 ? A.relu_mask[(slab_first * NWM + (NWM - 1)) * WAVE + lane] : 0u;
      if constexpr (NWM == 4) {
        mbm1n = A.relu_mask[(slab_first * NWM + 1) * WAVE + lane];
        mbm2n = A.relu_mask[(slab_first * NWM + 2) * WAVE + lane];
      }
      rstdn = A.rstd[slab_first * SLAB + i];
    }
  }
  PHASE(10);
  for (long slab = slab_first; slab < A.n_slabs; slab += slab_step) {
    float z[DAP];
    f32x4 xs[TRAIN ? H / 8 : 1];
    const ActorRow<DAP> rcur = rnext;
    const uint32_t mb0 = mb0n, mb1 = mb1n, mbm1 = mbm1n, mbm2 = mbm2n;
    const float rstd_cur = rstdn;
    {
      const long sn = slab + slab_step < A.n_slabs ? slab + slab_step : slab;
      actor_row_load<DAP, DISCRETE, TRAIN>(A, sn, lane, rnext);
      if constexpr (TRAIN) {
        mb0n = A.relu_mask[(sn * NWM + 0) * WAVE + lane];
        mb1n = NWM > 1 ? A.relu_mask[(sn * NWM + (NWM - 1)) * WAVE + lane] : 0u;
        if constexpr (NWM == 4) {
          mbm1n = A.relu_mask[(sn * NWM + 1) * WAVE + lane];
          mbm2n = A.relu_mask[(sn * NWM + 2) * WAVE + lane];
        }
        rstdn = A.rstd[sn * SLAB + i];
      }
    }
    if constexpr (TRAIN) {
      if constexpr (PREF) {
#pragma unroll
        for (int q = 0; q < H / 8; ++q) xs[q] = xnext[q];
        head_load_regs<H>(A.xL, slab + slab_step < A.n_slabs ? slab + slab_step : slab, lane, xnext);
      } else {
        head_load_regs<H>(A.xL, slab, lane, xs);
      }
      PHASE(0);
      head_fwd_regs<H, DAP>(xs, whl_h, cst, z);
    } else {
      head_fwd_stream<H, DAP>(A.xL, slab, lane, whl_h, cst, z);
    }
    PHASE(1);
    float dzh[DAP];
    float s1, s2;
    if (!actor_sample<DAP, DISCRETE, TRAIN>(A, cst, z, slab, lane, adv_mean, adv_den, sc, dzh, s1, s2, rcur)) continue;
    PHASE(2);
    const long j = slab * SLAB + i;
    if constexpr (FUSE) {  // head weight gradient right here (x_hat_L and dhead are both in registers)
      if constexpr (LDSACC) head_dw_step_lds<H, DAP, HROWS>(xs, dzh, tx, td, lane, hw);
      else head_dw_step<H, DAP>(xs, dzh, tx, td, lane, dwacc);
      if (h == 0) {
#pragma unroll
        for (int d = 0; d < DAP; ++d) dbacc[d] += dzh[d];
      }
    } else {  // head gradients for the dW kernel
      store_dhead<DAP>(A.dhead, slab, lane, dzh);
    }
    PHASE(3);
    if constexpr (TRAIN)
      head_bwd_regs_bits<H, DAP>(xs, mb0, mb1, rstd_cur, slab, lane, whl, dzh, s1, s2, A.dzL, mbm1, mbm2);
    PHASE(4);
  }

  if (TRAIN) {
    if constexpr (DISCRETE) {  // no per-dimension log_std sums: 8 scalars (a 64-wide Categorical head would overrun the row)
      float sc8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) sc8[k] = sc[k];
      block_reduce_store<8>(sc8, red, A.part_scalars + (long)blockIdx.x * PS_STRIDE);
    } else {
      block_reduce_store<8 + DAP>(sc, red, A.part_scalars + (long)blockIdx.x * PS_STRIDE);
    }
  }
  if constexpr (FUSE) {
    float *outp = A.dw_part + (long)blockIdx.x * HeadDw<H>::OUT_FLOATS;
    if constexpr (LDSACC) head_dw_finish_lds<H, DAP, HROWS, WAVES_PER_WG>(hacc, dbacc, dwl, outp);
    else head_dw_finish<H, DAP>(dwacc, dbacc, dwl, outp);
  }
  PHASE(11);
  PHASE_END((TRAIN && FUSE) ? 0 : 1);
}

// =============================================================================================
// critic head
// =============================================================================================

template <int H, bool TRAIN, bool FUSE = false>
__global__ __launch_bounds__(WG_THREADS, FUSE ? 2 : 1) void k_critic_head(CriticArgs A) {
  constexpr int DAP = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *whl = lds;
  float *cst = whl + 2 * (H / 2) * DAP;
  float *red = cst + 5 * DAP;
  float *dwl = red + 4 * PS_STRIDE;
  constexpr bool LDSACC = FUSE;  // one head output: the weight gradient is ONE row, accumulated in wave-private LDS
  constexpr int HROWS = 1;
  float *hacc = dwl + WAVES_PER_WG * HeadDw<H>::WAVE_FLOATS;  // [4 waves][1][H]
  if (FUSE) {
    for (int e = threadIdx.x; e < WAVES_PER_WG * HeadDw<H>::WAVE_FLOATS; e += WG_THREADS) dwl[e] = 0.f;
    for (int e = threadIdx.x; e < WAVES_PER_WG * HROWS * H; e += WG_THREADS) hacc[e] = 0.f;
  }
  stage_head<H, DAP>(whl, cst, A.Whp, A.bhp, 1);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const float *whl_h = whl + h * (H / 2) * DAP;

  float vmean = 0.f, vsd = 1.f;
  if (TRAIN && A.vn_stats) {  // valuenorm.py:38-45
    const float d = fmaxf(A.vn_stats[2], 1e-5f);
    vmean = A.vn_stats[0] / d;
    const float msq = A.vn_stats[1] / d;
    vsd = sqrtf(fmaxf(msq - vmean * vmean, 1e-2f));
  }
  float sc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sc[k] = 0.f;
  float dbacc[FUSE ? DAP : 1];
  float *tx = dwl + wave_id() * HeadDw<H>::WAVE_FLOATS, *td = tx + SLAB * HeadDw<H>::HX;
  float *hw = hacc + wave_id() * (HROWS * H);
  if (FUSE) {
#pragma unroll
    for (int d = 0; d < DAP; ++d) dbacc[d] = 0.f;
  }

  const long slab_first = (long)blockIdx.x * WAVES_PER_WG + wave, slab_step = (long)gridDim.x * WAVES_PER_WG;
  f32x4 xnext[LDSACC ? H / 8 : 1];  // next slab's x_hat_L, in flight during this slab's math (see k_actor_head)
  if constexpr (LDSACC)
    if (slab_first < A.n_slabs) head_load_regs<H>(A.xL, slab_first, lane, xnext);
  // value_preds / returns, ReLU-mask words and the LayerNorm statistic run one slab ahead as well (see ActorRow)
  constexpr int NWM = (H / 2 + 31) / 32;
  float voldn = 0.f, retn = 0.f, rstdn = 0.f;
  uint32_t mb0n = 0u, mb1n = 0u, mbm1n = 0u, mbm2n = 0u;
  if (TRAIN && slab_first < A.n_slabs) {
    critic_row_load(A, slab_first, lane, voldn, retn);
    mb0n = A.relu_mask[(slab_first * NWM + 0) * WAVE + lane];
    mb1n = NWM > 1 ? A.relu_mask[(slab_first * NWM + (NWM - 1)) * WAVE + lane] : 0u;
    if constexpr (NWM == 4) {
      mbm1n = A.relu_mask[(slab_first * NWM + 1) * WAVE + lane];
      mbm2n = A.relu_mask[(slab_first * NWM + 2) * WAVE + lane];
    }
    rstdn = A.rstd[slab_first * SLAB + i];
  }
  for (long slab = slab_first; slab < A.n_slabs; slab += slab_step) {
    float z[DAP];
    f32x4 xs[TRAIN ? H / 8 : 1];
    const float vold = voldn, ret = retn, rstd_cur = rstdn;
    const uint32_t mb0 = mb0n, mb1 = mb1n, mbm1 = mbm1n, mbm2 = mbm2n;
    if constexpr (TRAIN) {
      const long sn = slab + slab_step < A.n_slabs ? slab + slab_step : slab;
      critic_row_load(A, sn, lane, voldn, retn);
      mb0n = A.relu_mask[(sn * NWM + 0) * WAVE + lane];
      mb1n = NWM > 1 ? A.relu_mask[(sn * NWM + (NWM - 1)) * WAVE + lane] : 0u;
      if constexpr (NWM == 4) {
        mbm1n = A.relu_mask[(sn * NWM + 1) * WAVE + lane];
        mbm2n = A.relu_mask[(sn * NWM + 2) * WAVE + lane];
      }
      rstdn = A.rstd[sn * SLAB + i];
    }
    if constexpr (TRAIN) {
      if constexpr (LDSACC) {
#pragma unroll
        for (int q = 0; q < H / 8; ++q) xs[q] = xnext[q];
        head_load_regs<H>(A.xL, slab + slab_step < A.n_slabs ? slab + slab_step : slab, lane, xnext);
      } else {
        head_load_regs<H>(A.xL, slab, lane, xs);
      }
      head_fwd_regs<H, DAP>(xs, whl_h, cst, z);
    } else {
      head_fwd_stream<H, DAP>(A.xL, slab, lane, whl_h, cst, z);
    }
    float dv;
    if (!critic_sample<TRAIN>(A, z[0], slab, lane, vmean, vsd, sc, dv, vold, ret)) continue;
    const float v = z[0];
    const long j = slab * SLAB + i;
    const float dzh[DAP] = {dv, 0.f, 0.f, 0.f};
    if constexpr (FUSE) {
      head_dw_step_lds<H, DAP, HROWS>(xs, dzh, tx, td, lane, hw);
      if (h == 0) dbacc[0] += dv;
    } else {
      float *dh = A.dhead + j * DHEAD_LD + 16 * h;
#pragma unroll
      for (int c = 0; c < 16; ++c) dh[c] = (h == 0 && c == 0) ? dv : 0.f;
    }
    if constexpr (TRAIN)
      head_bwd_regs_bits<H, DAP>(xs, mb0, mb1, rstd_cur, slab, lane, whl, dzh, dv * cst[4 * DAP], dv * (v - cst[0]), A.dzL, mbm1, mbm2);
  }
  if (TRAIN) block_reduce_store<8>(sc, red, A.part_scalars + (long)blockIdx.x * PS_STRIDE);
  if constexpr (FUSE)
    head_dw_finish_lds<H, DAP, HROWS, WAVES_PER_WG>(hacc, dbacc, dwl, A.dw_part + (long)blockIdx.x * HeadDw<H>::OUT_FLOATS);
}

// =============================================================================================
// HATRPO Fisher-vector product, head part (harl/utils/trpo_util.py:132-158).  At theta_new == theta_old the
// Hessian of KL(old || new) is the Gauss-Newton matrix J^T M J (the first-order terms vanish identically), so
//   F v = J^T M (J v):   J v = forward-mode tangent of the head outputs (trunk tangent from harl_mlp_tangent_*),
//   M = diag(1 / sigma^2) for the Gaussian mean  (the log_std block is diagonal and handled by the caller),
//   M = identity on the normalised logits q = z - logsumexp(z) for `kl_approx` (trpo_util.py:47-51,83-86),
//       over ALL action entries, masked ones included (q_masked = -1e10 - lse still depends on theta through lse),
// followed by the ordinary backward pass.  This kernel: tangent of the head, M, backward through the head and the
// last LayerNorm/ReLU -> dz_L (ATL) + dhead (for the head dW); the caller divides by the batch size (kl.mean()).
// =============================================================================================
struct FvpArgs {
  const float *xL, *xLdot;
  const uint32_t *relu_mask;
  const float *rstd;
  long M;
  const float *Whp, *bhp, *Whdp, *bhdp, *log_std;
  float std_x_coef, std_y_coef;
  int act_dim;
  const float *avail;
  float *dzL, *dhead;
  long n_slabs;
  long m_valid, m_pad;  // recurrent batches: rows j with (j % m_pad) >= m_valid are padding sequences
};

template <int H, int DAP, bool DISCRETE>
__global__ __launch_bounds__(WG_THREADS) void k_actor_head_fvp(FvpArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *whl = lds;                       // [2][H/2][DAP]  Wh'
  float *cst = whl + 2 * (H / 2) * DAP;   // bias, sigma, logsigma, dsig, rowsum
  float *wdl = cst + 5 * DAP;             // [2][H/2][DAP]  Wh'_dot
  float *cdt = wdl + 2 * (H / 2) * DAP;   // bias_dot (+ unused slots)
  stage_head<H, DAP>(whl, cst, A.Whp, A.bhp, A.act_dim);
  stage_head<H, DAP>(wdl, cdt, A.Whdp, A.bhdp, A.act_dim);
  if (!DISCRETE) {
    for (int e = threadIdx.x; e < DAP; e += WG_THREADS) {
      float sig = 1.f;
      if (e < A.act_dim) sig = A.std_y_coef / (1.0f + expf(-A.log_std[e] / A.std_x_coef));
      cst[DAP + e] = sig;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const float *whl_h = whl + h * (H / 2) * DAP;
  const float *wdl_h = wdl + h * (H / 2) * DAP;
  const int D = A.act_dim;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < A.n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    // z = Wh' x_hat + b ;  zd = Wh' x_hat_dot + Wh'_dot x_hat + b_dot   (one sweep over both ATL images)
    float z[DAP], zd[DAP];
#pragma unroll
    for (int d = 0; d < DAP; ++d) z[d] = zd[d] = 0.f;
    const f32x4 *xp = reinterpret_cast<const f32x4 *>(A.xL + slab * (long)(H * SLAB)) + lane;
    const f32x4 *xdp = reinterpret_cast<const f32x4 *>(A.xLdot + slab * (long)(H * SLAB)) + lane;
#pragma unroll 1
    for (int q = 0; q < H / 8; ++q) {
      const f32x4 xv = xp[q * WAVE], xd = xdp[q * WAVE];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int dq = 0; dq < DAP / 4; ++dq) {
          const f32x4 w = *reinterpret_cast<const f32x4 *>(whl_h + (4 * q + c) * DAP + 4 * dq);
          const f32x4 wd = *reinterpret_cast<const f32x4 *>(wdl_h + (4 * q + c) * DAP + 4 * dq);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            z[4 * dq + k] += xv[c] * w[k];
            zd[4 * dq + k] += xd[c] * w[k] + xv[c] * wd[k];
          }
        }
      }
    }
    float zlin[DAP];
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      zlin[d] = wave_sum32(z[d]);
      z[d] = zlin[d] + cst[d];
      zd[d] = wave_sum32(zd[d]) + cdt[d];
    }
    const long j = slab * SLAB + i;
    const bool valid = j < A.M && (A.m_pad == 0 || (j % A.m_pad) < A.m_valid);
    float dzh[DAP];
#pragma unroll
    for (int d = 0; d < DAP; ++d) dzh[d] = 0.f;
    if (!DISCRETE) {
#pragma unroll
      for (int d = 0; d < DAP; ++d)
        if (d < D && valid) {
          const float sig = cst[DAP + d];
          dzh[d] = zd[d] / (sig * sig);
        }
    } else {
      const long jc = valid ? j : A.M - 1;
      float mx = -3.0e38f;
      bool un[DAP];
#pragma unroll
      for (int d = 0; d < DAP; ++d) {
        un[d] = d < D && !(A.avail && A.avail[jc * D + d] == 0.f);
        if (d < D) {
          if (!un[d]) z[d] = -1e10f;
          mx = fmaxf(mx, z[d]);
        }
      }
      float se = 0.f;
#pragma unroll
      for (int d = 0; d < DAP; ++d)
        if (d < D) se += expf(z[d] - mx);
      const float lse = mx + logf(se);
      float pr[DAP], S = 0.f, sum_zd = 0.f;
#pragma unroll
      for (int d = 0; d < DAP; ++d) {
        pr[d] = d < D ? expf(z[d] - lse) : 0.f;
        if (un[d]) {
          S += pr[d] * zd[d];
          sum_zd += zd[d];
        }
      }
      const float sum_dq = sum_zd - (float)D * S;  // sum over ALL D entries of q_dot_d = [unmasked] zd_d - S
#pragma unroll
      for (int d = 0; d < DAP; ++d)
        if (un[d] && valid) dzh[d] = (zd[d] - S) - pr[d] * sum_dq;  // log-softmax backward of dq = q_dot
    }
    store_dhead<DAP>(A.dhead, slab, lane, dzh);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      s1 += dzh[d] * cst[4 * DAP + d];
      s2 += dzh[d] * zlin[d];
    }
    head_bwd_stream<H, DAP>(A.xL, A.relu_mask, A.rstd[slab * SLAB + i], slab, lane, whl_h, dzh, s1, s2, A.dzL);
  }
}

// sum over samples of KL(old || new) per trpo_util.py:47-62,65-92 (Gaussian in fp64, Categorical kl_approx in fp32)
__global__ __launch_bounds__(256) void k_trpo_kl(const float *__restrict__ ho, const float *__restrict__ hn,
                                                 const float *__restrict__ ls_old, const float *__restrict__ ls_new,
                                                 float xc, float yc, long M, int D, int discrete,
                                                 double *__restrict__ out) {
  double acc = 0;
  for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < M; s += (long)gridDim.x * blockDim.x) {
    if (discrete) {
      float k = 0.f;
      for (int d = 0; d < D; ++d) {
        const float pp = ho[s * D + d], q = hn[s * D + d];
        k += (expf(q - pp) - 1.f - q) + pp;
      }
      acc += (double)k;
    } else {
      double k = 0;
      for (int d = 0; d < D; ++d) {
        const float sp = yc / (1.0f + expf(-ls_old[d] / xc)), sq = yc / (1.0f + expf(-ls_new[d] / xc));
        const double vr = ((double)sp / (double)sq) * ((double)sp / (double)sq);
        const double t = ((double)ho[s * D + d] - (double)hn[s * D + d]) / (double)sq;
        k += 0.5 * (vr + t * t - 1.0 - log(vr));
      }
      acc += k;
    }
  }
  acc = wave_reduce_sum_d(acc);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

// tangent of the LayerNorm-affine fold:  Wp_dot = W_dot*g + W*g_dot ;  bp_dot = b_dot + W_dot.beta + W.beta_dot
__global__ __launch_bounds__(64) void k_fold_tangent(const float *__restrict__ W, const float *__restrict__ g,
                                                     const float *__restrict__ be, const float *__restrict__ Wd,
                                                     const float *__restrict__ bd, const float *__restrict__ gd,
                                                     const float *__restrict__ bed, float *__restrict__ Wpd,
                                                     float *__restrict__ bpd, int in_dim) {
  const int o = blockIdx.x;
  float acc = 0.f;
  for (int k = threadIdx.x; k < in_dim; k += 64) {
    const float w = W[(long)o * in_dim + k], wd = Wd[(long)o * in_dim + k];
    Wpd[(long)o * in_dim + k] = g ? wd * g[k] + w * gd[k] : wd;
    if (be) acc += wd * be[k] + w * bed[k];
  }
  acc = wave_reduce_sum(acc);
  if (threadIdx.x == 0) bpd[o] = bd[o] + acc;
}

template <int H, int DAP, bool DISC>
void launch_fvp(const FvpArgs &A, int grid, hipStream_t s) {
  const size_t shm = ((size_t)4 * (H / 2) * DAP + 10 * DAP) * sizeof(float);
  allow_big_lds(k_actor_head_fvp<H, DAP, DISC>, shm);  // (256-wide layers with 32 outputs: 65 KiB)
  hipLaunchKernelGGL((k_actor_head_fvp<H, DAP, DISC>), dim3(grid), dim3(WG_THREADS), shm, s, A);
}

int head_grid(long M) { return persistent_grid(n_slabs_of(M), 4); }

template <int H, int DAP, bool DISC, bool TRAIN>
void launch_actor(const ActorArgs &A, int grid, hipStream_t s) {
  const size_t base = ((size_t)2 * (H / 2) * DAP + 7 * DAP + 4 * PS_STRIDE) * sizeof(float);
  if constexpr (TRAIN && DAP <= 32) {  // (the fused head dW handles up to 32 outputs; wider heads write dhead)
    if (A.dw_part) {
      size_t fl = (size_t)WAVES_PER_WG * HeadDw<H>::WAVE_FLOATS;
      if (fl < (size_t)HeadDw<H>::OUT_FLOATS) fl = HeadDw<H>::OUT_FLOATS;
      if (DAP <= 8) fl = (size_t)WAVES_PER_WG * HeadDw<H>::WAVE_FLOATS + (size_t)WAVES_PER_WG * DAP * H;  // + LDS accumulators
      const size_t shm = base + fl * sizeof(float);
      allow_big_lds(k_actor_head<H, DAP, DISC, true, true>, shm);
      hipLaunchKernelGGL((k_actor_head<H, DAP, DISC, true, true>), dim3(grid), dim3(WG_THREADS), shm, s, A);
      return;
    }
  }
  hipLaunchKernelGGL((k_actor_head<H, DAP, DISC, TRAIN>), dim3(grid), dim3(WG_THREADS), base, s, A);
}

template <bool TRAIN>
int dispatch_actor(const ActorArgs &A, int H, int discrete, int grid, hipStream_t s) {
  const int D = A.act_dim;
  if (D < 1 || D > 64 || (D > 32 && !discrete)) {
    set_error("actor head: act_dim must be in [1, 32] (Categorical heads: [1, 64])");
    return -2;
  }
  if (D > 32 && A.dw_part) {
    set_error("actor head: the fused head weight gradient stops at 32 outputs (pass dw_part = NULL and run harl_mlp_dw_partials on the ATL(64) dhead image)");
    return -2;
  }
  const int dap = D <= 4 ? 4 : (D <= 8 ? 8 : (D <= 16 ? 16 : (D <= 32 ? 32 : 64)));
#define CASE(Hv, DAPv)                                                    \
  if (H == Hv && dap == DAPv) {                                           \
    if (discrete) launch_actor<Hv, DAPv, true, TRAIN>(A, grid, s);        \
    else launch_actor<Hv, DAPv, false, TRAIN>(A, grid, s);                \
    return check_launch("harl_actor_head");                               \
  }
  CASE(128, 4) CASE(128, 8) CASE(128, 16) CASE(128, 32) CASE(64, 4) CASE(64, 8) CASE(64, 16) CASE(64, 32)
  CASE(256, 4) CASE(256, 8) CASE(256, 16) CASE(256, 32)  // hidden width 256 (csrc/panel.hip)
#undef CASE
  if (dap == 64 && (H == 64 || H == 128)) {  // Categorical only (checked above)
    if (H == 64) launch_actor<64, 64, true, TRAIN>(A, grid, s);
    else launch_actor<128, 64, true, TRAIN>(A, grid, s);
    return check_launch("harl_actor_head");
  }
  set_error("actor head: hidden width must be 64 or 128");
  return -2;
}
}  // namespace

HARL_PHASE_ACCESSOR(heads)

extern "C" int harl_head_blocks(long M) { return head_grid(M); }

extern "C" int harl_actor_head_logp(const float *xL, long M, int H, const float *Whp, const float *bhp,
                                    const float *log_std, float std_x_coef, float std_y_coef, int discrete,
                                    int act_dim, const float *actions, const float *avail, float *logp_out,
                                    const float *old_logp, float *factor, int agg_mean, float *head_out, long m_valid,
                                    long m_pad, void *stream) {
  if (M <= 0) return 0;
  ActorArgs A{};
  A.head_out = head_out;
  A.m_valid = m_valid; A.m_pad = m_pad;
  A.xL = xL; A.M = M; A.Whp = Whp; A.bhp = bhp; A.log_std = log_std;
  A.std_x_coef = std_x_coef; A.std_y_coef = std_y_coef; A.act_dim = act_dim;
  A.actions = actions; A.avail = avail; A.old_logp = old_logp; A.agg_mean = agg_mean;
  A.logp_out = logp_out; A.factor_out = factor; A.n_slabs = n_slabs_of(M);
  if (factor && !old_logp) {
    set_error("harl_actor_head_logp: factor update needs old_logp");
    return -2;
  }
  return dispatch_actor<false>(A, H, discrete, head_grid(M), (hipStream_t)stream);
}

extern "C" int harl_actor_head_loss(const float *xL, const uint32_t *relu_mask, const float *rstd, long M, int H,
                                    const float *Whp, const float *bhp, const float *log_std, float std_x_coef,
                                    float std_y_coef, int discrete, int act_dim, const int64_t *idx,
                                    const float *actions, const float *avail, const float *old_logp, const float *adv,
                                    const double *adv_moments, const float *factor, const float *active,
                                    double clip_param, float entropy_coef, int agg_mean, int trpo, long m_valid,
                                    long m_pad, float *logp_out, float *dzL, float *dhead, float *part_scalars,
                                    float *dw_part, int n_wg, void *stream) {
  if (M <= 0) return 0;
  ActorArgs A{};
  A.trpo = trpo;
  A.logp_out = logp_out;
  A.dw_part = dw_part;
  if (dw_part && n_wg <= 0) { set_error("harl_actor_head_loss: fused head gradient needs n_wg > 0"); return -2; }
  A.m_valid = m_valid; A.m_pad = m_pad;
  A.xL = xL; A.relu_mask = relu_mask; A.rstd = rstd; A.M = M; A.Whp = Whp; A.bhp = bhp; A.log_std = log_std;
  A.std_x_coef = std_x_coef; A.std_y_coef = std_y_coef; A.act_dim = act_dim; A.idx = idx;
  A.actions = actions; A.avail = avail; A.old_logp = old_logp; A.adv = adv; A.adv_moments = adv_moments;
  A.factor_in = factor; A.active = active; A.entropy_coef = entropy_coef;
  A.clip_lo = (float)(1.0 - clip_param); A.clip_hi = (float)(1.0 + clip_param);  // torch.clamp(imp, 1 - c, 1 + c): Python doubles
  A.agg_mean = agg_mean; A.dzL = dzL; A.dhead = dhead; A.part_scalars = part_scalars; A.n_slabs = n_slabs_of(M);
  return dispatch_actor<true>(A, H, discrete, dw_part ? n_wg : head_grid(M), (hipStream_t)stream);
}

extern "C" int harl_critic_head_values(const float *xL, long M, int H, const float *Whp, const float *bhp,
                                       float *values, void *stream) {
  if (M <= 0) return 0;
  CriticArgs A{};
  A.xL = xL; A.M = M; A.Whp = Whp; A.bhp = bhp; A.values_out = values; A.n_slabs = n_slabs_of(M);
  const int grid = head_grid(M);
  const size_t shm = ((size_t)2 * (H / 2) * 4 + 20 + 4 * PS_STRIDE) * sizeof(float);
  if (H == 128) hipLaunchKernelGGL((k_critic_head<128, false>), dim3(grid), dim3(WG_THREADS), shm, (hipStream_t)stream, A);
  else if (H == 64) hipLaunchKernelGGL((k_critic_head<64, false>), dim3(grid), dim3(WG_THREADS), shm, (hipStream_t)stream, A);
  else if (H == 256) hipLaunchKernelGGL((k_critic_head<256, false>), dim3(grid), dim3(WG_THREADS), shm, (hipStream_t)stream, A);
  else { set_error("critic head: hidden width must be 64, 128 or 256"); return -2; }
  return check_launch("harl_critic_head_values");
}

extern "C" int harl_critic_head_loss(const float *xL, const uint32_t *relu_mask, const float *rstd, long M, int H,
                                     const float *Whp, const float *bhp, const int64_t *idx, const float *value_preds,
                                     const float *returns, const float *vn_stats, float clip_param, int use_clipped,
                                     int use_huber, float huber_delta, long m_valid, long m_pad, float *dzL, float *dhead,
                                     float *part_scalars, float *dw_part, int n_wg, void *stream) {
  if (M <= 0) return 0;
  CriticArgs A{};
  A.m_valid = m_valid; A.m_pad = m_pad;
  A.xL = xL; A.relu_mask = relu_mask; A.rstd = rstd; A.M = M; A.Whp = Whp; A.bhp = bhp; A.idx = idx;
  A.value_preds = value_preds; A.returns = returns; A.vn_stats = vn_stats; A.clip_param = clip_param;
  A.huber_delta = huber_delta; A.use_clipped = use_clipped; A.use_huber = use_huber; A.dzL = dzL; A.dhead = dhead;
  A.part_scalars = part_scalars; A.n_slabs = n_slabs_of(M);
  A.dw_part = dw_part;
  if (dw_part && n_wg <= 0) { set_error("harl_critic_head_loss: fused head gradient needs n_wg > 0"); return -2; }
  const int grid = dw_part ? n_wg : head_grid(M);
  const size_t base = ((size_t)2 * (H / 2) * 4 + 20 + 4 * PS_STRIDE) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define CL(Hv)                                                                                                     \
  if (H == Hv) {                                                                                                   \
    if (dw_part) {                                                                                                 \
      const size_t fl = (size_t)WAVES_PER_WG * HeadDw<Hv>::WAVE_FLOATS + (size_t)WAVES_PER_WG * Hv;  /* + LDS acc */  \
      const size_t shm = base + fl * sizeof(float);                                                                \
      allow_big_lds(k_critic_head<Hv, true, true>, shm);                                                           \
      hipLaunchKernelGGL((k_critic_head<Hv, true, true>), dim3(grid), dim3(WG_THREADS), shm, s, A);                \
    } else {                                                                                                       \
      hipLaunchKernelGGL((k_critic_head<Hv, true>), dim3(grid), dim3(WG_THREADS), base, s, A);                     \
    }                                                                                                              \
    return check_launch("harl_critic_head_loss");                                                                  \
  }
  CL(128) CL(64) CL(256)
#undef CL
  set_error("critic head: hidden width must be 64 or 128");
  return -2;
}

extern "C" int harl_actor_head_fvp(const float *xL, const float *xLdot, const uint32_t *relu_mask, const float *rstd, long M,
                                   int H, const float *Whp, const float *bhp, const float *Whdp, const float *bhdp,
                                   const float *log_std, float std_x_coef, float std_y_coef, int discrete, int act_dim,
                                   const float *avail, long m_valid, long m_pad, float *dzL, float *dhead, void *stream) {
  if (M <= 0) return 0;
  FvpArgs A{};
  A.m_valid = m_valid;
  A.m_pad = m_pad;
  A.xL = xL; A.xLdot = xLdot; A.relu_mask = relu_mask; A.rstd = rstd; A.M = M; A.Whp = Whp; A.bhp = bhp; A.Whdp = Whdp;
  A.bhdp = bhdp; A.log_std = log_std; A.std_x_coef = std_x_coef; A.std_y_coef = std_y_coef; A.act_dim = act_dim;
  A.avail = avail; A.dzL = dzL; A.dhead = dhead; A.n_slabs = n_slabs_of(M);
  const int grid = head_grid(M);
  hipStream_t s = (hipStream_t)stream;
  if (act_dim < 1 || act_dim > 64 || (act_dim > 32 && !discrete)) {
    set_error("head fvp: act_dim must be in [1, 32] (Categorical heads: [1, 64])");
    return -2;
  }
  const int dap = act_dim <= 4 ? 4 : (act_dim <= 8 ? 8 : (act_dim <= 16 ? 16 : (act_dim <= 32 ? 32 : 64)));
  if (dap == 64 && (H == 64 || H == 128)) {
    if (H == 64) launch_fvp<64, 64, true>(A, grid, s);
    else launch_fvp<128, 64, true>(A, grid, s);
    return check_launch("harl_actor_head_fvp");
  }
#define CASE(Hv, DAPv)                                          \
  if (H == Hv && dap == DAPv) {                                 \
    if (discrete) launch_fvp<Hv, DAPv, true>(A, grid, s);       \
    else launch_fvp<Hv, DAPv, false>(A, grid, s);               \
    return check_launch("harl_actor_head_fvp");                 \
  }
  CASE(128, 4) CASE(128, 8) CASE(128, 16) CASE(128, 32) CASE(64, 4) CASE(64, 8) CASE(64, 16) CASE(64, 32)
  CASE(256, 4) CASE(256, 8) CASE(256, 16) CASE(256, 32)  // hidden width 256 (csrc/panel.hip)
#undef CASE
  set_error("head fvp: hidden width must be 64, 128 or 256");
  return -2;
}

extern "C" int harl_trpo_kl_sum(const float *head_old, const float *head_new, const float *log_std_old,
                                const float *log_std_new, float std_x_coef, float std_y_coef, long M, int act_dim,
                                int discrete, double *out_sum, void *stream) {
  if (M <= 0) return 0;
  long nb = (M + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_trpo_kl, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, head_old, head_new, log_std_old,
                     log_std_new, std_x_coef, std_y_coef, M, act_dim, discrete, out_sum);
  return check_launch("harl_trpo_kl_sum");
}

extern "C" int harl_fold_linear_tangent(const float *W, const float *gamma, const float *beta, const float *Wd,
                                        const float *bd, const float *gammad, const float *betad, float *Wpd,
                                        float *bpd, int out_dim, int in_dim, void *stream) {
  hipLaunchKernelGGL(k_fold_tangent, dim3(out_dim), dim3(64), 0, (hipStream_t)stream, W, gamma, beta, Wd, bd, gammad,
                     betad, Wpd, bpd, in_dim);
  return check_launch("harl_fold_linear_tangent");
}
