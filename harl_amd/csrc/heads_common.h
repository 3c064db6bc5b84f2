// heads_common.h -- pieces of the action / value heads shared by heads.hip (stand-alone head kernels) and update.hip (the
// fused forward + loss kernel): argument blocks, head weight staging, the VALU head forward, the MFMA head backward, the
// per-sample loss arithmetic (actor_sample / critic_sample) and the block-level scalar reduction.
// Reference: harl/models/base/act.py:104-157, harl/models/base/distributions.py:7-89,
// harl/algorithms/actors/happo.py:56-91, harl/algorithms/critics/v_critic.py:75-114.
#pragma once
#include "common.h"

using namespace harl;

namespace {
constexpr float LOG_SQRT_2PI = 0.918938533204672741780329736406f;
constexpr float HALF_LOG_2PI_PLUS_HALF = 1.418938533204672741780329736406f;

struct ActorArgs {
  const float *xL;
  const uint32_t *relu_mask;
  const float *rstd;
  long M;
  const float *Whp, *bhp, *log_std;
  float std_x_coef, std_y_coef;
  int act_dim;
  const int64_t *idx;
  const float *actions, *avail, *old_logp, *adv;
  const double *adv_moments;
  const float *factor_in, *active;
  float clip_lo, clip_hi, entropy_coef;  // clip bounds 1 -+ clip_param, formed in double by the launcher
  int agg_mean;
  float *dzL, *dhead, *part_scalars;
  float *dw_part;   // fused head weight gradient: per-workgroup partials [gridDim.x][32*H + 32] (NULL: write dhead instead)
  float *logp_out, *factor_out;
  float *head_out;  // [M, act_dim]: Gaussian mean / normalised Categorical logits (rollout sampling, HATRPO KL)
  long m_valid, m_pad;  // recurrent batches: row j counts only if (j % m_pad) < m_valid   (m_pad = 0: every j < M)
  int trpo;         // surrogate: 0 HAPPO (clipped, happo.py:71-85); 1 HATRPO +ratio*f*adv*active, no entropy term
                    // (hatrpo.py:82-90); 2 HAA2C -ratio*f*adv*active, no clip (haa2c.py:70-80)
  long n_slabs;
};

template <int H, int DAP>
__device__ __forceinline__ void stage_head(float *whl, float *cst, const float *__restrict__ Whp,
                                           const float *__restrict__ bhp, int act_dim) {
  // whl[hh][R][d] = Whp[d][f(R,hh)] ; cst[0..DAP) = bias
  for (int e = threadIdx.x; e < 2 * (H / 2) * DAP; e += WG_THREADS) {
    const int d = e % DAP, R = (e / DAP) % (H / 2), hh = e / (DAP * (H / 2));
    const int f = feat_base(R) + 4 * hh;
    whl[e] = d < act_dim ? Whp[d * H + f] : 0.f;
  }
  for (int e = threadIdx.x; e < DAP; e += WG_THREADS) {
    cst[e] = e < act_dim ? bhp[e] : 0.f;
    float rs = 0.f;  // row sum of the folded head weights: mean_f(dx_hat) needs no per-feature pass (see head_bwd_stream)
    if (e < act_dim)
      for (int f = 0; f < H; ++f) rs += Whp[e * H + f];
    cst[4 * DAP + e] = rs;
  }
}

// z[d] = bias[d] + sum_f x_hat[f] * Whp[d][f].  x_hat is streamed from the ATL image one float4 per
// lane at a time (prefetched one step ahead) in a *rolled* loop: nothing but the DAP accumulators
// stays live, so these HBM-bound kernels keep a small register footprint / high occupancy.
template <int H, int DAP>
__device__ __forceinline__ void head_fwd_stream(const float *__restrict__ xL, long slab, int lane,
                                                const float *whl_h, const float *cst, float (&z)[DAP]) {
#pragma unroll
  for (int d = 0; d < DAP; ++d) z[d] = 0.f;
  const f32x4 *xp = reinterpret_cast<const f32x4 *>(xL + slab * (long)(H * SLAB)) + lane;
  f32x4 xv = xp[0];
#pragma unroll 1
  for (int q = 0; q < H / 8; ++q) {
    const f32x4 xn = xp[(q + 1 < H / 8 ? q + 1 : q) * WAVE];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int dq = 0; dq < DAP / 4; ++dq) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(whl_h + (4 * q + c) * DAP + 4 * dq);
        z[4 * dq + 0] += xv[c] * w[0];
        z[4 * dq + 1] += xv[c] * w[1];
        z[4 * dq + 2] += xv[c] * w[2];
        z[4 * dq + 3] += xv[c] * w[3];
      }
    }
    xv = xn;
  }
#pragma unroll
  for (int d = 0; d < DAP; ++d) z[d] = wave_sum32(z[d]) + cst[d];
}

// dz_L = relu_mask ? rstd (dx_hat - mean_f(dx_hat) - x_hat mean_f(dx_hat x_hat)) : 0  with
// dx_hat[f] = sum_d dzh[d] Whp[d][f].  Both feature means are closed forms of head-level quantities:
//   mean_f(dx_hat)       = sum_d dzh[d] * rowsum(Whp[d]) / H
//   mean_f(dx_hat x_hat) = sum_d dzh[d] * (z[d] - bias[d]) / H        (z = head forward output)
// so the backward is a single streaming pass over x_hat (second read; L2 / Infinity-Cache resident).
template <int H, int DAP>
__device__ __forceinline__ void head_bwd_stream(const float *__restrict__ xL, const uint32_t *__restrict__ mask_in,
                                                float rstd, long slab, int lane, const float *whl_h,
                                                const float (&dzh)[DAP], float s1, float s2,
                                                float *__restrict__ dz_out) {
  constexpr int NW = (H / 2 + 31) / 32;
  uint32_t b0 = mask_in[(slab * NW + 0) * WAVE + lane];
  uint32_t b1 = NW > 1 ? mask_in[(slab * NW + (NW - 1)) * WAVE + lane] : 0u;
  s1 *= (1.0f / H);
  s2 *= (1.0f / H);
  const f32x4 *xp = reinterpret_cast<const f32x4 *>(xL + slab * (long)(H * SLAB)) + lane;
  f32x4 *op = reinterpret_cast<f32x4 *>(dz_out + slab * (long)(H * SLAB)) + lane;
  f32x4 xv = xp[0];
#pragma unroll 1
  for (int q = 0; q < H / 8; ++q) {
    const f32x4 xn = xp[(q + 1 < H / 8 ? q + 1 : q) * WAVE];
    if (q > 0 && (q & 7) == 0) b0 = q == 8 * (NW - 1) ? b1 : mask_in[(slab * NW + (q >> 3)) * WAVE + lane];  // next mask word; bits are consumed MSB-first
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float dx = 0.f;
#pragma unroll
      for (int dq = 0; dq < DAP / 4; ++dq) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(whl_h + (4 * q + c) * DAP + 4 * dq);
        dx += dzh[4 * dq + 0] * w[0] + dzh[4 * dq + 1] * w[1] + dzh[4 * dq + 2] * w[2] + dzh[4 * dq + 3] * w[3];
      }
      const float da = rstd * (dx - s1 - xv[c] * s2);
      o[c] = (int)b0 < 0 ? da : 0.f;  // MSB-first mask (common.h); plain C here: these kernels are HBM-bound and
      b0 <<= 1;                        // the asm form of mask_pop only restricts the scheduler
    }
    op[q * WAVE] = o;
    xv = xn;
  }
}

// ---- training kernels: x_hat_L is needed twice (head forward, then LayerNorm backward), so it is loaded ONCE, as a
// burst of H/8 independent float4 loads per lane (64 VGPRs for H = 128), and stays in registers across the loss.
// The loops are fully unrolled (static register indexing); sched_barrier(0) after every q-step keeps hipcc from hoisting
// the broadcast LDS weight reads of later steps (which is what spilled the first unrolled version of this kernel).
template <int H>
__device__ __forceinline__ void head_load_regs(const float *__restrict__ xL, long slab, int lane, f32x4 (&xs)[H / 8]) {
  const f32x4 *xp = reinterpret_cast<const f32x4 *>(xL + slab * (long)(H * SLAB)) + lane;
#pragma unroll
  for (int q = 0; q < H / 8; ++q) xs[q] = xp[q * WAVE];
}

// DA = number of head outputs that exist (<= DAP, the padded width of the LDS image): the kernels that are instantiated per
// action width (update.hip) multiply only those columns -- 5 x 64 instead of 8 x 64 FMAs per lane for a Box(5) policy
template <int H, int DAP, int DA = DAP>
__device__ __forceinline__ void head_fwd_regs(const f32x4 (&xs)[H / 8], const float *whl_h, const float *cst,
                                              float (&z)[DAP]) {
#pragma unroll
  for (int d = 0; d < DAP; ++d) z[d] = 0.f;
#pragma unroll
  for (int q = 0; q < H / 8; ++q) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int dq = 0; dq < (DA + 3) / 4; ++dq) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(whl_h + (4 * q + c) * DAP + 4 * dq);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * dq + e < DA) z[4 * dq + e] += xs[q][c] * w[e];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int d = 0; d < DA; ++d) z[d] = wave_sum32(z[d]) + cst[d];
}

// ---- the head weights as a TRANSPOSED image whlT[h][d][R] (the fused kernel of update.hip; rows d < DAE = DA rounded up to
// even, the odd padding row zero; lane half h at h * head_t_block(H, DA) floats: 16 floats of padding keep the two halves of an
// MFMA operand read on different banks).  Forward: one ds_read_b128 = four features of ONE output -- DA x H/8 reads per
// lane, all of them used (the [R][DAP] image above costs 2 x H/2 reads at 4 < DA <= 8 and discards the padding), requested one
// q-step ahead of the FMAs that consume them: the phase was bound by exposed LDS latency, not by its FMAs (phase timers,
// round 4: 14 % of the fused kernel for 320 FMAs per lane).  Backward: head_bwd_regs_bits<..., WT = true> reads its MFMA A
// operand from the same image, so the kernel holds the head weights once.
constexpr int HEAD_T_PAD = 16;
__host__ __device__ constexpr int head_t_rows(int da) { return 2 * ((da + 1) / 2); }
__host__ __device__ constexpr int head_t_block(int H, int da) { return head_t_rows(da) * (H / 2) + HEAD_T_PAD; }

template <int H, int DAP, int DA>
__device__ __forceinline__ void head_fwd_regs_t(const f32x4 (&xs)[H / 8], const float *whlT_h, const float *cst,
                                                float (&z)[DAP]) {
#pragma unroll
  for (int d = 0; d < DAP; ++d) z[d] = 0.f;
  f32x4 w[2][DA];
#pragma unroll
  for (int d = 0; d < DA; ++d) w[0][d] = *reinterpret_cast<const f32x4 *>(whlT_h + d * (H / 2));
#pragma unroll
  for (int q = 0; q < H / 8; ++q) {
    if (q + 1 < H / 8) {
#pragma unroll
      for (int d = 0; d < DA; ++d) w[(q + 1) & 1][d] = *reinterpret_cast<const f32x4 *>(whlT_h + d * (H / 2) + 4 * (q + 1));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int d = 0; d < DA; ++d) z[d] += xs[q][c] * w[q & 1][d][c];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int d = 0; d < DA; ++d) z[d] = wave_sum32(z[d]) + cst[d];
}

// whlT[hh][d][R] = Whp[d][f(R,hh)]; cst[0..DAP) = bias, cst[4 DAP + d] = row sums (as stage_head)
template <int H, int DAP, int DA>
__device__ __forceinline__ void stage_head_t(float *whlT, float *cst, const float *__restrict__ Whp,
                                             const float *__restrict__ bhp, int n_threads) {
  constexpr int DAE = head_t_rows(DA), BLK = head_t_block(H, DA);
  for (int e = threadIdx.x; e < 2 * DAE * (H / 2); e += n_threads) {
    const int R = e % (H / 2), d = (e / (H / 2)) % DAE, hh = e / ((H / 2) * DAE);
    whlT[hh * BLK + d * (H / 2) + R] = d < DA ? Whp[d * H + feat_base(R) + 4 * hh] : 0.f;
  }
  for (int e = threadIdx.x; e < DAP; e += n_threads) {
    cst[e] = e < DA ? bhp[e] : 0.f;
    float rs = 0.f;
    if (e < DA)
      for (int f = 0; f < H; ++f) rs += Whp[e * H + f];
    cst[4 * DAP + e] = rs;
  }
}

// PACKED: the LayerNorm / ReLU backward of the epilogue on register pairs (2 x v_pk_fma_f32 per pair + the two-instruction
// mask_pop per element: 3 VALU per element instead of 6) -- for the issue-bound fused kernel (update.hip); the stand-alone head
// kernels are HBM-bound and keep the plain form, which leaves the scheduler free.
template <int H, int DAP, int DA = DAP, bool PACKED = false, bool WT = false>
__device__ __forceinline__ void head_bwd_regs_bits(const f32x4 (&xs)[H / 8], uint32_t b0, const uint32_t b1, float rstd,
                                                   long slab, int lane, const float *whl /* base, both halves */,
                                                   const float (&dzh)[DAP], float s1, float s2,
                                                   float *__restrict__ dz_out, const uint32_t bm1 = 0u,
                                                   const uint32_t bm2 = 0u) {
  // (H = 256 has four mask words per lane: b0, bm1, bm2, b1)
  // b0 / b1: the lane's first / last ReLU-mask word (the same word when H = 64), MSB-first (common.h)
  s1 *= (1.0f / H);
  s2 *= (1.0f / H);
  // dx_hat^T[f][n] = sum_d W'[d][f] dz[n][d] on the fp32 MFMA: k = d (DAP/2 steps of 2), 4 output tiles, and the result
  // lands in the accumulator layout = the layout of xs.  Replaces 64 x DAP FMAs + 16 x DAP broadcast LDS reads per lane
  // (LDS-latency bound: the loss kernels spent half their wave time in s_waitcnt) by DAP/2 x H/32 MFMAs and as many
  // conflicted-but-few ds_read_b32 of the same weight image.
  const int i = lane & 31, h = lane >> 5;
  // W'[2s + h][32 t + i]: [R][DAP] image at + 16 t DAP + 2 s, transposed image (WT, whl = whlT) at + 16 t + 2 s H/2
  const float *wa = WT ? whl + ((i >> 2) & 1) * head_t_block(H, DA) + (i & 3) + 4 * (i >> 3) + h * (H / 2)
                       : whl + (((i >> 2) & 1) * (H / 2) + (i & 3) + 4 * (i >> 3)) * DAP + h;
  f32x16 acc[H / 32];
#pragma unroll
  for (int t = 0; t < H / 32; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int st = 0; st < (DA + 1) / 2; ++st) {  // (entries >= DA of dzh are zero)
    // (bit select: written as `h ? dzh[2 st + 1] : dzh[2 st]` the compiler turned the pair into a two-entry SCRATCH array
    // indexed by h when both entries were cheap to materialise -- the critic's {dv, 0} -- one scratch round trip per slab)
    const unsigned hm = 0u - (unsigned)h;
    const float bsel = __uint_as_float((__float_as_uint(dzh[2 * st + 1]) & hm) | (__float_as_uint(dzh[2 * st]) & ~hm));
#pragma unroll
    for (int t = 0; t < H / 32; ++t)
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[WT ? 16 * t + 2 * st * (H / 2) : 16 * t * DAP + 2 * st], bsel, acc[t], 0, 0, 0);
  }
  f32x4 *op = reinterpret_cast<f32x4 *>(dz_out + slab * (long)(H * SLAB)) + lane;
  const float c1 = -s1 * rstd, c2 = -s2 * rstd;
  const f32x2 c1v = {c1, c1}, c2v = {c2, c2}, rv = {rstd, rstd};
#pragma unroll
  for (int q = 0; q < H / 8; ++q) {
    if (q == 8) b0 = H == 256 ? bm1 : b1;  // next mask word; bits are consumed MSB-first
    if (H == 256 && q == 16) b0 = bm2;
    if (H == 256 && q == 24) b0 = b1;
    f32x4 o;
    if constexpr (PACKED) {
      // da = fma(x_hat, -s2 rstd, fma(dx_hat, rstd, -s1 rstd)) on pairs, as ln_bwd_relu_mbits (common.h)
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        const int e = 4 * q + 2 * P;
        const f32x2 d = {acc[e >> 4][e & 15], acc[(e + 1) >> 4][(e + 1) & 15]}, x = {xs[q][2 * P], xs[q][2 * P + 1]};
        const f32x2 da = fma2(x, c2v, fma2(d, rv, c1v));
        o[2 * P] = mask_pop(da[0], b0);
        o[2 * P + 1] = mask_pop(da[1], b0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float dx = acc[(4 * q + c) >> 4][(4 * q + c) & 15];
        const float da = rstd * (dx - s1 - xs[q][c] * s2);
        o[c] = (int)b0 < 0 ? da : 0.f;  // MSB-first mask (common.h)
        b0 <<= 1;
      }
    }
    op[q * WAVE] = o;
  }
}

template <int H, int DAP>
__device__ __forceinline__ void head_bwd_regs(const f32x4 (&xs)[H / 8], const uint32_t *__restrict__ mask_in,
                                              float rstd, long slab, int lane, const float *whl /* base, both halves */,
                                              const float (&dzh)[DAP], float s1, float s2,
                                              float *__restrict__ dz_out) {
  constexpr int NW = (H / 2 + 31) / 32;
  const uint32_t b0 = mask_in[(slab * NW + 0) * WAVE + lane];
  const uint32_t b1 = NW > 1 ? mask_in[(slab * NW + (NW - 1)) * WAVE + lane] : 0u;
  const uint32_t bm1 = NW == 4 ? mask_in[(slab * NW + 1) * WAVE + lane] : 0u, bm2 = NW == 4 ? mask_in[(slab * NW + 2) * WAVE + lane] : 0u;
  head_bwd_regs_bits<H, DAP>(xs, b0, b1, rstd, slab, lane, whl, dzh, s1, s2, dz_out, bm1, bm2);
}

// ---- head weight gradient fused into the loss kernels:  dW_head'[d][f] += sum_s dhead[s][d] x_hat_L[s][f].
// The loss kernel already holds x_hat_L (registers, lane = sample) and dhead; the reduction runs over samples, so both
// operands go through a wave-private LDS transpose ([sample][feature], the staging layout of k_dw) and 64 (H = 128)
// MFMAs per slab -- instead of a separate pass that re-reads x_hat_L (512 B/sample) and a [M][32] dhead matrix from HBM.
template <int H>
struct HeadDw {
  static constexpr int HX = 64 + 4;                       // one 64-feature half of x_hat per pass (row stride, floats)
  static constexpr int TD = 33;                           // dhead tile row stride
  static constexpr int WAVE_FLOATS = SLAB * HX + SLAB * TD;
  static constexpr int OUT_FLOATS = 32 * H + 32;          // per-workgroup partial: dWp[32][H] then dbp[32] (k_dw layout)
};

template <int H, int DAP>
__device__ __forceinline__ void head_dw_step(const f32x4 (&xs)[H / 8], const float (&dzh)[DAP], float *tx, float *td,
                                             int lane, f32x16 (&acc)[H / 32]) {
  constexpr int HX = HeadDw<H>::HX, TD = HeadDw<H>::TD;
  const int i = lane & 31, h = lane >> 5;
  if (h == 0) {
#pragma unroll
    for (int d = 0; d < DAP; ++d) td[i * TD + d] = dzh[d];  // columns >= DAP stay zero (cleared once)
  }
#pragma unroll
  for (int half = 0; half < H / 64; ++half) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      *reinterpret_cast<f32x4 *>(tx + i * HX + 32 * (q >> 2) + 8 * (q & 3) + 4 * h) = xs[8 * half + q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private hand-off between lanes (see mlp.hip)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < SLAB / 2; ++t) {
      const float a = td[(2 * t + h) * TD + i];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const float b = tx[(2 * t + h) * HX + 32 * n + i];
        acc[2 * half + n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2 * half + n], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // every lane is done reading tx / td before the next pass overwrites them
  }
}

// Variant with the accumulators in a wave-private LDS tile hw[HROWS][H] (HROWS = number of head outputs that can be non-zero,
// <= 8): the 64 persistent accumulator registers of head_dw_step become 32 transient ones per 64-feature pass, which is what
// lets the loss kernels keep the NEXT slab's x_hat_L in flight (they are latency-bound otherwise).  Row d of a 32x32 tile
// sits in register r = d & 3 of lane half h = d >> 2 (d < 8).
template <int H, int DAP, int HROWS>
__device__ __forceinline__ void head_dw_step_lds(const f32x4 (&xs)[H / 8], const float (&dzh)[DAP], float *tx, float *td,
                                                 int lane, float *hw) {
  constexpr int HX = HeadDw<H>::HX, TD = HeadDw<H>::TD;
  static_assert(HROWS <= 8, "rows 0..7 of a 16 x 16 tile live in the four registers of lane groups 0 and 1");
  const int i = lane & 31, h = lane >> 5;
  const int m16 = lane & 15, kg = lane >> 4;  // v_mfma_f32_16x16x4_f32: A lane (m, k), B lane (n, k), D lane n holds rows 4 kg + r
  if (h == 0) {
#pragma unroll
    for (int d = 0; d < DAP; ++d) td[i * TD + d] = dzh[d];  // columns >= DAP stay zero (cleared once)
  }
#pragma unroll
  for (int half = 0; half < H / 64; ++half) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      *reinterpret_cast<f32x4 *>(tx + i * HX + 32 * (q >> 2) + 8 * (q & 3) + 4 * h) = xs[8 * half + q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private hand-off between lanes (see mlp.hip)
    __builtin_amdgcn_wave_barrier();
    // dW_head'[d][f] += sum_s dzh[s][d] x_hat[s][f] on the 16 x 16 x 4 fp32 MFMA: at most 8 of the M rows are head outputs, so
    // the 32 x 32 x 2 shape (round 2) spent 64 cycles per instruction on a tile that is three quarters padding; still 32 MFMAs
    // per 64-feature pass, at half the pipe time each, and 16 instead of 32 transient accumulator registers
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < SLAB / 4; ++t) {
      const float a = td[(4 * t + kg) * TD + m16];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float b = tx[(4 * t + kg) * HX + 16 * n + m16];
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // plain read-modify-write: every (row, column) of the wave's tile belongs to exactly one lane, and a wave's LDS
        // accesses execute in order (ds_add_f32 measured slower here: LDS atomics run at a fraction of the ds_write rate)
        if (r < HROWS && kg == 0) hw[r * H + 64 * half + 16 * n + m16] += acc[n][r];
        if (4 + r < HROWS && kg == 1) hw[(4 + r) * H + 64 * half + 16 * n + m16] += acc[n][r];
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // every lane is done reading tx / td before the next pass overwrites them
  }
}

// end of kernel, LDS-accumulator variant: the waves' tiles hacc[NWAVES][HROWS][H] and bias sums in fixed order -> ONE partial
// row dWp[32][H] | dbp[32] (rows >= HROWS are zero).  `dbl` = NWAVES * PS_STRIDE floats of scratch LDS.
template <int H, int DAP, int HROWS, int NWAVES>
__device__ __forceinline__ void head_dw_finish_lds(const float *hacc, float (&dbacc)[DAP], float *dbl,
                                                   float *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  float dbs[DAP];
#pragma unroll
  for (int d = 0; d < DAP; ++d) dbs[d] = wave_reduce_sum(dbacc[d]);
  __syncthreads();  // every wave's LDS accumulation is complete; dbl is free
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < DAP; ++d) dbl[wave * PS_STRIDE + d] = dbs[d];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < HeadDw<H>::OUT_FLOATS; e += 64 * NWAVES) {
    float t = 0.f;
    if (e < 32 * H) {
      const int row = e / H;
      if (row < HROWS) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) t += hacc[(w * HROWS + row) * H + (e - row * H)];
      }
    } else {
      const int d = e - 32 * H;
      if (d < DAP) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) t += dbl[w * PS_STRIDE + d];
      }
    }
    out[e] = t;
  }
}

// end of kernel: combine the four waves' accumulators through LDS (one wave at a time, fixed order -> deterministic)
// and write this workgroup's partial in the layout harl_reduce_partials_multi expects.
template <int H, int DAP>
__device__ __forceinline__ void head_dw_finish(f32x16 (&acc)[H / 32], float (&dbacc)[DAP], float *buf /* >= 32*H+32 */,
                                               float *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  float dbs[DAP];
#pragma unroll
  for (int d = 0; d < DAP; ++d) dbs[d] = wave_reduce_sum(dbacc[d]);
  __syncthreads();  // staging tiles no longer in use: buf aliases them
  for (int w = 0; w < WAVES_PER_WG; ++w) {
    if (wave == w) {
#pragma unroll
      for (int n = 0; n < H / 32; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = (r & 3) + 8 * (r >> 2) + 4 * h;
          float *p = buf + o * H + 32 * n + i;
          *p = (w == 0 ? 0.f : *p) + acc[n][r];
        }
      if (lane < 32) {
        float v = 0.f;
#pragma unroll
        for (int d = 0; d < DAP; ++d)
          if (lane == d) v = dbs[d];
        float *p = buf + 32 * H + lane;
        *p = (w == 0 ? 0.f : *p) + v;
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < HeadDw<H>::OUT_FLOATS; e += WG_THREADS) out[e] = buf[e];
}

// block-level reduction of NV per-lane partial sums -> part_scalars[blockIdx.x][0..NV)
// Head gradients of a slab for the separate dW pass.  DAP <= 32: row-major [M_pad][32] (k_dw A_KIND 1; lane half h writes 16
// columns).  DAP = 64 (Categorical heads with 33..64 actions): an ATL(64) image -- the layout of every other dz tensor --
// so that the head's weight gradient is an ordinary harl_mlp_dw_partials(a_kind = 0, HO = 64) launch.
template <int DAP>
__device__ __forceinline__ void store_dhead(float *__restrict__ dhead, long slab, int lane, const float (&dzh)[DAP]) {
  const int i = lane & 31, h = lane >> 5;
  if constexpr (DAP <= 32) {
    float *dh = dhead + (slab * SLAB + i) * DHEAD_LD + 16 * h;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float v0 = 0.f, v1 = 0.f;
      if (c < DAP) v0 = dzh[c < DAP ? c : 0];
      if (16 + c < DAP) v1 = dzh[16 + c < DAP ? 16 + c : 0];
      dh[c] = h ? v1 : v0;
    }
  } else {
    static_assert(DAP == 64, "wide heads are padded to 64 outputs");
    float v[32];
#pragma unroll
    for (int R = 0; R < 32; ++R) {
      const int f0 = 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2);  // feature of register R in lane half 0 (+4 in half 1)
      v[R] = h ? dzh[f0 + 4] : dzh[f0];
    }
    atl_store<64>(dhead, slab, lane, v);
  }
}

template <int NV>
__device__ __forceinline__ void block_reduce_store(float (&v)[NV], float *red /*[4][PS_STRIDE]*/, float *out_row) {
  const int lane = threadIdx.x & 63, wave = wave_id();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float t = wave_reduce_sum(v[k]);
    if (lane == 0) red[wave * PS_STRIDE + k] = t;
  }
  __syncthreads();
  if (threadIdx.x < PS_STRIDE) {
    float t = 0.f;
    if (threadIdx.x < NV)
      t = (red[0 * PS_STRIDE + threadIdx.x] + red[1 * PS_STRIDE + threadIdx.x]) +
          (red[2 * PS_STRIDE + threadIdx.x] + red[3 * PS_STRIDE + threadIdx.x]);
    out_row[threadIdx.x] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// Per-sample actor-head arithmetic of lane (sample i = lane & 31, half h); both lanes of a sample hold the same head
// outputs z.  Log-prob pass (!TRAIN): writes log-probs / head outputs / factor product and returns false.  Training:
// ratio, clipped surrogate x factor, entropy -> per-lane partial sums `sc`, d(unscaled loss)/d(head output) `dzh`, and the
// two closed-form LayerNorm-backward means s1 = sum_d dzh[d] rowsum(W'_d), s2 = sum_d dzh[d] (z_d - b_d); returns true.
// (happo.py:66-91, act.py:104-157, distributions.py:7-89, on_policy_ha_runner.py:116-124)
// ---------------------------------------------------------------------------------------------
// The per-row loss inputs of one lane's sample (a few dwords gathered through the minibatch index).  Loaded by
// actor_row_load AHEAD of the head arithmetic that consumes them -- the kernels issue the next slab's rows while the current
// slab is being worked on -- so that their latency is not paid in the middle of every slab (measured: the loss kernel spent
// > 50 % of its wave cycles in s_waitcnt with these loads issued at their point of use).
template <int DAP>
struct ActorRow {
  float a[DAP];    // actions (Categorical: a[0])
  float olp[DAP];  // stored log-probs (Categorical: olp[0])
  float av[DAP];   // availability mask (Categorical with avail only)
  float act, adv, fct;
};
// DA > 0: the action width is a compile-time constant (update.hip instantiates per width; the run-time `d < D` tests of the
// generic kernels cost ~50 scalar branches and ~230 v_readlane / v_writelane of spilled lane masks per slab there)
template <int DAP, bool DISCRETE, bool TRAIN, int DA = 0>
__device__ __forceinline__ void actor_row_load(const ActorArgs &A, long slab, int lane, ActorRow<DAP> &R) {
  const int i = lane & 31;
  const int D = DA ? DA : A.act_dim;
  const long j = slab * SLAB + i;
  const long jc = j < A.M ? j : A.M - 1;
  const long row = A.idx ? A.idx[jc] : jc;
  const long orow = TRAIN ? row : jc;
  if (!DISCRETE) {
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      R.a[d] = (A.actions && d < D) ? A.actions[row * D + d] : 0.f;
      R.olp[d] = ((TRAIN || A.old_logp) && d < D) ? A.old_logp[orow * D + d] : 0.f;
    }
  } else {
    R.a[0] = A.actions ? A.actions[row] : 0.f;
    R.olp[0] = (TRAIN || A.old_logp) ? A.old_logp[orow] : 0.f;
#pragma unroll
    for (int d = 0; d < DAP; ++d) R.av[d] = (A.avail && d < D) ? A.avail[row * D + d] : 1.f;
  }
  if (TRAIN) {
    R.act = A.active ? A.active[row] : 1.f;
    R.adv = A.adv[row];
    R.fct = A.factor_in ? A.factor_in[row] : 1.f;
  } else {
    // factor product pass: the OLD factor of the row travels with the other row inputs (one slab ahead in the callers that
    // prefetch) instead of being loaded, waited for and written back inside the sample arithmetic; every row is touched once
    R.fct = A.factor_out ? A.factor_out[jc] : 1.f;
  }
}

// ---- the same loads for the persistent fused kernel (update.hip), WITHOUT a branch and without a wait.  (1) The gathered row
// index travels one slab further ahead than the rows (row_index_request in one iteration, row_index_resolve in the next):
// loaded at its point of use, `idx ? idx[j] : j` put an s_waitcnt vmcnt(0) behind the join of the two paths -- executed with
// idx == NULL as well -- which drained every outstanding memory operation once per slab, the 16 dz_2 stores the wave had just
// issued included (ISA, round 4).  (2) Optional arrays (actions, stored log-probs, availability, active masks, factors) are
// read unconditionally, a NULL pointer replaced by the head weights (always mapped, >= 8 floats); the consumers select
// (actor_sample tests the same pointers).  A load under a wave-uniform branch makes the number of operations in flight depend
// on the path, and the compiler's vmcnt then assumes the shortest one: the top of the loop waited for the row loads issued a
// few thousand cycles earlier instead of only for the inputs requested a whole iteration before.
__device__ __forceinline__ long row_index_request(const int64_t *idx, const float *mapped, long slab, int lane, long M) {
  const long j = slab * SLAB + (lane & 31);
  const long jc = j < M ? j : M - 1;
  const int64_t *p = idx ? idx + jc : reinterpret_cast<const int64_t *>(mapped);
  return *p;
}
__device__ __forceinline__ long row_index_resolve(const int64_t *idx, long raw, long slab, int lane, long M) {
  const long j = slab * SLAB + (lane & 31);
  return idx ? raw : (j < M ? j : M - 1);
}
template <int DAP, bool DISCRETE, bool TRAIN, int DA>
__device__ __forceinline__ void actor_row_load_at(const ActorArgs &A, long slab, int lane, long row, ActorRow<DAP> &R) {
  static_assert(DA > 0 && DA <= DAP, "compile-time action width");
  const long j = slab * SLAB + (lane & 31);
  const long jc = j < A.M ? j : A.M - 1;
  const long orow = TRAIN ? row : jc;
  const float *mapped = A.Whp;
  const bool has_olp = TRAIN || A.old_logp;
  if (!DISCRETE) {
    const float *ap = A.actions ? A.actions + row * DA : mapped;
    const float *op = has_olp ? A.old_logp + orow * DA : mapped;
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      R.a[d] = d < DA ? ap[d < DA ? d : 0] : 0.f;
      R.olp[d] = d < DA ? op[d < DA ? d : 0] : 0.f;
    }
  } else {
    R.a[0] = *(A.actions ? A.actions + row : mapped);
    R.olp[0] = *(has_olp ? A.old_logp + orow : mapped);
    const float *vp = A.avail ? A.avail + row * DA : mapped;
#pragma unroll
    for (int d = 0; d < DAP; ++d) R.av[d] = d < DA ? vp[d < DA ? d : 0] : 1.f;
  }
  if (TRAIN) {
    R.act = *(A.active ? A.active + row : mapped);      // (consumer: A.active ? R.act : 1)
    R.adv = A.adv[row];
    R.fct = *(A.factor_in ? A.factor_in + row : mapped);  // (consumer: A.factor_in ? R.fct : 1)
  } else {
    R.fct = *(A.factor_out ? A.factor_out + jc : mapped);  // (consumer: only under A.factor_out)
  }
}

template <int DAP, bool DISCRETE, bool TRAIN, int DA = 0>
__device__ __forceinline__ bool actor_sample(const ActorArgs &A, const float *cst, float (&z)[DAP], long slab, int lane,
                                             float adv_mean, float adv_den, float (&sc)[8 + DAP], float (&dzh)[DAP],
                                             float &s1_out, float &s2_out, const ActorRow<DAP> &R) {
  const int i = lane & 31, h = lane >> 5;
  const int D = DA ? DA : A.act_dim;
  const int act_w = DISCRETE ? 1 : D;
  const float inv_D = uniform_f(1.0f / (float)D);
  float zlin[DAP];  // x_hat . Whp[d]  (= z - bias), needed by the closed-form LayerNorm backward
#pragma unroll
  for (int d = 0; d < DAP; ++d) zlin[d] = z[d] - cst[d];

  const long j = slab * SLAB + i;
  const bool valid = j < A.M && (A.m_pad == 0 || (j % A.m_pad) < A.m_valid);
  const long jc = j < A.M ? j : A.M - 1;
  const long row = A.idx ? A.idx[jc] : jc;
  const bool count_me = valid && h == 0;

#pragma unroll
  for (int d = 0; d < DAP; ++d) dzh[d] = 0.f;

  float imp = 1.f;      // aggregated importance weight
  float ratio_d[DAP];   // per-dim ratios (Gaussian) / [0] only (Categorical)
  float ent = 0.f;
  float logp_d[DAP];

  if (!DISCRETE) {
    float prod = 1.f, sum = 0.f;
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      ratio_d[d] = 1.f;
      logp_d[d] = 0.f;
      if (d < D) {
        const float sig = cst[DAP + d], lsig = cst[2 * DAP + d];
        const float a = A.actions ? R.a[d] : z[d];  // actions == NULL: head outputs only
        const float diff = a - z[d];
        const float var = sig * sig;
        const float lp = -(diff * diff) * (0.5f * cst[6 * DAP + d]) - lsig - LOG_SQRT_2PI;  // torch Normal.log_prob (1/var from the prologue)
        logp_d[d] = lp;
        ent += HALF_LOG_2PI_PLUS_HALF + lsig;
        if (TRAIN || A.old_logp) {
          const float r = expf(lp - R.olp[d]);
          ratio_d[d] = r;
          prod *= r;
          sum += r;
        }
      }
    }
    imp = A.agg_mean ? sum * inv_D : prod;
  } else {
    // Categorical: logits masked to -1e10 where unavailable, normalised by logsumexp (distributions.py:52-55)
    float mx = -3.0e38f;
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      if (d < D) {
        if (A.avail && R.av[d] == 0.f) z[d] = -1e10f;
        mx = fmaxf(mx, z[d]);
      }
    }
    float se = 0.f;
#pragma unroll
    for (int d = 0; d < DAP; ++d)
      if (d < D) se += expf(z[d] - mx);
    const float lse = mx + logf(se);
    const int a = A.actions ? (int)R.a[0] : 0;
    float lpa = 0.f;
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      logp_d[d] = 0.f;
      ratio_d[d] = 0.f;  // reused as p_d below
      if (d < D) {
        const float lp = z[d] - lse;
        logp_d[d] = lp;
        const float p = expf(lp);
        ratio_d[d] = p;
        ent -= fmaxf(lp, -3.4028234663852886e38f) * p;
        if (d == a) lpa = lp;
      }
    }
    if (TRAIN || A.old_logp) imp = expf(lpa - R.olp[0]);
    // stash log p(a) in z[0] for the logp output below
    z[0] = lpa;
  }

  if (!TRAIN) {
    if (valid && h == 0 && A.head_out) {
#pragma unroll
      for (int d = 0; d < DAP; ++d)
        if (d < D) A.head_out[j * D + d] = DISCRETE ? logp_d[d] : z[d];
    }
    if (valid && h == 0) {
      if (A.logp_out) {
        if (DISCRETE) A.logp_out[j] = z[0];
        else {
#pragma unroll
          for (int d = 0; d < DAP; ++d)
            if (d < D) A.logp_out[j * D + d] = logp_d[d];
        }
      }
      if (A.factor_out) A.factor_out[j] = R.fct * imp;  // on_policy_ha_runner.py:116-124 (R.fct = the row's old factor, j < M here)
    }
    return false;
  }

  // ---------------- loss + backward (happo.py:66-91) ----------------
  if (A.logp_out && valid && h == 0) {  // log pi(a|o) under the CURRENT parameters, by batch position
    if (DISCRETE) A.logp_out[j] = z[0];
    else {
#pragma unroll
      for (int d = 0; d < DAP; ++d)
        if (d < D) A.logp_out[j * D + d] = logp_d[d];
    }
  }
  const float act = A.active ? R.act : 1.f;  // (actor_row_load_at reads a mapped dummy word for absent arrays)
  const float advn = (R.adv - adv_mean) * adv_den;  // adv_den: RECIPROCAL of (std + 1e-5), formed once per kernel
  const float fct = A.factor_in ? R.fct : 1.f;  // 1 without a sequential-update factor (MAPPO)
  const float lo = A.clip_lo, hi = A.clip_hi;
  const float surr1 = imp * advn;
  const float impc = fminf(fmaxf(imp, lo), hi);
  const float surr2 = impc * advn;
  const float mn = fminf(surr1, surr2);
  const float inrange = (imp >= lo && imp <= hi) ? 1.f : 0.f;
  // torch.min(a, b) backward: ties split the gradient evenly between the two inputs
  float gsel = surr1 < surr2 ? 1.f : (surr1 > surr2 ? inrange : 0.5f + 0.5f * inrange);
  if (A.trpo != 0) gsel = 1.f;
  // HAPPO/HAA2C: d(sum_s -f*min|surr*active)/d(imp) ; HATRPO: d(sum_s +imp*f*adv*active)/d(imp)
  const float dimp = valid ? (A.trpo == 1 ? fct * act * advn : -fct * act * advn * gsel) : 0.f;
  const float ecoef = (valid && A.trpo != 1) ? -A.entropy_coef * act : 0.f;  // weight of d(ent_s)

  if (count_me) {
    sc[0] += A.trpo == 1 ? surr1 * fct * act : -fct * (A.trpo == 2 ? surr1 : mn) * act;
    sc[1] += act;
    sc[2] += ent * act;
    sc[3] += imp;
    sc[4] += 1.f;
  }

  if (!DISCRETE) {
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      if (d < D) {
        const float sig = cst[DAP + d];
        const float var = sig * sig;
        const float a = R.a[d];
        const float diff = a - z[d];
        // d imp / d logp_d : prod -> prod/r_d * r_d ; mean -> r_d / D
        const float dlp = dimp * (A.agg_mean ? ratio_d[d] * inv_D : imp);
        const float isig = cst[5 * DAP + d], ivar = cst[6 * DAP + d];  // 1/sigma, 1/sigma^2 (prologue)
        dzh[d] = dlp * diff * ivar;
        const float dsig = dlp * (diff * diff * ivar * isig - isig) + ecoef * isig;
        if (h == 0) sc[8 + d] += dsig * cst[3 * DAP + d];
      }
    }
  } else {
    const int a = (int)R.a[0];
    const float dlp = dimp * imp;
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      if (d < D) {
        const float p = ratio_d[d];
        const float onehot = d == a ? 1.f : 0.f;
        // d logp_a/dz_d = onehot - p_d ;  d ent/dz_d = -p_d (log p_d + ent)
        dzh[d] = dlp * (onehot - p) + ecoef * (-p * (logp_d[d] + ent));
        if (p == 0.f) dzh[d] = dlp * onehot;  // masked logits receive no gradient
      }
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int d = 0; d < DAP; ++d) {
    s1 += dzh[d] * cst[4 * DAP + d];
    s2 += dzh[d] * zlin[d];
  }
  s1_out = s1;
  s2_out = s2;
  return true;
}

struct CriticArgs {
  const float *xL;
  const uint32_t *relu_mask;
  const float *rstd;
  long M;
  const float *Whp, *bhp;
  const int64_t *idx;
  const float *value_preds, *returns, *vn_stats;
  float clip_param, huber_delta;
  long m_valid, m_pad;
  int use_clipped, use_huber;
  float *dzL, *dhead, *part_scalars, *values_out;
  float *dw_part;  // fused head weight gradient partials (see ActorArgs)
  long n_slabs;
};

// Per-sample critic-head arithmetic (v_critic.py:75-114): clipped value loss with optional ValueNorm'd targets and
// Huber loss; returns d(unscaled loss)/d(value) (0 for padding rows) and adds {loss, count} to `sc`.  !TRAIN: writes the
// value and returns false.
__device__ __forceinline__ void critic_row_load(const CriticArgs &A, long slab, int lane, float &vold, float &ret) {
  const long j = slab * SLAB + (lane & 31);
  const long jc = j < A.M ? j : A.M - 1;
  const long row = A.idx ? A.idx[jc] : jc;
  vold = A.value_preds[row];
  ret = A.returns[row];
}

__device__ __forceinline__ void critic_row_load_at(const CriticArgs &A, long row, float &vold, float &ret) {
  vold = A.value_preds[row];
  ret = A.returns[row];
}

template <bool TRAIN>
__device__ __forceinline__ bool critic_sample(const CriticArgs &A, float v, long slab, int lane, float vmean, float vsd,
                                              float (&sc)[8], float &dv_out, float vold, float ret) {
  const int i = lane & 31, h = lane >> 5;
  const long j = slab * SLAB + i;
  const bool valid = j < A.M && (A.m_pad == 0 || (j % A.m_pad) < A.m_valid);
  if (!TRAIN) {
    if (j < A.M && h == 0) A.values_out[j] = v;
    return false;
  }
  const float eps = A.clip_param, dl = A.huber_delta;
  const float diff = v - vold;
  const float vclip = vold + fminf(fmaxf(diff, -eps), eps);
  const float tgt = A.vn_stats ? (ret - vmean) / vsd : ret;
  const float ec = tgt - vclip, eo = tgt - v;
  float lc, lo_, gc, go;  // losses and d(loss)/d(e)
  if (A.use_huber) {      // models_tools.py:64-68
    lc = fabsf(ec) <= dl ? ec * ec / 2.f : dl * (fabsf(ec) - dl / 2.f);
    lo_ = fabsf(eo) <= dl ? eo * eo / 2.f : dl * (fabsf(eo) - dl / 2.f);
    gc = fabsf(ec) <= dl ? ec : (ec > 0.f ? dl : -dl);
    go = fabsf(eo) <= dl ? eo : (eo > 0.f ? dl : -dl);
  } else {
    lc = ec * ec / 2.f;
    lo_ = eo * eo / 2.f;
    gc = ec;
    go = eo;
  }
  const float inr = (diff >= -eps && diff <= eps) ? 1.f : 0.f;
  float loss = lo_, dv = -go;
  if (A.use_clipped) {  // torch.max: ties split the gradient evenly
    if (lc > lo_) {
      loss = lc;
      dv = -gc * inr;
    } else if (lc == lo_) {
      dv = 0.5f * (-go) + 0.5f * (-gc * inr);
    }
  }
  if (!valid) dv = 0.f;
  if (valid && h == 0) {
    sc[0] += loss;
    sc[1] += 1.f;
  }
  dv_out = dv;
  return true;
}
}  // namespace
