// gru.hip -- GRU recurrent layer (harl/models/base/rnn.py:8-81, recurrent_n = 1) forward and BPTT on the MFMA pipes, gfx950.
//
// A recurrent batch is L time steps x m sequences, row (l, j) at index l*m_pad + j (m_pad = m rounded up to 32, so a
// wave-slab never straddles two time steps).  A wave owns 32 sequences for the whole chunk: the hidden state lives in
// its registers in the accumulator layout (common.h), which is also a valid MFMA B operand, so h_{l-1} feeds the
// W_hh GEMM of step l straight from the register file -- no LDS staging, no transposes along the recurrence.
//   per step:  h~ = h_{l-1} * mask_l
//              r = sigma(W_ir x + b_ir + W_hr h~ + b_hr)     z = sigma(W_iz x + b_iz + W_hz h~ + b_hz)
//              n = tanh(W_in x + b_in + r * (W_hn h~ + b_hn))   h_l = (1 - z) * n + z * h~        (torch.nn.GRU, gates r,z,n)
//              y_l = (h_l - mean) * rstd        (rnn.norm; its affine part is folded into the head weights)
// Both weight matrices stay in LDS (W_ih fp32 [3H][H+1], W_hh as three bf16 images: 122 KiB for H = 64 -> one workgroup per CU).
// H = 64 only (every recurrent tuned HARL config: SMAC / SMACv2 / football use hidden 64).
#include "common.h"
#include <stdlib.h>
#include "split_mfma.h"
#include "../../include/harl_hip.h"

using namespace harl;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {
constexpr int GH = 64;         // hidden width
constexpr int GR = GH / 2;     // registers per lane per width-64 activation
constexpr int GT = GH / 32;    // 32-row tiles per gate
constexpr long GRU_TP_MAX_GROUPS = 512;  // <= this many 32-sequence groups (one per SIMD at most): latency variant of the forward

// Gate nonlinearities on the hardware exp2 / rcp (1 ulp each): 4-5 VALU ops per value.  With libm's expf / tanhf and IEEE
// division the 96 transcendental values of a step cost ~2500 VALU instructions -- more than the step's 144 MFMAs, and on
// the recurrence's critical path (one wave, L dependent steps).  Per-tensor gradient error of a 10-25 step BPTT update
// against the oracle stays <= 1e-5 of the tensor's inf-norm (tools/rnn_diag.py), as with libm.
__device__ __forceinline__ float rcp_nr(float d) {  // v_rcp_f32 + one Newton step: <= 0.5 ulp
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float exp_scaled(float x, float c_hi, float c_lo) {  // 2^{x (c_hi + c_lo)}: constant in two pieces
  return __builtin_amdgcn_exp2f(fminf(fmaf(x, c_hi, x * c_lo), 126.0f));  // clamped: 1 + 2^126 is finite, so rcp_nr never sees inf * 0
}
__device__ __forceinline__ float sigmoidf_(float x) {
  return rcp_nr(1.0f + exp_scaled(x, -1.44269502162933349609f, -1.92596303029868504e-8f));
}
__device__ __forceinline__ float tanhf_(float x) {  // 1 - 2 / (1 + e^{2x});  saturates cleanly for |x| large
  return fmaf(-2.0f, rcp_nr(1.0f + exp_scaled(x, 2.88539004325866699219f, 3.85192606059737008e-8f)), 1.0f);
}

__device__ __forceinline__ void load_act(const float *__restrict__ base, long slab, int lane, float (&x)[GR]) {
  atl_load<GH>(base, slab, lane, x);
}
__device__ __forceinline__ void store_act(float *__restrict__ base, long slab, int lane, const float (&x)[GR]) {
  atl_store<GH>(base, slab, lane, x);
}
// row-major [rows][64] <-> accumulator layout for the 32 sequences of a wave (initial / final hidden state)
__device__ __forceinline__ void load_rows(const float *__restrict__ base, long row, int h, float (&x)[GR]) {
#pragma unroll
  for (int q = 0; q < GR / 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(base + row * GH + 32 * (q >> 2) + 8 * (q & 3) + 4 * h);
    x[4 * q + 0] = v[0];
    x[4 * q + 1] = v[1];
    x[4 * q + 2] = v[2];
    x[4 * q + 3] = v[3];
  }
}
__device__ __forceinline__ void store_rows(float *__restrict__ base, long row, int h, const float (&x)[GR]) {
#pragma unroll
  for (int q = 0; q < GR / 4; ++q) {
    f32x4 v{x[4 * q + 0], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
    *reinterpret_cast<f32x4 *>(base + row * GH + 32 * (q >> 2) + 8 * (q & 3) + 4 * h) = v;
  }
}

// acc[g][t] += W[(g*64 + 32t + i)][f(R,h)] * b[R]   over R = 0..31, for the gates listed in GATES (bit mask)
template <int GATES>
__device__ __forceinline__ void gemm_gates(f32x16 (&acc)[3][GT], const float *wl_lane, const float (&b)[GR]) {
  constexpr int LDW = GH + 1;
#pragma unroll
  for (int R = 0; R < GR; ++R) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      if (!((GATES >> g) & 1)) continue;
#pragma unroll
      for (int t = 0; t < GT; ++t) {
        const float a = wl_lane[(g * GH + 32 * t) * LDW + feat_base(R)];
        acc[g][t] = MFMA(a, b[R], acc[g][t]);
      }
    }
    if ((R & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads from being hoisted en bloc
  }
}

// =============================================================================================
// input half of the gates for ALL time steps at once: gi_g = W_ig' x_hat + b_ig' (+ b_hg for r, z).  No recurrence here,
// so every (step, sequence-group) slab is an independent wave-task; the sequential kernel below is left with the W_hh
// products only (192 instead of 384 MFMAs per step), which matters when few sequences are unrolled over many steps
// (the runner's full-length log-prob passes: 512 sequences x 160 steps in the SMAC-sized configuration).
// =============================================================================================
__global__ __launch_bounds__(WG_THREADS, 1) void k_gru_gates_x(const float *__restrict__ xin,
                                                               const float *__restrict__ Wih,
                                                               const float *__restrict__ bih,
                                                               const float *__restrict__ bhh, long n_slabs,
                                                               float *__restrict__ gi_r, float *__restrict__ gi_z,
                                                               float *__restrict__ gi_n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LDW = GH + 1;
  float *Wil = lds;                 // [192][65]
  float *bl = Wil + 3 * GH * LDW;   // [192]  b_i (+ b_h for the r and z gates)
  stage_matrix<3 * GH, GH, LDW, WG_THREADS>(Wil, Wih);
  for (int e = threadIdx.x; e < 3 * GH; e += WG_THREADS) bl[e] = bih[e] + (e < 2 * GH ? bhh[e] : 0.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const float *wi_lane = Wil + i * LDW + 4 * h;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float x[GR];
    load_act(xin, slab, lane, x);
    f32x16 acc[3][GT];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][t][r] = bl[g * GH + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    gemm_gates<7>(acc, wi_lane, x);
    float o[GR];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int R = 0; R < GR; ++R) o[R] = acc[g][R >> 4][R & 15];
      store_act(g == 0 ? gi_r : (g == 1 ? gi_z : gi_n), slab, lane, o);
    }
  }
}

// The same product on the bf16 pipe with the exact three-way operand split (round 5): 6 row tiles x 4 k-steps x 6 products = 144
// MFMAs of 32 cycles per slab instead of 192 fp32 MFMAs of 64 (each behind its own ds_read_b32), the split images of W_ih'
// (72 KiB) leave room for two workgroups per CU.  62 launches per 8-agent recurrent update over 2 560 slabs each: the kernel was
// at twice its own matrix-pipe bound and is now bound by its 84 MB of row traffic.  HARL_GRU_GATES_F32=1 keeps the fp32 kernel.
__global__ __launch_bounds__(WG_THREADS, 2) void k_gru_gates_xs(const float *__restrict__ xin,
                                                                const float *__restrict__ Wih,
                                                                const float *__restrict__ bih,
                                                                const float *__restrict__ bhh, long n_slabs,
                                                                float *__restrict__ gi_r, float *__restrict__ gi_z,
                                                                float *__restrict__ gi_n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MTH = 3 * GT, NJH = GH / 16;
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);                      // [3 terms][6 tiles][4 k-steps][64 lanes] x 16 B
  float *bl = reinterpret_cast<float *>(img + 3 * MTH * NJH * 64);  // [192]  b_i (+ b_h for the r and z gates)
  stage_split_matrix<3 * GH, GH, false, WG_THREADS>(img, Wih);
  for (int e = threadIdx.x; e < 3 * GH; e += WG_THREADS) bl[e] = bih[e] + (e < 2 * GH ? bhh[e] : 0.f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int h = lane >> 5;
  const u32x4 *wl = img + lane;
  const long s0 = (long)blockIdx.x * WAVES_PER_WG + wave, stride = (long)gridDim.x * WAVES_PER_WG;
  float x[GR];
  if (s0 < n_slabs) load_act(xin, s0, lane, x);
  for (long slab = s0; slab < n_slabs; slab += stride) {
    u32x4 x1[NJH], x2[NJH], x3[NJH];
    split_acts<GR>(x, x1, x2, x3);
    if (slab + stride < n_slabs) load_act(xin, slab + stride, lane, x);  // one slab ahead
    f32x16 a6[MTH];
#pragma unroll
    for (int t6 = 0; t6 < MTH; ++t6)
#pragma unroll
      for (int r = 0; r < 16; ++r) a6[t6][r] = bl[32 * t6 + (r & 3) + 8 * (r >> 2) + 4 * h];
    split_gemm<MTH, NJH>(wl, x1, x2, x3, a6, [](int) {});
    float o[GR];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int R = 0; R < GR; ++R) o[R] = a6[g * GT + (R >> 4)][R & 15];
      store_act(g == 0 ? gi_r : (g == 1 ? gi_z : gi_n), slab, lane, o);
    }
  }
}

// =============================================================================================
// forward
// =============================================================================================
template <bool PRE>
__global__ __launch_bounds__(WG_THREADS, 1) void k_gru_fwd(
    const float *__restrict__ xin, const float *__restrict__ gi_r, const float *__restrict__ gi_z,
    const float *__restrict__ gi_n, const float *__restrict__ mrow, const float *__restrict__ h0,
    const float *__restrict__ Wih, const float *__restrict__ bih, const float *__restrict__ Whh,
    const float *__restrict__ bhh, int L, long m_pad, float *__restrict__ y, float *__restrict__ rstd_y,
    float *__restrict__ hpm_s, float *__restrict__ r_s, float *__restrict__ z_s, float *__restrict__ n_s,
    float *__restrict__ hn_s, float *__restrict__ h_last, int save) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LDW = GH + 1;
  // W_hh: the recurrence's critical path (one wave, L dependent steps) runs on the bf16 pipe with the exact three-way
  // operand split (split_mfma.h): 6 tiles x 4 k-steps x 6 products = 144 MFMAs of 32 cycles per step instead of 192 of 64
  constexpr int MTH = 3 * GT, NJH = GH / 16;
  u32x4 *Whimg = reinterpret_cast<u32x4 *>(lds);  // [3 terms][6 tiles][4 k-steps][64 lanes] x 16 B = 72 KiB
  float *Wil = reinterpret_cast<float *>(Whimg + 3 * MTH * NJH * 64);  // [192][65]  (absent when the x half is precomputed)
  float *bil = Wil + (PRE ? 0 : 3 * GH * LDW);    // [192]
  float *bhl = bil + 3 * GH;                      // [192]
  if (!PRE) stage_matrix<3 * GH, GH, LDW, WG_THREADS>(Wil, Wih);
  stage_split_matrix<3 * GH, GH, false, WG_THREADS>(Whimg, Whh);
  for (int e = threadIdx.x; e < 3 * GH; e += WG_THREADS) {
    bil[e] = bih[e];
    bhl[e] = bhh[e];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const long groups = m_pad / SLAB;
  const float *wi_lane = Wil + i * LDW + 4 * h;
  const u32x4 *wh_img = Whimg + lane;
  for (long G = (long)blockIdx.x * WAVES_PER_WG + wave; G < groups; G += (long)gridDim.x * WAVES_PER_WG) {
    float hs[GR];
    load_rows(h0, G * SLAB + i, h, hs);
    for (int l = 0; l < L; ++l) {
      const long slab = (long)l * groups + G;
      const float mk = mrow[slab * SLAB + i];
#pragma unroll
      for (int R = 0; R < GR; ++R) hs[R] *= mk;  // h~ = h_{l-1} * mask_l
      if (save) store_act(hpm_s, slab, lane, hs);
      // accumulators: [0] r, [1] z share the x- and h- GEMMs; n keeps W_in x (acc[2]) and W_hn h~ (acch) apart
      f32x16 acc[3][GT], acch[3][GT];
      if (PRE) {  // x half (with b_i, and b_h for r/z) comes from k_gru_gates_x
        float gx[GR];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          load_act(g == 0 ? gi_r : (g == 1 ? gi_z : gi_n), slab, lane, gx);
#pragma unroll
          for (int R = 0; R < GR; ++R) acc[g][R >> 4][R & 15] = gx[R];
        }
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acch[2][t][r] = bhl[2 * GH + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
      } else {
        float x[GR];
        load_act(xin, slab, lane, x);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int o = g * GH + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
              acc[g][t][r] = g < 2 ? bil[o] + bhl[o] : bil[o];
              acch[g][t][r] = bhl[o];
            }
        gemm_gates<7>(acc, wi_lane, x);
      }
      {  // W_hh h~:  r, z onto the x-half accumulators, n (W_hn h~ + b_hn) on its own
        u32x4 h1[NJH], h2[NJH], h3[NJH];
        split_acts<GR>(hs, h1, h2, h3);
        f32x16 a6[MTH];
#pragma unroll
        for (int t = 0; t < GT; ++t) {
          a6[0 * GT + t] = acc[0][t];
          a6[1 * GT + t] = acc[1][t];
          a6[2 * GT + t] = acch[2][t];
        }
        split_gemm<MTH, NJH>(wh_img, h1, h2, h3, a6, [](int) {});
#pragma unroll
        for (int t = 0; t < GT; ++t) {
          acc[0][t] = a6[0 * GT + t];
          acc[1][t] = a6[1 * GT + t];
          acch[2][t] = a6[2 * GT + t];
        }
      }
      float rg[GR], zg[GR], ng[GR], hn[GR];
      float sum = 0.f;
#pragma unroll
      for (int R = 0; R < GR; ++R) {
        rg[R] = sigmoidf_(acc[0][R >> 4][R & 15]);
        zg[R] = sigmoidf_(acc[1][R >> 4][R & 15]);
        hn[R] = acch[2][R >> 4][R & 15];
        ng[R] = tanhf_(acc[2][R >> 4][R & 15] + rg[R] * hn[R]);
        hs[R] = (1.f - zg[R]) * ng[R] + zg[R] * hs[R];
        sum += hs[R];
      }
      if (save) {
        store_act(r_s, slab, lane, rg);
        store_act(z_s, slab, lane, zg);
        store_act(n_s, slab, lane, ng);
        store_act(hn_s, slab, lane, hn);
      }
      // rnn.norm (pure normalisation; affine folded into the head)
      sum = wave_sum32(sum);
      const float mean = sum * (1.0f / GH);
      float vs = 0.f;
#pragma unroll
      for (int R = 0; R < GR; ++R) {
        const float d = hs[R] - mean;
        vs += d * d;
      }
      vs = wave_sum32(vs);
      const float rstd = 1.0f / sqrtf(vs * (1.0f / GH) + 1e-5f);
      float yo[GR];
#pragma unroll
      for (int R = 0; R < GR; ++R) yo[R] = (hs[R] - mean) * rstd;
      store_act(y, slab, lane, yo);
      if (lane < 32) rstd_y[slab * SLAB + lane] = rstd;
    }
    if (h_last) store_rows(h_last, G * SLAB + i, h, hs);
  }
}

// =============================================================================================
// forward, LATENCY variant for few sequences unrolled over many steps (the runner's full-length log-prob passes: N
// sequences x T steps, one dependent chain per 32 sequences; at N = 512 the plain kernel keeps 16 waves busy for T x 6 us).
// Two waves share one slab and split the FEATURES: wave w owns hidden features [32 w, 32 w + 32) of all three gates, i.e.
// the accumulator-layout registers R = 16 w .. 16 w + 15 of every lane, the W_hh row tiles {g * 2 + w} (72 of the 144
// MFMAs of a step) and half of the gate nonlinearities.  Per step ONE exchange through LDS: the owner's half of
// h~_{l+1} = h_l * mask_{l+1}, already split into its three bf16 terms (= the partner's missing B-operand k-steps), and the
// half-row LayerNorm partials (mean_w, M2_w), merged with the exact pairwise formula
//     mean = (mean_0 + mean_1) / 2,   M2 = M2_0 + M2_1 + 16 (mean_0 - mean_1)^2      (32 + 32 features).
// The input half of the gates comes from k_gru_gates_x and is fetched one step ahead.  Round 5: it also SAVES the internals
// the backward needs (h~, r, z, n, hn: every wave stores its own half of each ATL(64) image, four float4 pieces), so the
// training forward of a chunked minibatch takes it too: 8192 sequences x 10 steps are 256 chains -- 64 workgroups of the
// one-wave-per-slab kernel on a 256-CU chip, 76 us per launch, 45 launches per 8-agent update.
// =============================================================================================
__global__ __launch_bounds__(WG_THREADS, 1) void k_gru_fwd_tp(
    const float *__restrict__ gi_r, const float *__restrict__ gi_z, const float *__restrict__ gi_n,
    const float *__restrict__ mrow, const float *__restrict__ h0, const float *__restrict__ Whh,
    const float *__restrict__ bhh, int L, long m_pad, float *__restrict__ y, float *__restrict__ rstd_y,
    float *__restrict__ h_last, float *__restrict__ hpm_s, float *__restrict__ r_s, float *__restrict__ z_s,
    float *__restrict__ n_s, float *__restrict__ hn_s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MTH = 3 * GT, NJH = GH / 16, TS = MTH * NJH * 64, HR = GR / 2;  // HR = 16 registers per lane and wave
  u32x4 *Whimg = reinterpret_cast<u32x4 *>(lds);                     // [3 terms][6 tiles][4 k-steps][64 lanes] x 16 B
  u32x4 *xch = Whimg + 3 * TS;                                        // [2 buffers][2 pairs][2 waves][6 slots][64 lanes]
  float *xst = reinterpret_cast<float *>(xch + 2 * 2 * 2 * 6 * 64);   // [2 buffers][2 pairs][2 waves][64 lanes][2]
  float *bhl = xst + 2 * 2 * 2 * 64 * 2;                              // b_hn [64]
  stage_split_matrix<3 * GH, GH, false, WG_THREADS>(Whimg, Whh);
  for (int e = threadIdx.x; e < GH; e += WG_THREADS) bhl[e] = bhh[2 * GH + e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), pair = wave >> 1, w = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  const long groups = m_pad / SLAB;
  const u32x4 *wl = Whimg + lane;
  for (long G0 = (long)blockIdx.x * 2; G0 < groups; G0 += (long)gridDim.x * 2) {  // uniform trip count (barriers inside)
    const bool live = G0 + pair < groups;
    const long G = live ? G0 + pair : groups - 1;
    float hs[HR], hm[HR];
    {  // own half of h_0: pieces q = 4 w .. 4 w + 3 of the row-major [m][64] state
#pragma unroll
      for (int q = 0; q < HR / 4; ++q) {
        const int qq = 4 * w + q;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(h0 + (G * SLAB + i) * GH + 32 * (qq >> 2) + 8 * (qq & 3) + 4 * h);
        hs[4 * q + 0] = v[0];
        hs[4 * q + 1] = v[1];
        hs[4 * q + 2] = v[2];
        hs[4 * q + 3] = v[3];
      }
    }
    auto own_pieces = [&](const float *base, long slab, f32x4 (&dst)[HR / 4]) {
      const f32x4 *pp = reinterpret_cast<const f32x4 *>(base + slab * (long)(GH * SLAB)) + lane;
#pragma unroll
      for (int q = 0; q < HR / 4; ++q) dst[q] = pp[(4 * w + q) * WAVE];
    };
    auto store_own = [&](float *base, long slab, const float (&v)[HR]) {  // this wave's half of an ATL(64) image
      f32x4 *pp = reinterpret_cast<f32x4 *>(base + slab * (long)(GH * SLAB)) + lane;
#pragma unroll
      for (int q = 0; q < HR / 4; ++q) pp[(4 * w + q) * WAVE] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    };
    f32x4 gr[HR / 4], gz[HR / 4], gn[HR / 4];
    own_pieces(gi_r, G, gr);
    own_pieces(gi_z, G, gz);
    own_pieces(gi_n, G, gn);
    float mk = mrow[G * SLAB + i];
    int buf = 0;
    // B operands by LOCAL k-step: [0], [1] = this wave's own features (global k-steps 2w, 2w+1), [2], [3] = the partner's
    // (2(w^1), 2(w^1)+1).  The k order of a dot product is free as long as the weight fragments follow it -- and a register
    // array indexed by the run-time `w` would be lowered to select chains (measured: 3x the VALU instructions).
    u32x4 x1[NJH], x2[NJH], x3[NJH];
    float mean_own = 0.f, m2_own = 0.f;
    // l = -1 is the hand-off of h~_0 = h_0 * mask_0; l = 0 .. L-1 are the steps (their tail hands h~_{l+1} over and
    // finishes y_l)
    for (int l = -1; l < L; ++l) {
      if (l >= 0) {
        const long slab = (long)l * groups + G;
        f32x16 ar, az, ah;
        float gnx[HR];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ar[r] = gr[r >> 2][r & 3];
          az[r] = gz[r >> 2][r & 3];
          gnx[r] = gn[r >> 2][r & 3];
          ah[r] = bhl[32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
        }
        if (l + 1 < L) {  // next step's input halves and mask: a whole step of latency to land
          own_pieces(gi_r, slab + groups, gr);
          own_pieces(gi_z, slab + groups, gz);
          own_pieces(gi_n, slab + groups, gn);
          mk = mrow[(slab + groups) * SLAB + i];
        } else {
          mk = 1.f;
        }
        // W_hh h~ for this wave's row tiles: 4 k-steps x 3 gates x 6 products, three independent accumulators in turn
        u32x4 wb[2][3];
        const int jg0 = 2 * w, jg1 = 2 * (w ^ 1);  // global k-step of local k-steps 0 / 2
#pragma unroll
        for (int term = 0; term < 3; ++term) wb[0][term] = wl[term * TS + ((0 * GT + w) * NJH + jg0) * 64];
#pragma unroll
        for (int sidx = 0; sidx < 3 * NJH; ++sidx) {
          const int j = sidx / 3, g = sidx % 3, cur = sidx & 1, nxt = cur ^ 1;
          if (sidx + 1 < 3 * NJH) {
            const int j1 = (sidx + 1) / 3, g1 = (sidx + 1) % 3;
            const int jglob = (j1 < 2 ? jg0 : jg1) + (j1 & 1);
#pragma unroll
            for (int term = 0; term < 3; ++term) wb[nxt][term] = wl[term * TS + ((g1 * GT + w) * NJH + jglob) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x16 &a = g == 0 ? ar : (g == 1 ? az : ah);
          a = mfma_bf16(wb[cur][2], x1[j], a);
          a = mfma_bf16(wb[cur][0], x3[j], a);
          a = mfma_bf16(wb[cur][1], x2[j], a);
          a = mfma_bf16(wb[cur][1], x1[j], a);
          a = mfma_bf16(wb[cur][0], x2[j], a);
          a = mfma_bf16(wb[cur][0], x1[j], a);
          __builtin_amdgcn_sched_barrier(0);
        }
        float sum = 0.f;
        float rgs[HR], zgs[HR], ngs[HR], hns[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) {
          const float rg = sigmoidf_(ar[r]);
          const float zg = sigmoidf_(az[r]);
          const float ng = tanhf_(gnx[r] + rg * ah[r]);
          rgs[r] = rg;
          zgs[r] = zg;
          ngs[r] = ng;
          hns[r] = ah[r];
          hs[r] = (1.f - zg) * ng + zg * hm[r];
          sum += hs[r];
        }
        if (r_s && live) {  // (wave-uniform) the backward's operands, as k_gru_fwd<true> leaves them
          store_own(r_s, slab, rgs);
          store_own(z_s, slab, zgs);
          store_own(n_s, slab, ngs);
          store_own(hn_s, slab, hns);
        }
        sum = wave_sum32(sum);
        mean_own = sum * (1.0f / 32.f);
        float vs = 0.f;
#pragma unroll
        for (int r = 0; r < HR; ++r) {
          const float d = hs[r] - mean_own;
          vs += d * d;
        }
        m2_own = wave_sum32(vs);
      }
      // ---- hand-off: h~ (own half) as split operands + LayerNorm partials
#pragma unroll
      for (int r = 0; r < HR; ++r) hm[r] = hs[r] * mk;
      if (hpm_s && live && l + 1 < L) store_own(hpm_s, (long)(l + 1) * groups + G, hm);  // h~_{l+1} = h_l * mask_{l+1}
      u32x4 o1[HR / 8], o2[HR / 8], o3[HR / 8];
      split_acts<HR>(hm, o1, o2, o3);
      u32x4 *xo = xch + (((buf * 2 + pair) * 2 + w) * 6) * 64 + lane;
      const u32x4 *xp = xch + (((buf * 2 + pair) * 2 + (w ^ 1)) * 6) * 64 + lane;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        xo[(0 * 2 + jj) * 64] = o1[jj];
        xo[(1 * 2 + jj) * 64] = o2[jj];
        xo[(2 * 2 + jj) * 64] = o3[jj];
      }
      float *so = xst + (((buf * 2 + pair) * 2 + w) * 64 + lane) * 2;
      const float *sp = xst + (((buf * 2 + pair) * 2 + (w ^ 1)) * 64 + lane) * 2;
      so[0] = mean_own;
      so[1] = m2_own;
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        x1[jj] = o1[jj];
        x2[jj] = o2[jj];
        x3[jj] = o3[jj];
        x1[2 + jj] = xp[(0 * 2 + jj) * 64];
        x2[2 + jj] = xp[(1 * 2 + jj) * 64];
        x3[2 + jj] = xp[(2 * 2 + jj) * 64];
      }
      if (l >= 0) {  // y_l = rnn.norm(h_l), own half
        const float mean_p = sp[0], m2_p = sp[1];
        const float dm = mean_own - mean_p;
        const float mean = 0.5f * (mean_own + mean_p);
        const float m2 = m2_own + m2_p + 16.f * dm * dm;
        const float rstd = 1.0f / sqrtf(m2 * (1.0f / GH) + 1e-5f);
        const long slab = (long)l * groups + G;
        if (live) {
          f32x4 *yp = reinterpret_cast<f32x4 *>(y + slab * (long)(GH * SLAB)) + lane;
#pragma unroll
          for (int q = 0; q < HR / 4; ++q)
            yp[(4 * w + q) * WAVE] = f32x4{(hs[4 * q] - mean) * rstd, (hs[4 * q + 1] - mean) * rstd, (hs[4 * q + 2] - mean) * rstd,
                                           (hs[4 * q + 3] - mean) * rstd};
          if (w == 0 && lane < 32) rstd_y[slab * SLAB + lane] = rstd;
        }
      }
      buf ^= 1;
    }
    if (h_last && live) {
#pragma unroll
      for (int q = 0; q < HR / 4; ++q) {
        const int qq = 4 * w + q;
        *reinterpret_cast<f32x4 *>(h_last + (G * SLAB + i) * GH + 32 * (qq >> 2) + 8 * (qq & 3) + 4 * h) =
            f32x4{hs[4 * q], hs[4 * q + 1], hs[4 * q + 2], hs[4 * q + 3]};
      }
    }
  }
}

// =============================================================================================
// forward, FOUR waves per slab (round 5): the latency variant above still spends 3.3 us per dependent step -- 72 MFMAs of 32
// cycles and 48 gate values (6 quarter-rate transcendentals each) per wave -- and the full-length log-prob passes of a
// recurrent policy are 160 such steps, sixteen passes per 8-agent update, eight of them in series with the updates (the next
// agent's factor needs them).  Here a workgroup is ONE slab and its four waves split the step two ways at once:
//   * wave (w, kh) multiplies the row tiles {g * 2 + w} (hidden features [32 w, 32 w + 32) of the three gates) by the k-steps
//     {2 kh, 2 kh + 1} of h~ only: 36 MFMAs, PARTIAL sums;
//   * the two waves of a feature half exchange the halves of their partial tiles through LDS (registers [8 (kh^1), +8) of each
//     of the three accumulators go out, the partner's [8 kh, +8) come in): every wave ends up with the complete pre-activations
//     of 8 accumulator registers = 16 features per sample, i.e. a QUARTER of the gate nonlinearities;
//   * its 8 registers of h_l are exactly k-step 2 w + kh of the next step's B operand: split into the three bf16 terms, handed
//     over through LDS together with the (mean, M2) of its 16 features; rnn.norm's statistics are merged four ways
//     (mean = sum mean_k / 4, M2 = sum M2_k + 16 sum (mean_k - mean)^2).
// Two workgroup barriers per step (partials / hand-off); both LDS areas are single-buffered: a wave writes the partials of step
// l + 1 only after the hand-off barrier of step l, behind which nobody reads step l's partials any more, and the hand-off of
// step l + 1 only after the partial barrier of step l + 1, which every reader of hand-off l has passed.
// Saves the backward's operands when asked (training chunks), like k_gru_fwd_tp.
// =============================================================================================
template <int KH>
__device__ __forceinline__ void quad_pack(const f32x16 &a, f32x4 (&snd)[2], float (&keep)[8]) {
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) keep[rr] = a[8 * KH + rr];
  snd[0] = f32x4{a[8 * (1 - KH) + 0], a[8 * (1 - KH) + 1], a[8 * (1 - KH) + 2], a[8 * (1 - KH) + 3]};
  snd[1] = f32x4{a[8 * (1 - KH) + 4], a[8 * (1 - KH) + 5], a[8 * (1 - KH) + 6], a[8 * (1 - KH) + 7]};
}

__global__ __launch_bounds__(WG_THREADS, 1) void k_gru_fwd_q(
    const float *__restrict__ gi_r, const float *__restrict__ gi_z, const float *__restrict__ gi_n,
    const float *__restrict__ mrow, const float *__restrict__ h0, const float *__restrict__ Whh,
    const float *__restrict__ bhh, int L, long m_pad, float *__restrict__ y, float *__restrict__ rstd_y,
    float *__restrict__ h_last, float *__restrict__ hpm_s, float *__restrict__ r_s, float *__restrict__ z_s,
    float *__restrict__ n_s, float *__restrict__ hn_s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MTH = 3 * GT, NJH = GH / 16, TS = MTH * NJH * 64, QR = 8;  // QR = registers per lane and wave
  u32x4 *Whimg = reinterpret_cast<u32x4 *>(lds);                   // [3 terms][6 tiles][4 k-steps][64 lanes] x 16 B
  u32x4 *xch = Whimg + 3 * TS;                                      // hand-off: [4 k-steps][3 terms][64 lanes]
  f32x4 *xpp = reinterpret_cast<f32x4 *>(xch + 4 * 3 * 64);         // partials: [4 waves][3 gates][2 pieces][64 lanes]
  float *xst = reinterpret_cast<float *>(xpp + 4 * 3 * 2 * 64);     // LayerNorm partials: [4 waves][64 lanes][2]
  float *bhl = xst + 4 * 64 * 2;                                    // b_hn [64]
  stage_split_matrix<3 * GH, GH, false, WG_THREADS>(Whimg, Whh);
  for (int e = threadIdx.x; e < GH; e += WG_THREADS) bhl[e] = bhh[2 * GH + e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), w = wave >> 1, kh = wave & 1;
  const int i = lane & 31, h = lane >> 5;
  const long groups = m_pad / SLAB;
  const u32x4 *wl = Whimg + lane;
  const int q0 = 4 * w + 2 * kh;  // this wave's two float4 pieces of an ATL(64) image = accumulator registers 16 w + 8 kh .. + 7
  float bown[QR];                 // b_hn of those registers' features
#pragma unroll
  for (int rr = 0; rr < QR; ++rr) {
    const int r = 8 * kh + rr;
    bown[rr] = bhl[32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
  }
  for (long G = blockIdx.x; G < groups; G += gridDim.x) {  // one slab per workgroup and iteration (uniform: barriers inside)
    float hs[QR], hm[QR];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int qq = q0 + q;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(h0 + (G * SLAB + i) * GH + 32 * (qq >> 2) + 8 * (qq & 3) + 4 * h);
      hs[4 * q + 0] = v[0];
      hs[4 * q + 1] = v[1];
      hs[4 * q + 2] = v[2];
      hs[4 * q + 3] = v[3];
    }
    auto own_pieces = [&](const float *base, long slab, f32x4 (&dst)[2]) {
      const f32x4 *pp = reinterpret_cast<const f32x4 *>(base + slab * (long)(GH * SLAB)) + lane;
      dst[0] = pp[(q0 + 0) * WAVE];
      dst[1] = pp[(q0 + 1) * WAVE];
    };
    auto store_own = [&](float *base, long slab, const float (&v)[QR]) {
      f32x4 *pp = reinterpret_cast<f32x4 *>(base + slab * (long)(GH * SLAB)) + lane;
      pp[(q0 + 0) * WAVE] = f32x4{v[0], v[1], v[2], v[3]};
      pp[(q0 + 1) * WAVE] = f32x4{v[4], v[5], v[6], v[7]};
    };
    f32x4 gr[2], gz[2], gn[2];
    own_pieces(gi_r, G, gr);
    own_pieces(gi_z, G, gz);
    own_pieces(gi_n, G, gn);
    float mk = mrow[G * SLAB + i];
    u32x4 x1[2], x2[2], x3[2];  // B operands of this wave's k-steps 2 kh, 2 kh + 1
    float mean_own = 0.f, m2_own = 0.f;
    for (int l = -1; l < L; ++l) {
      if (l >= 0) {
        const long slab = (long)l * groups + G;
        float pr[QR], pz[QR], pn[QR];
#pragma unroll
        for (int rr = 0; rr < QR; ++rr) {
          pr[rr] = gr[rr >> 2][rr & 3];
          pz[rr] = gz[rr >> 2][rr & 3];
          pn[rr] = gn[rr >> 2][rr & 3];
        }
        if (l + 1 < L) {  // next step's input halves and mask: a whole step of latency to land
          own_pieces(gi_r, slab + groups, gr);
          own_pieces(gi_z, slab + groups, gz);
          own_pieces(gi_n, slab + groups, gn);
          mk = mrow[(slab + groups) * SLAB + i];
        } else {
          mk = 1.f;
        }
        // partial W_hh h~: 3 gates x 2 k-steps x 6 products
        f32x16 ar, az, ah;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ar[r] = 0.f;
          az[r] = 0.f;
          ah[r] = 0.f;
        }
        u32x4 wb[2][3];
#pragma unroll
        for (int term = 0; term < 3; ++term) wb[0][term] = wl[term * TS + ((0 * GT + w) * NJH + 2 * kh) * 64];
#pragma unroll
        for (int sidx = 0; sidx < 6; ++sidx) {
          const int j = sidx / 3, g = sidx % 3, cur = sidx & 1, nxt = cur ^ 1;
          if (sidx + 1 < 6) {
            const int j1 = (sidx + 1) / 3, g1 = (sidx + 1) % 3;
#pragma unroll
            for (int term = 0; term < 3; ++term) wb[nxt][term] = wl[term * TS + ((g1 * GT + w) * NJH + 2 * kh + j1) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x16 &a = g == 0 ? ar : (g == 1 ? az : ah);
          a = mfma_bf16(wb[cur][2], x1[j], a);
          a = mfma_bf16(wb[cur][0], x3[j], a);
          a = mfma_bf16(wb[cur][1], x2[j], a);
          a = mfma_bf16(wb[cur][1], x1[j], a);
          a = mfma_bf16(wb[cur][0], x2[j], a);
          a = mfma_bf16(wb[cur][0], x1[j], a);
          __builtin_amdgcn_sched_barrier(0);
        }
        // exchange the partial tiles' halves with the other k-half of this feature half (wave ^ 1)
        f32x4 snd[3][2];
        float kr[QR], kz[QR], kn[QR];
        if (kh == 0) {  // (wave-uniform: register indices stay compile-time constants in either branch)
          quad_pack<0>(ar, snd[0], kr);
          quad_pack<0>(az, snd[1], kz);
          quad_pack<0>(ah, snd[2], kn);
        } else {
          quad_pack<1>(ar, snd[0], kr);
          quad_pack<1>(az, snd[1], kz);
          quad_pack<1>(ah, snd[2], kn);
        }
        f32x4 *po = xpp + (wave * 3 * 2) * 64 + lane;
        const f32x4 *pi = xpp + ((wave ^ 1) * 3 * 2) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          po[(g * 2 + 0) * 64] = snd[g][0];
          po[(g * 2 + 1) * 64] = snd[g][1];
        }
        __syncthreads();
        float sum = 0.f;
        float rgs[QR], zgs[QR], ngs[QR], hns[QR];
        {
          f32x4 rc[3][2];
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            rc[g][0] = pi[(g * 2 + 0) * 64];
            rc[g][1] = pi[(g * 2 + 1) * 64];
          }
#pragma unroll
          for (int rr = 0; rr < QR; ++rr) {
            const float rg = sigmoidf_(pr[rr] + (kr[rr] + rc[0][rr >> 2][rr & 3]));
            const float zg = sigmoidf_(pz[rr] + (kz[rr] + rc[1][rr >> 2][rr & 3]));
            const float hn = bown[rr] + (kn[rr] + rc[2][rr >> 2][rr & 3]);
            const float ng = tanhf_(pn[rr] + rg * hn);
            rgs[rr] = rg;
            zgs[rr] = zg;
            ngs[rr] = ng;
            hns[rr] = hn;
            hs[rr] = (1.f - zg) * ng + zg * hm[rr];
            sum += hs[rr];
          }
        }
        if (r_s) {  // (kernel-uniform) the backward's operands, as k_gru_fwd<true> leaves them
          store_own(r_s, slab, rgs);
          store_own(z_s, slab, zgs);
          store_own(n_s, slab, ngs);
          store_own(hn_s, slab, hns);
        }
        sum = wave_sum32(sum);
        mean_own = sum * (1.0f / 16.f);
        float vs = 0.f;
#pragma unroll
        for (int rr = 0; rr < QR; ++rr) {
          const float d = hs[rr] - mean_own;
          vs += d * d;
        }
        m2_own = wave_sum32(vs);
      }
      // ---- hand-off: h~ (own quarter) as the split operands of k-step 2 w + kh, and the LayerNorm partials
#pragma unroll
      for (int rr = 0; rr < QR; ++rr) hm[rr] = hs[rr] * mk;
      if (hpm_s && l + 1 < L) store_own(hpm_s, (long)(l + 1) * groups + G, hm);  // h~_{l+1} = h_l * mask_{l+1}
      {
        u32x4 o1[1], o2[1], o3[1];
        split_acts<QR>(hm, o1, o2, o3);
        u32x4 *xo = xch + ((2 * w + kh) * 3) * 64 + lane;
        xo[0 * 64] = o1[0];
        xo[1 * 64] = o2[0];
        xo[2 * 64] = o3[0];
        float *so = xst + (wave * 64 + lane) * 2;
        so[0] = mean_own;
        so[1] = m2_own;
      }
      __syncthreads();
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const u32x4 *xi = xch + ((2 * kh + jj) * 3) * 64 + lane;
        x1[jj] = xi[0 * 64];
        x2[jj] = xi[1 * 64];
        x3[jj] = xi[2 * 64];
      }
      if (l >= 0) {  // y_l = rnn.norm(h_l), own quarter
        float mk_[4], m2_[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          mk_[k] = xst[(k * 64 + lane) * 2 + 0];
          m2_[k] = xst[(k * 64 + lane) * 2 + 1];
        }
        const float mean = 0.25f * ((mk_[0] + mk_[1]) + (mk_[2] + mk_[3]));
        float dev2 = 0.f, m2 = (m2_[0] + m2_[1]) + (m2_[2] + m2_[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d = mk_[k] - mean;
          dev2 += d * d;
        }
        m2 += 16.f * dev2;
        const float rstd = 1.0f / sqrtf(m2 * (1.0f / GH) + 1e-5f);
        const long slab = (long)l * groups + G;
        float yo[QR];
#pragma unroll
        for (int rr = 0; rr < QR; ++rr) yo[rr] = (hs[rr] - mean) * rstd;
        store_own(y, slab, yo);
        if (wave == 0 && lane < 32) rstd_y[slab * SLAB + lane] = rstd;
      }
    }
    if (h_last) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int qq = q0 + q;
        *reinterpret_cast<f32x4 *>(h_last + (G * SLAB + i) * GH + 32 * (qq >> 2) + 8 * (qq & 3) + 4 * h) =
            f32x4{hs[4 * q], hs[4 * q + 1], hs[4 * q + 2], hs[4 * q + 3]};
      }
    }
    __syncthreads();  // the next slab's first hand-off overwrites areas the slowest wave may still be reading
  }
}

// =============================================================================================
// backward (BPTT over the chunk, reverse in l).  Input: dh_out = d(loss)/d(h_l) through the output path (the head
// kernels already applied the rnn.norm backward); output: the four gate-gradient tensors (for the weight-gradient
// kernel) and dz of the last MLP layer (LayerNorm/ReLU backward of d(loss)/d(x_hat_mlp) applied here).
//   A operands are W^T: lane i -> input feature, step -> gate output f(R,h); row-major W in LDS reads conflict-free.
// =============================================================================================

// The recurrence carries only d h~ = W_hh^T [dr, dz, dhn] (144 bf16 MFMAs per step on the split images of W_hh^T); the input
// side, d x_hat_mlp = W_ih'^T [dr, dz, dn] followed by the LayerNorm/ReLU backward of the last MLP layer, has no recurrence
// and runs afterwards over all L x groups slabs in parallel (k_gru_dx) from the gate gradients this kernel stores anyway.
// (Before: both products on the fp32 MFMA with one LDS read per MFMA inside the chain, 17 us per step.)
__global__ __launch_bounds__(WG_THREADS, 1) void k_gru_bwd(
    const float *__restrict__ dhout, const float *__restrict__ mrow, const float *__restrict__ Whh,
    const float *__restrict__ hpm_s, const float *__restrict__ r_s, const float *__restrict__ z_s,
    const float *__restrict__ n_s, const float *__restrict__ hn_s, int L, long m_pad, float *__restrict__ dr_s,
    float *__restrict__ dz_s, float *__restrict__ dn_s, float *__restrict__ dhn_s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = GT, NJ = 3 * GH / 16;  // out = 64 hidden features (2 tiles), k = 192 gate outputs (12 k-steps)
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  stage_split_matrix<3 * GH, GH, true, WG_THREADS>(img, Whh);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31;
  const long groups = m_pad / SLAB;
  const u32x4 *wl = img + lane;
  for (long G = (long)blockIdx.x * WAVES_PER_WG + wave; G < groups; G += (long)gridDim.x * WAVES_PER_WG) {
    float dcarry[GR];
#pragma unroll
    for (int R = 0; R < GR; ++R) dcarry[R] = 0.f;
    float dh[GR], rg[GR], zg[GR], ng[GR], hn[GR], hp[GR];
    {
      const long slab = (long)(L - 1) * groups + G;
      load_act(dhout, slab, lane, dh);
      load_act(r_s, slab, lane, rg);
      load_act(z_s, slab, lane, zg);
      load_act(n_s, slab, lane, ng);
      load_act(hn_s, slab, lane, hn);
      load_act(hpm_s, slab, lane, hp);
    }
    float mk = mrow[((long)(L - 1) * groups + G) * SLAB + i];
    for (int l = L - 1; l >= 0; --l) {
      const long slab = (long)l * groups + G;
      float gates[3 * GR], dnp[GR], dhp[GR];  // gates = [dr | dz | dhn]: the 192-wide B operand of W_hh^T
#pragma unroll
      for (int R = 0; R < GR; ++R) {
        const float d = dh[R] + dcarry[R];
        const float dz_ = d * (hp[R] - ng[R]);
        const float dn_ = d * (1.f - zg[R]);
        dhp[R] = d * zg[R];
        dnp[R] = dn_ * (1.f - ng[R] * ng[R]);
        gates[2 * GR + R] = dnp[R] * rg[R];
        gates[R] = (dnp[R] * hn[R]) * rg[R] * (1.f - rg[R]);
        gates[GR + R] = dz_ * zg[R] * (1.f - zg[R]);
      }
      const float mk_cur = mk;
      {
        float t[GR];
#pragma unroll
        for (int R = 0; R < GR; ++R) t[R] = gates[R];
        store_act(dr_s, slab, lane, t);
#pragma unroll
        for (int R = 0; R < GR; ++R) t[R] = gates[GR + R];
        store_act(dz_s, slab, lane, t);
        store_act(dn_s, slab, lane, dnp);
#pragma unroll
        for (int R = 0; R < GR; ++R) t[R] = gates[2 * GR + R];
        store_act(dhn_s, slab, lane, t);
      }
      // d h~ += W_hh^T [dr, dz, dhn] ;  carry = d h~ * mask_l
      u32x4 g1[NJ], g2[NJ], g3[NJ];
      split_acts<3 * GR>(gates, g1, g2, g3);
      __builtin_amdgcn_sched_barrier(0);
      if (l > 0) {  // the previous step's operands (the gate registers are dead now): the GEMM's time to land
        const long sp = slab - groups;
        load_act(dhout, sp, lane, dh);
        load_act(r_s, sp, lane, rg);
        load_act(z_s, sp, lane, zg);
        load_act(n_s, sp, lane, ng);
        load_act(hn_s, sp, lane, hn);
        load_act(hpm_s, sp, lane, hp);
        mk = mrow[sp * SLAB + i];
      }
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      split_gemm<MT, NJ>(wl, g1, g2, g3, acc, [](int) {});
#pragma unroll
      for (int R = 0; R < GR; ++R) dcarry[R] = (dhp[R] + acc[R >> 4][R & 15]) * mk_cur;
    }
  }
}

// Four waves per slab (round 5), the backward twin of k_gru_fwd_q: wave q owns the accumulator registers R = 8 q .. 8 q + 7 (16
// hidden features per sample) of EVERYTHING that is per-feature -- the carry, the saved gates, the four gate gradients it stores --
// and therefore the k-steps {q, 4 + q, 8 + q} of the 192-wide operand [dr | dz | dhn] of W_hh^T: 36 of the step's 144 MFMAs,
// a quarter of the gate arithmetic and of the operand split, 12 instead of 48 row loads.  Its two accumulator tiles are PARTIAL
// sums over its k-steps: all four waves park them in LDS (8 float4 pieces each), one barrier, and every wave adds up the four
// partials of its own two pieces, in wave order.  Two buffers: the partials of step l + 1 go where step l - 1's were, which
// every wave has finished reading when it arrives at step l's barrier.  One workgroup = one slab: the 256 chains of a
// chunked SMAC minibatch (8192 sequences x 10 steps) occupy 256 CUs instead of 64.
__global__ __launch_bounds__(WG_THREADS, 1) void k_gru_bwd_q(
    const float *__restrict__ dhout, const float *__restrict__ mrow, const float *__restrict__ Whh,
    const float *__restrict__ hpm_s, const float *__restrict__ r_s, const float *__restrict__ z_s,
    const float *__restrict__ n_s, const float *__restrict__ hn_s, int L, long m_pad, float *__restrict__ dr_s,
    float *__restrict__ dz_s, float *__restrict__ dn_s, float *__restrict__ dhn_s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = GT, NJ = 3 * GH / 16, TS = MT * NJ * 64, QR = 8;
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  f32x4 *xp = reinterpret_cast<f32x4 *>(img + 3 * TS);  // [2 buffers][4 waves][8 pieces][64 lanes]
  stage_split_matrix<3 * GH, GH, true, WG_THREADS>(img, Whh);
  __syncthreads();
  const int lane = threadIdx.x & 63, q = wave_id();
  const int i = lane & 31;
  const long groups = m_pad / SLAB;
  const u32x4 *wl = img + lane;
  int buf = 0;
  for (long G = blockIdx.x; G < groups; G += gridDim.x) {  // uniform: barriers inside
    auto own = [&](const float *base, long slab, float (&dst)[QR]) {
      const f32x4 *pp = reinterpret_cast<const f32x4 *>(base + slab * (long)(GH * SLAB)) + lane;
      const f32x4 a = pp[(2 * q) * WAVE], b = pp[(2 * q + 1) * WAVE];
      dst[0] = a[0]; dst[1] = a[1]; dst[2] = a[2]; dst[3] = a[3];
      dst[4] = b[0]; dst[5] = b[1]; dst[6] = b[2]; dst[7] = b[3];
    };
    auto store_own = [&](float *base, long slab, const float (&v)[QR]) {
      f32x4 *pp = reinterpret_cast<f32x4 *>(base + slab * (long)(GH * SLAB)) + lane;
      pp[(2 * q) * WAVE] = f32x4{v[0], v[1], v[2], v[3]};
      pp[(2 * q + 1) * WAVE] = f32x4{v[4], v[5], v[6], v[7]};
    };
    float dcarry[QR];
#pragma unroll
    for (int rr = 0; rr < QR; ++rr) dcarry[rr] = 0.f;
    float dh[QR], rg[QR], zg[QR], ng[QR], hn[QR], hp[QR];
    {
      const long slab = (long)(L - 1) * groups + G;
      own(dhout, slab, dh);
      own(r_s, slab, rg);
      own(z_s, slab, zg);
      own(n_s, slab, ng);
      own(hn_s, slab, hn);
      own(hpm_s, slab, hp);
    }
    float mk = mrow[((long)(L - 1) * groups + G) * SLAB + i];
    for (int l = L - 1; l >= 0; --l) {
      const long slab = (long)l * groups + G;
      float g_r[QR], g_z[QR], g_hn[QR], dnp[QR], dhp[QR];
#pragma unroll
      for (int rr = 0; rr < QR; ++rr) {
        const float d = dh[rr] + dcarry[rr];
        const float dz_ = d * (hp[rr] - ng[rr]);
        const float dn_ = d * (1.f - zg[rr]);
        dhp[rr] = d * zg[rr];
        dnp[rr] = dn_ * (1.f - ng[rr] * ng[rr]);
        g_hn[rr] = dnp[rr] * rg[rr];
        g_r[rr] = (dnp[rr] * hn[rr]) * rg[rr] * (1.f - rg[rr]);
        g_z[rr] = dz_ * zg[rr] * (1.f - zg[rr]);
      }
      const float mk_cur = mk;
      store_own(dr_s, slab, g_r);
      store_own(dz_s, slab, g_z);
      store_own(dn_s, slab, dnp);
      store_own(dhn_s, slab, g_hn);
      // this wave's three k-steps of [dr | dz | dhn]: q, 4 + q, 8 + q
      u32x4 b1[3], b2[3], b3[3];
      {
        u32x4 t1[1], t2[1], t3[1];
        split_acts<QR>(g_r, t1, t2, t3);
        b1[0] = t1[0]; b2[0] = t2[0]; b3[0] = t3[0];
        split_acts<QR>(g_z, t1, t2, t3);
        b1[1] = t1[0]; b2[1] = t2[0]; b3[1] = t3[0];
        split_acts<QR>(g_hn, t1, t2, t3);
        b1[2] = t1[0]; b2[2] = t2[0]; b3[2] = t3[0];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (l > 0) {  // the previous step's operands: the GEMM's and the exchange's time to land
        const long sp = slab - groups;
        own(dhout, sp, dh);
        own(r_s, sp, rg);
        own(z_s, sp, zg);
        own(n_s, sp, ng);
        own(hn_s, sp, hn);
        own(hpm_s, sp, hp);
        mk = mrow[sp * SLAB + i];
      }
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      {
        u32x4 wb[2][3];
#pragma unroll
        for (int term = 0; term < 3; ++term) wb[0][term] = wl[term * TS + (0 * NJ + q) * 64];
#pragma unroll
        for (int sidx = 0; sidx < 3 * MT; ++sidx) {
          const int ty = sidx / MT, t = sidx % MT, cur = sidx & 1, nxt = cur ^ 1;
          if (sidx + 1 < 3 * MT) {
            const int ty1 = (sidx + 1) / MT, t1 = (sidx + 1) % MT;
#pragma unroll
            for (int term = 0; term < 3; ++term) wb[nxt][term] = wl[term * TS + (t1 * NJ + 4 * ty1 + q) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[t] = mfma_bf16(wb[cur][2], b1[ty], acc[t]);
          acc[t] = mfma_bf16(wb[cur][0], b3[ty], acc[t]);
          acc[t] = mfma_bf16(wb[cur][1], b2[ty], acc[t]);
          acc[t] = mfma_bf16(wb[cur][1], b1[ty], acc[t]);
          acc[t] = mfma_bf16(wb[cur][0], b2[ty], acc[t]);
          acc[t] = mfma_bf16(wb[cur][0], b1[ty], acc[t]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      f32x4 *po = xp + ((buf * 4 + q) * 8) * 64 + lane;
#pragma unroll
      for (int p8 = 0; p8 < 8; ++p8)
        po[p8 * 64] = f32x4{acc[p8 >> 2][4 * (p8 & 3) + 0], acc[p8 >> 2][4 * (p8 & 3) + 1], acc[p8 >> 2][4 * (p8 & 3) + 2],
                            acc[p8 >> 2][4 * (p8 & 3) + 3]};
      __syncthreads();
      {
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const f32x4 *pi = xp + ((buf * 4 + k) * 8) * 64 + lane;
          s0[k] = pi[(2 * q) * 64];
          s1[k] = pi[(2 * q + 1) * 64];
        }
        const f32x4 a0 = (s0[0] + s0[1]) + (s0[2] + s0[3]), a1 = (s1[0] + s1[1]) + (s1[2] + s1[3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          dcarry[c] = (dhp[c] + a0[c]) * mk_cur;
          dcarry[4 + c] = (dhp[4 + c] + a1[c]) * mk_cur;
        }
      }
      buf ^= 1;
    }
  }
}

// d x_hat_mlp = W_ih'^T [dr, dz, dn]  ->  LayerNorm/ReLU backward of the last MLP layer -> dz_mlp, every slab independent
__global__ __launch_bounds__(WG_THREADS, 2) void k_gru_dx(const float *__restrict__ Wih, const float *__restrict__ dr_s,
                                                          const float *__restrict__ dz_s, const float *__restrict__ dn_s,
                                                          const float *__restrict__ xmlp, const uint32_t *__restrict__ mask_mlp,
                                                          const float *__restrict__ rstd_mlp, float *__restrict__ dz_mlp,
                                                          long n_slabs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = GT, NJ = 3 * GH / 16;
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  stage_split_matrix<3 * GH, GH, true, WG_THREADS>(img, Wih);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31;
  const u32x4 *wl = img + lane;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float gates[3 * GR];
    {
      float t[GR];
      load_act(dr_s, slab, lane, t);
#pragma unroll
      for (int R = 0; R < GR; ++R) gates[R] = t[R];
      load_act(dz_s, slab, lane, t);
#pragma unroll
      for (int R = 0; R < GR; ++R) gates[GR + R] = t[R];
      load_act(dn_s, slab, lane, t);
#pragma unroll
      for (int R = 0; R < GR; ++R) gates[2 * GR + R] = t[R];
    }
    float xh[GR];
    load_act(xmlp, slab, lane, xh);
    const float rstd = rstd_mlp[slab * SLAB + i];
    constexpr int NWM = (GH / 2 + 31) / 32;
    uint32_t mb[NWM];
#pragma unroll
    for (int w = 0; w < NWM; ++w) mb[w] = mask_mlp[(slab * NWM + w) * WAVE + lane];
    u32x4 g1[NJ], g2[NJ], g3[NJ];
    split_acts<3 * GR>(gates, g1, g2, g3);
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    split_gemm<MT, NJ>(wl, g1, g2, g3, acc, [](int) {});
    float dx[GR], out[GR];
#pragma unroll
    for (int R = 0; R < GR; ++R) dx[R] = acc[R >> 4][R & 15];
    ln_bwd_relu_mbits<GH>(dx, xh, mb, rstd, out);
    store_act(dz_mlp, slab, lane, out);
  }
}
}  // namespace

// =============================================================================================
// Forward-mode tangent through the recurrence (HATRPO's Fisher-vector product J v for GRU policies).
//   a_r = W_ir x + W_hr h~ + b  ->  a_r' = [W_ir x' + W_ir' x + W_hr' h~] + W_hr h~' + b_ir' + b_hr' ;  r' = r (1 - r) a_r'
//   hn  = W_hn h~ + b_hn        ->  hn'  = [W_hn' h~] + W_hn h~' + b_hn'
//   a_n = W_in x + b_in + r hn  ->  a_n' = [W_in x' + W_in' x] + b_in' + r' hn + r hn' ;       n' = (1 - n^2) a_n'
//   h   = (1 - z) n + z h~      ->  h'   = (1 - z) n' - z' n + z' h~ + z h~'
//   y   = norm(h)               ->  y'   = rstd (h' - mean(h') - y mean(h' y))
// The bracketed terms have no recurrence: harl_gru_gates computes them for all steps in parallel (g_r, g_z, g_nx, g_nh);
// this kernel carries h' in registers and only multiplies W_hh h~' per step (the same bf16 images as the forward).
// =============================================================================================
__global__ __launch_bounds__(WG_THREADS, 1) void k_gru_gates_lin(const float *__restrict__ xin, const float *__restrict__ W,
                                                                 long n_slabs, float *__restrict__ o_r,
                                                                 float *__restrict__ o_z, float *__restrict__ o_n,
                                                                 int acc_mask) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LDW = GH + 1;
  float *Wl = lds;  // [192][65]
  stage_matrix<3 * GH, GH, LDW, WG_THREADS>(Wl, W);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const float *w_lane = Wl + i * LDW + 4 * h;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float x[GR];
    load_act(xin, slab, lane, x);
    f32x16 acc[3][GT];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float *og = g == 0 ? o_r : (g == 1 ? o_z : o_n);
      float prev[GR];
      if ((acc_mask >> g) & 1) load_act(og, slab, lane, prev);
#pragma unroll
      for (int R = 0; R < GR; ++R) acc[g][R >> 4][R & 15] = ((acc_mask >> g) & 1) ? prev[R] : 0.f;
    }
    gemm_gates<7>(acc, w_lane, x);
    float o[GR];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
      for (int R = 0; R < GR; ++R) o[R] = acc[g][R >> 4][R & 15];
      store_act(g == 0 ? o_r : (g == 1 ? o_z : o_n), slab, lane, o);
    }
  }
}

__global__ __launch_bounds__(WG_THREADS, 1) void k_gru_tangent(
    const float *__restrict__ g_r, const float *__restrict__ g_z, const float *__restrict__ g_nx,
    const float *__restrict__ g_nh, const float *__restrict__ mrow, const float *__restrict__ Whh,
    const float *__restrict__ bihd, const float *__restrict__ bhhd, const float *__restrict__ hpm_s,
    const float *__restrict__ r_s, const float *__restrict__ z_s, const float *__restrict__ n_s,
    const float *__restrict__ hn_s, const float *__restrict__ y_s, const float *__restrict__ rstd_y, int L, long m_pad,
    float *__restrict__ ydot) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MTH = 3 * GT, NJH = GH / 16;
  u32x4 *Whimg = reinterpret_cast<u32x4 *>(lds);
  float *bl = reinterpret_cast<float *>(Whimg + 3 * MTH * NJH * 64);  // [192] b_i' + b_h' (r, z) / b_i' (n) ; [192..256) b_hn'
  stage_split_matrix<3 * GH, GH, false, WG_THREADS>(Whimg, Whh);
  for (int e = threadIdx.x; e < 3 * GH; e += WG_THREADS) bl[e] = bihd[e] + (e < 2 * GH ? bhhd[e] : 0.f);
  for (int e = threadIdx.x; e < GH; e += WG_THREADS) bl[3 * GH + e] = bhhd[2 * GH + e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5, i = lane & 31;
  const long groups = m_pad / SLAB;
  const u32x4 *wh_img = Whimg + lane;
  for (long G = (long)blockIdx.x * WAVES_PER_WG + wave; G < groups; G += (long)gridDim.x * WAVES_PER_WG) {
    float hd[GR];
#pragma unroll
    for (int R = 0; R < GR; ++R) hd[R] = 0.f;  // the initial hidden state is data
    for (int l = 0; l < L; ++l) {
      const long slab = (long)l * groups + G;
      const float mk = mrow[slab * SLAB + i];
      // the 128 bias values a lane touches are loop invariant; hoisted into registers they push the kernel into spills,
      // so the LDS offset is made opaque per step (re-reading them costs 128 ds_read_b32 against ~250 global loads)
      int bo = 4 * h;
      asm volatile("" : "+v"(bo));
      const float *blh = bl + bo;
#pragma unroll
      for (int R = 0; R < GR; ++R) hd[R] *= mk;  // h~' = h' * mask
      f32x16 a6[MTH];
      {
        float gx[GR];
        load_act(g_r, slab, lane, gx);
#pragma unroll
        for (int R = 0; R < GR; ++R) a6[0 + (R >> 4)][R & 15] = gx[R] + blh[0 * GH + feat_base(R)];
        load_act(g_z, slab, lane, gx);
#pragma unroll
        for (int R = 0; R < GR; ++R) a6[GT + (R >> 4)][R & 15] = gx[R] + blh[1 * GH + feat_base(R)];
        load_act(g_nh, slab, lane, gx);
#pragma unroll
        for (int R = 0; R < GR; ++R) a6[2 * GT + (R >> 4)][R & 15] = gx[R] + blh[3 * GH + feat_base(R)];
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the seven primal loads below from being hoisted over the GEMM (spills)
      {
        u32x4 h1[NJH], h2[NJH], h3[NJH];
        split_acts<GR>(hd, h1, h2, h3);
        split_gemm<MTH, NJH>(wh_img, h1, h2, h3, a6, [](int) {});
      }
      __builtin_amdgcn_sched_barrier(0);
      // gate Jacobians piece by piece (one float4 of every saved tensor at a time: 28 live registers instead of 224)
      auto piece = [&](const float *base, int q) { return (reinterpret_cast<const f32x4 *>(base + slab * (long)(GH * SLAB)) + lane)[q * WAVE]; };
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < GR / 4; ++q) {
        const f32x4 rg = piece(r_s, q), zg = piece(z_s, q), ng = piece(n_s, q), hn = piece(hn_s, q), hp = piece(hpm_s, q),
                    gnx = piece(g_nx, q), yv = piece(y_s, q);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int R = 4 * q + c;
          const float rd = rg[c] * (1.f - rg[c]) * a6[0 + (R >> 4)][R & 15];
          const float zd = zg[c] * (1.f - zg[c]) * a6[GT + (R >> 4)][R & 15];
          const float hnd = a6[2 * GT + (R >> 4)][R & 15];
          const float and_ = gnx[c] + blh[2 * GH + feat_base(R)] + rd * hn[c] + rg[c] * hnd;
          const float nd = (1.f - ng[c] * ng[c]) * and_;
          hd[R] = (1.f - zg[c]) * nd - zd * ng[c] + zd * hp[c] + zg[c] * hd[R];
          s1 += hd[R];
          s2 += hd[R] * yv[c];
        }
      }
      s1 = wave_sum32(s1);
      s2 = wave_sum32(s2);
      s1 *= (1.0f / GH);
      s2 *= (1.0f / GH);
      const float rstd = rstd_y[slab * SLAB + i];
      float yo[GR];
#pragma unroll
      for (int q = 0; q < GR / 4; ++q) {
        const f32x4 yv = piece(y_s, q);
#pragma unroll
        for (int c = 0; c < 4; ++c) yo[4 * q + c] = rstd * (hd[4 * q + c] - s1 - yv[c] * s2);
      }
      store_act(ydot, slab, lane, yo);
    }
  }
}

extern "C" int harl_gru_gates(const float *xin, const float *W, int H, long n_slabs, float *out_r, float *out_z,
                              float *out_n, int acc_mask, void *stream) {
  if (n_slabs <= 0) return 0;
  if (H != GH) { set_error("harl_gru_gates: hidden width must be 64"); return -2; }
  const size_t shm = (size_t)3 * GH * (GH + 1) * sizeof(float);
  allow_big_lds(k_gru_gates_lin, shm);
  hipLaunchKernelGGL(k_gru_gates_lin, dim3(persistent_grid(n_slabs, 1)), dim3(WG_THREADS), shm, (hipStream_t)stream, xin, W,
                     n_slabs, out_r, out_z, out_n, acc_mask);
  return check_launch("harl_gru_gates");
}

extern "C" int harl_gru_tangent(const float *g_r, const float *g_z, const float *g_nx, const float *g_nh,
                                const float *mask_rows, const float *Whh, const float *bihd, const float *bhhd,
                                const float *hpm, const float *r, const float *z, const float *n, const float *hn,
                                const float *y, const float *rstd_y, int H, int L, long m_pad, float *ydot, void *stream) {
  if (L <= 0 || m_pad <= 0) return 0;
  if (H != GH) { set_error("harl_gru_tangent: hidden width must be 64"); return -2; }
  if (m_pad % SLAB) { set_error("harl_gru_tangent: m_pad must be a multiple of 32"); return -2; }
  const long groups = m_pad / SLAB;
  long wgs = (groups + WAVES_PER_WG - 1) / WAVES_PER_WG;
  const int grid = (int)(wgs < 256 ? wgs : 256);
  const size_t shm = split_image_bytes(3 * GH, GH) + (size_t)4 * GH * sizeof(float);
  allow_big_lds(k_gru_tangent, shm);
  hipLaunchKernelGGL(k_gru_tangent, dim3(grid), dim3(WG_THREADS), shm, (hipStream_t)stream, g_r, g_z, g_nx, g_nh, mask_rows,
                     Whh, bihd, bhhd, hpm, r, z, n, hn, y, rstd_y, L, m_pad, ydot);
  return check_launch("harl_gru_tangent");
}

extern "C" int harl_gru_fwd(const float *xin, const float *mask_rows, const float *h0, const float *Wih,
                            const float *bih, const float *Whh, const float *bhh, int H, int L, long m_pad, float *y,
                            float *rstd_y, float *hpm, float *r, float *z, float *n, float *hn, float *h_last, int save,
                            float *gi_ws, void *stream) {
  if (L <= 0 || m_pad <= 0) return 0;
  if (H != GH) { set_error("harl_gru_fwd: hidden width must be 64"); return -2; }
  if (m_pad % SLAB) { set_error("harl_gru_fwd: m_pad must be a multiple of 32"); return -2; }
  const long groups = m_pad / SLAB;
  long wgs = (groups + WAVES_PER_WG - 1) / WAVES_PER_WG;
  const int grid = (int)(wgs < 256 ? wgs : 256);
  hipStream_t s = (hipStream_t)stream;
  // bit 1 of `save`: gi_ws already holds the input half of the gates (harl_mlp_fwd_trunk wrote it from the last MLP layer's
  // registers: the same product, bit for bit) -- the parallel gate launch is skipped, xin is not read
  const bool gates_done = (save & 2) != 0;
  save &= 1;
  if (gates_done && !gi_ws) { set_error("harl_gru_fwd: save bit 1 (gates precomputed) needs gi_ws"); return -2; }
  if (gi_ws) {  // two phases: parallel x half over all L*groups slabs, then the recurrence with W_hh only
    const long n_slabs = (long)L * groups, M = n_slabs * SLAB;
    float *gr = gi_ws, *gz = gi_ws + M * GH, *gn = gi_ws + 2 * M * GH;
    static const bool gates_f32 = [] { const char *e = getenv("HARL_GRU_GATES_F32"); return e && e[0] == '1'; }();
    if (gates_done) {
    } else if (gates_f32) {
      const size_t shm_x = ((size_t)3 * GH * (GH + 1) + 3 * GH) * sizeof(float);
      allow_big_lds(k_gru_gates_x, shm_x);
      hipLaunchKernelGGL(k_gru_gates_x, dim3(persistent_grid(n_slabs, 1)), dim3(WG_THREADS), shm_x, s, xin, Wih, bih, bhh,
                         n_slabs, gr, gz, gn);
    } else {
      const size_t shm_x = split_image_bytes(3 * GH, GH) + (size_t)3 * GH * sizeof(float);
      allow_big_lds(k_gru_gates_xs, shm_x);
      hipLaunchKernelGGL(k_gru_gates_xs, dim3(persistent_grid(n_slabs, 2)), dim3(WG_THREADS), shm_x, s, xin, Wih, bih, bhh,
                         n_slabs, gr, gz, gn);
    }
    // few dependent chains: two waves per slab (k_gru_fwd_tp); HARL_GRU_TP_SAVE=0 keeps training forwards on the one-wave kernel (A/B)
    static const bool tp_save = [] { const char *e = getenv("HARL_GRU_TP_SAVE"); return !(e && e[0] == '0'); }();
    // four waves per slab (k_gru_fwd_q) while a slab per CU does not queue: up to 256 chains.  HARL_GRU_QUAD=0: the two-wave kernel
    static const bool quad = [] { const char *e = getenv("HARL_GRU_QUAD"); return !(e && e[0] == '0'); }();
    if (quad && groups <= 256 && (!save || tp_save)) {
      const size_t shm_q = split_image_bytes(3 * GH, GH) + (size_t)4 * 3 * 64 * 16 + (size_t)4 * 3 * 2 * 64 * 16 +
                           ((size_t)4 * 64 * 2 + GH) * sizeof(float);
      allow_big_lds(k_gru_fwd_q, shm_q);
      hipLaunchKernelGGL(k_gru_fwd_q, dim3((unsigned)groups), dim3(WG_THREADS), shm_q, s, gr, gz, gn, mask_rows, h0, Whh, bhh, L,
                         m_pad, y, rstd_y, h_last, save ? hpm : nullptr, save ? r : nullptr, save ? z : nullptr,
                         save ? n : nullptr, save ? hn : nullptr);
      return check_launch("harl_gru_fwd");
    }
    if (groups <= GRU_TP_MAX_GROUPS && (!save || tp_save)) {
      const size_t shm_tp = split_image_bytes(3 * GH, GH) + (size_t)2 * 2 * 2 * 6 * 64 * 16 + ((size_t)2 * 2 * 2 * 64 * 2 + GH) * sizeof(float);
      allow_big_lds(k_gru_fwd_tp, shm_tp);
      const long wg2 = (groups + 1) / 2;
      hipLaunchKernelGGL(k_gru_fwd_tp, dim3((unsigned)(wg2 < 256 ? wg2 : 256)), dim3(WG_THREADS), shm_tp, s, gr, gz, gn, mask_rows,
                         h0, Whh, bhh, L, m_pad, y, rstd_y, h_last, save ? hpm : nullptr, save ? r : nullptr,
                         save ? z : nullptr, save ? n : nullptr, save ? hn : nullptr);
      return check_launch("harl_gru_fwd");
    }
    const size_t shm = split_image_bytes(3 * GH, GH) + ((size_t)2 * 3 * GH) * sizeof(float);
    allow_big_lds(k_gru_fwd<true>, shm);
    hipLaunchKernelGGL(k_gru_fwd<true>, dim3(grid), dim3(WG_THREADS), shm, s, xin, gr, gz, gn, mask_rows, h0, Wih, bih,
                       Whh, bhh, L, m_pad, y, rstd_y, hpm, r, z, n, hn, h_last, save);
  } else {
    const size_t shm = split_image_bytes(3 * GH, GH) + ((size_t)3 * GH * (GH + 1) + 2 * 3 * GH) * sizeof(float);
    allow_big_lds(k_gru_fwd<false>, shm);
    hipLaunchKernelGGL(k_gru_fwd<false>, dim3(grid), dim3(WG_THREADS), shm, s, xin, nullptr, nullptr, nullptr, mask_rows,
                       h0, Wih, bih, Whh, bhh, L, m_pad, y, rstd_y, hpm, r, z, n, hn, h_last, save);
  }
  return check_launch("harl_gru_fwd");
}

extern "C" int harl_gru_bwd(const float *dhout, const float *mask_rows, const float *Wih, const float *Whh,
                            const float *hpm, const float *r, const float *z, const float *n, const float *hn, int H,
                            int L, long m_pad, const float *xmlp, const uint32_t *mask_mlp, const float *rstd_mlp,
                            float *dr, float *dz, float *dn, float *dhn, float *dz_mlp, void *stream) {
  if (L <= 0 || m_pad <= 0) return 0;
  if (H != GH) { set_error("harl_gru_bwd: hidden width must be 64"); return -2; }
  const size_t shm = split_image_bytes(3 * GH, GH);
  const long groups = m_pad / SLAB;
  long wgs = (groups + WAVES_PER_WG - 1) / WAVES_PER_WG;
  const int grid = (int)(wgs < 256 ? wgs : 256);
  static const bool quad = [] { const char *e = getenv("HARL_GRU_QUAD"); return !(e && e[0] == '0'); }();
  if (quad && groups <= 256) {  // four waves per slab (k_gru_bwd_q): one chain per CU
    const size_t shm_q = shm + (size_t)2 * 4 * 8 * 64 * 16;
    allow_big_lds(k_gru_bwd_q, shm_q);
    hipLaunchKernelGGL(k_gru_bwd_q, dim3((unsigned)groups), dim3(WG_THREADS), shm_q, (hipStream_t)stream, dhout, mask_rows, Whh,
                       hpm, r, z, n, hn, L, m_pad, dr, dz, dn, dhn);
  } else {
    allow_big_lds(k_gru_bwd, shm);
    hipLaunchKernelGGL(k_gru_bwd, dim3(grid), dim3(WG_THREADS), shm, (hipStream_t)stream, dhout, mask_rows, Whh, hpm, r, z, n, hn,
                       L, m_pad, dr, dz, dn, dhn);
  }
  if (!dz_mlp) return check_launch("harl_gru_bwd");  // the input side is the caller's (harl_mlp_bwd_trunk's first stage)
  const long n_slabs = (long)L * groups;
  allow_big_lds(k_gru_dx, shm);
  hipLaunchKernelGGL(k_gru_dx, dim3(persistent_grid(n_slabs, 2)), dim3(WG_THREADS), shm, (hipStream_t)stream, Wih, dr, dz, dn,
                     xmlp, mask_mlp, rstd_mlp, dz_mlp, n_slabs);
  return check_launch("harl_gru_bwd");
}
