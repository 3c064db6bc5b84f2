// mlp.hip -- actor / critic MLP forward and backward on the gfx950 matrix pipes: the H x H GEMMs on
// v_mfma_f32_32x32x16_bf16 with an exact three-way fp32 operand split (split_mfma.h), the narrow ones (first layer,
// fused first-layer weight gradient) on v_mfma_f32_32x32x2_f32.
//
// Replaces MLPBase/MLPLayer forward + autograd backward of the reference
// (harl/models/base/mlp.py:7-70) for the HAPPO / V-critic update.
//
// Formulation (see common.h): every GEMM is computed transposed, Y^T[feature, sample] =
// W[feature, k] * X^T[k, sample], with a wave owning 32 samples (the MFMA N dimension).  The
// accumulator ("C") layout then has lane <-> sample and registers <-> features, which is *also*
// a valid B-operand layout of the next GEMM (the k order of a dot product is free), so
// activations chain from layer to layer with no transpose, LayerNorm statistics are an in-lane
// sum plus one exchange with lane^32, and the activation tensors stored to HBM between kernels
// are register images ("ATL"), read and written with full 1 KiB wave transactions.
// The only transposes are in the weight-gradient kernel (the reduction runs over samples, so
// samples must become the MFMA k index); they go through LDS.
//
// Weights live in LDS for the lifetime of a persistent workgroup (three bf16 images, 96 KiB for 128 x 128).
#include "common.h"
#include "split_mfma.h"
#include "mfma_transpose.h"
#include "dw_common.h"
#include "fwd_epilogue.h"
#include "../../include/harl_hip.h"
#include <stdlib.h>
#include <type_traits>

using namespace harl;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// forward-mode (tangent) epilogue: given the tangent of the pre-activation in acc and the PRIMAL x_hat / relu mask /
// rstd of this layer,  a_dot = mask ? z_dot : 0 ;  x_hat_dot = rstd (a_dot - mean_f(a_dot) - x_hat mean_f(a_dot x_hat))
// (the LayerNorm Jacobian is symmetric, so this is the backward formula with the mask applied first).
template <int HO>
__device__ __forceinline__ void ln_jac_store(f32x16 (&acc)[HO / 32], const float *__restrict__ xprimal,
                                             const uint32_t *__restrict__ mask_in, const float *__restrict__ rstd_in,
                                             int lane, long slab, float *__restrict__ xdot_out) {
  constexpr int NR = HO / 2, NW = (NR + 31) / 32;
  float xh[NR];
  atl_load<HO>(xprimal, slab, lane, xh);
  uint32_t bits[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) bits[w] = mask_in[(slab * NW + w) * WAVE + lane];
  const float rstd = rstd_in[slab * SLAB + (lane & 31)];
  float ad[NR];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int R = 0; R < NR; ++R) {
    ad[R] = mask_pop(acc[R >> 4][R & 15], bits[R >> 5]);
    s1 += ad[R];
    s2 += ad[R] * xh[R];
  }
  s1 = wave_sum32(s1);
  s2 = wave_sum32(s2);
  s1 *= (1.0f / HO);
  s2 *= (1.0f / HO);
#pragma unroll
  for (int R = 0; R < NR; ++R) ad[R] = rstd * (ad[R] - s1 - xh[R] * s2);
  atl_store<HO>(xdot_out, slab, lane, ad);
}

// =============================================================================================
// hidden layer forward:  xout = norm(relu(Wp * xin + bp))       (ATL(HI) -> ATL(HO))
// GEMM on the bf16 matrix pipe with the exact three-way operand split (split_mfma.h).  LDS: the three weight images
// (3 * HO * HI * 2 B = 96 KiB for 128 x 128 -> one workgroup per CU, one wave per SIMD with the whole register file) +
// bias.  Per wave-slab: the slab's activations arrive as 16 float4 per lane (issued one slab ahead), are split once
// (~5 VALU ops per value), and feed HO/32 * HI/16 * 6 MFMAs; HBM-bound (profiles/r01_mfma_bf16x3.txt).
// =============================================================================================
constexpr bool split_one_wg(int ho, int hi, size_t extra = 0) { return 2 * (split_image_bytes(ho, hi) + ho * 4 + extra) > 160 * 1024; }

// MODE 0: x_hat_out = norm(relu(Wp x_in + bp)) (+ mask, rstd).   MODE 1: raw  z = Wp x_in + bp  stored as an ATL image.
// MODE 2: z = xout (read) + Wp x_in, then the LayerNorm Jacobian with the PRIMAL x_hat / mask / rstd of this layer ->
// xout.  (1 then 2 = the forward-mode tangent of a hidden layer, z_dot = Wp x_in_dot + Wp_dot x_hat_in + bp_dot, as two
// split-bf16 GEMMs: both weight matrices do not fit the LDS at once.)
template <int HI, int HO, int MODE = 0>
__global__ __launch_bounds__(WG_THREADS, split_one_wg(HO, HI) ? 1 : 2) void k_fwd_hidden(
    const float *__restrict__ xin, const float *__restrict__ Wp, const float *__restrict__ bp, float *__restrict__ xout,
    uint32_t *__restrict__ mask_out, float *__restrict__ rstd_out, long n_slabs, const float *__restrict__ xprimal = nullptr,
    const uint32_t *__restrict__ mask_in = nullptr, const float *__restrict__ rstd_in = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = HO / 32, NJ = HI / 16, NR = HI / 2;
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  float *bl = reinterpret_cast<float *>(img + 3 * MT * NJ * 64);
  stage_split_matrix<HO, HI, false, WG_THREADS>(img, Wp);
  if (MODE != 2)
    for (int e = threadIdx.x; e < HO; e += WG_THREADS) bl[e] = bp[e];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5;
  const long slab0 = (long)blockIdx.x * WAVES_PER_WG + wave, slab_stride = (long)gridDim.x * WAVES_PER_WG;
  const u32x4 *wl = img + lane;
  float raw[NR];
  atl_load<HI>(xin, slab0 < n_slabs ? slab0 : 0, lane, raw);
  for (long slab = slab0; slab < n_slabs; slab += slab_stride) {
    u32x4 x1[NJ], x2[NJ], x3[NJ];
    split_acts<NR>(raw, x1, x2, x3);
    // the next slab's activations: a whole slab of MFMA time (> 6000 cycles) to land
    atl_load<HI>(xin, slab + slab_stride < n_slabs ? slab + slab_stride : slab, lane, raw);
    f32x16 acc[MT];
    if constexpr (MODE == 2) {
      float z0[HO / 2];
      atl_load<HO>(xout, slab, lane, z0);
#pragma unroll
      for (int R = 0; R < HO / 2; ++R) acc[R >> 4][R & 15] = z0[R];
    } else {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bl[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    }
    split_gemm<MT, NJ>(wl, x1, x2, x3, acc, [](int) {});
    if constexpr (MODE == 0) {
      relu_norm_store<HO>(acc, lane, slab, xout, mask_out, rstd_out);
    } else if constexpr (MODE == 1) {
      float z[HO / 2];
#pragma unroll
      for (int R = 0; R < HO / 2; ++R) z[R] = acc[R >> 4][R & 15];
      atl_store<HO>(xout, slab, lane, z);
    } else {
      ln_jac_store<HO>(acc, xprimal, mask_in, rstd_in, lane, slab, xout);
    }
  }
}

// =============================================================================================
// first layer forward: rows of X (optionally gathered by idx), optional feature LayerNorm on the
// raw input, then the same epilogue.  k is processed in chunks of 32 input features; within chunk
// c (Dc valid features, KS = ceil(Dc/2) MFMA steps) step j feeds feature 32c + h*KS + j from lane
// half h.  The transposed weights W'^T[k][o] stay resident in LDS when they fit (D <= 128 for
// H = 128), otherwise they are re-staged chunk by chunk (wide observations, e.g. Humanoid 393).
// =============================================================================================
template <int HO, bool TANGENT>
__global__ __launch_bounds__(WG_THREADS, 2) void k_fwd_input(const float *__restrict__ X, long ldx,
                                                             const int64_t *__restrict__ idx, long M, int D,
                                                             const float *__restrict__ Wp,
                                                             const float *__restrict__ bp, int use_ln0,
                                                             float *__restrict__ xout,
                                                             uint32_t *__restrict__ mask_out,
                                                             float *__restrict__ rstd_out, float *__restrict__ mu0_out,
                                                             float *__restrict__ rstd0_out, long n_slabs, int nch,
                                                             int resident, const float *__restrict__ xprimal,
                                                             const uint32_t *__restrict__ mask_in,
                                                             const float *__restrict__ rstd_in) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int krows = resident ? nch * 32 : 32;
  float *Wt = lds;               // [krows][HO]
  float *bl = lds + krows * HO;  // [HO]
  for (int e = threadIdx.x; e < HO; e += WG_THREADS) bl[e] = bp[e];
  if (resident) {
    for (int e = threadIdx.x; e < HO * krows; e += WG_THREADS) {
      int o = e / krows, k = e - o * krows;
      Wt[k * HO + o] = k < D ? Wp[(long)o * D + k] : 0.f;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const long stride = (long)gridDim.x * WAVES_PER_WG;
  const long iters = (n_slabs + stride - 1) / stride;
  for (long it = 0; it < iters; ++it) {
    const long slab = (it * gridDim.x + blockIdx.x) * WAVES_PER_WG + wave;
    const bool active = slab < n_slabs;
    long j = slab * SLAB + i;
    if (j > M - 1) j = M - 1;
    if (j < 0) j = 0;
    const long row = idx ? idx[j] : j;
    const float *xr = X + row * ldx;

    float mean = 0.f, rstd = 1.f;
    if (use_ln0) {
      float s = 0.f;
      for (int c = 0; c < nch; ++c) {
        const int Dc = min(32, D - 32 * c), KS = (Dc + 1) >> 1;
        for (int jj = 0; jj < KS; ++jj) {
          const int kl = h * KS + jj;
          if (kl < Dc) s += xr[32 * c + kl];
        }
      }
      s = wave_sum32(s);
      mean = s / (float)D;
      float vs = 0.f;
      for (int c = 0; c < nch; ++c) {
        const int Dc = min(32, D - 32 * c), KS = (Dc + 1) >> 1;
        for (int jj = 0; jj < KS; ++jj) {
          const int kl = h * KS + jj;
          if (kl < Dc) {
            float d = xr[32 * c + kl] - mean;
            vs += d * d;
          }
        }
      }
      vs = wave_sum32(vs);
      rstd = 1.0f / sqrtf(vs / (float)D + 1e-5f);
    }

    f32x16 acc[HO / 32];
#pragma unroll
    for (int t = 0; t < HO / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = bl[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];

    for (int c = 0; c < nch; ++c) {
      if (!resident) {
        __syncthreads();  // previous chunk fully consumed
        for (int e = threadIdx.x; e < HO * 32; e += WG_THREADS) {
          int o = e >> 5, kl = e & 31;
          int k = 32 * c + kl;
          Wt[kl * HO + o] = k < D ? Wp[(long)o * D + k] : 0.f;
        }
        __syncthreads();
      }
      const int Dc = min(32, D - 32 * c), KS = (Dc + 1) >> 1;
      float xv[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int kl = h * KS + jj;
        xv[jj] = (jj < KS && kl < Dc) ? (xr[32 * c + kl] - mean) * rstd : 0.f;
      }
      const float *wt_lane = Wt + ((resident ? 32 * c : 0) + h * KS) * HO + i;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        if (jj < KS) {
#pragma unroll
          for (int t = 0; t < HO / 32; ++t) {
            const float a = wt_lane[jj * HO + 32 * t];
            acc[t] = MFMA(a, xv[jj], acc[t]);
          }
        }
      }
    }
    if (active) {
      if (TANGENT) {  // acc = W1'_dot x_hat0 + b1'_dot  ->  x_hat1_dot
        ln_jac_store<HO>(acc, xprimal, mask_in, rstd_in, lane, slab, xout);
      } else {
        relu_norm_store<HO>(acc, lane, slab, xout, mask_out, rstd_out);
        if (lane < 32) {
          mu0_out[slab * SLAB + lane] = mean;
          rstd0_out[slab * SLAB + lane] = rstd;
        }
      }
    }
  }
}

// normalised input rows of a wave-slab (parked in LDS as xr = my sample's row; columns >= D hold zeros from the
// kernel prologue) -> ATL(32*NCH) image in HBM: the B operand of the first layer's weight-gradient kernel (saves it the
// gather + re-normalisation of raw rows).  Pad columns are exact zeros (the same image k_x0n_wide writes).
template <int NCH>
__device__ __forceinline__ void x0n_store(const float *xr, int D, float mean, float rstd, int lane, long slab,
                                          float *__restrict__ out) {
  constexpr int HW = 32 * NCH;
  const float *xl = xr + 4 * (lane >> 5);
  float v[HW / 2];
#pragma unroll
  for (int R = 0; R < HW / 2; ++R)  // mean = 0, rstd = 1 without input LN; pad columns are exact zeros (as in k_x0n_wide)
    v[R] = feat_base(R) + 4 * (lane >> 5) < D ? (xl[feat_base(R)] - mean) * rstd : 0.f;
  // the LAST pad column (feature HW-1 = register HW/2-1 of the upper lane half) is a column of ones when D < HW: the
  // weight-gradient GEMM then yields db' = sum_s dz[s] in that column for free (used by the fused path of k_bwd_dx)
  if (D < HW && (lane >> 5) == 1) v[HW / 2 - 1] = 1.0f;
  atl_store<HW>(out, slab, lane, v);
}

// =============================================================================================
// first layer, narrow inputs (D <= 64: MPE 18/54, MAMuJoCo 17/23): the 32 rows of a wave-slab are fetched
// row-by-row with the lanes sweeping the row (each load instruction touches 1-2 cache lines instead of 64),
// parked in LDS as xs[row][LDX] (LDX odd: the "lane = sample" reads below are conflict-free), and the input
// LayerNorm statistics and the MFMA B operands are then taken from LDS.  Weights W'^T[k][o] resident in LDS.
// =============================================================================================
template <int HO, int RPI>
__global__ __launch_bounds__(WG_THREADS, 2) void k_fwd_input_staged(const float *__restrict__ X, long ldx,
                                                                    const int64_t *__restrict__ idx, long M, int D,
                                                                    const float *__restrict__ Wp,
                                                                    const float *__restrict__ bp, int use_ln0,
                                                                    float *__restrict__ xout,
                                                                    uint32_t *__restrict__ mask_out,
                                                                    float *__restrict__ rstd_out,
                                                                    float *__restrict__ mu0_out,
                                                                    float *__restrict__ rstd0_out,
                                                                    float *__restrict__ x0n_out, long n_slabs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NCH = RPI == 2 ? 1 : 2, krows = NCH * 32, LDX = krows + 1, NPF = SLAB / RPI;
  float *xs = lds;                              // [4 waves][32 rows][LDX]  (compile-time odd stride)
  float *bl = xs + WAVES_PER_WG * SLAB * LDX;   // [HO]
  float *Wt = bl + HO;                          // [krows][HO]
  for (int e = threadIdx.x; e < WAVES_PER_WG * SLAB * LDX; e += WG_THREADS) xs[e] = 0.f;  // pad columns stay zero
  for (int e = threadIdx.x; e < HO; e += WG_THREADS) bl[e] = bp[e];
  for (int e = threadIdx.x; e < HO * krows; e += WG_THREADS) {
    int o = e / krows, k = e - o * krows;
    Wt[k * HO + o] = k < D ? Wp[(long)o * D + k] : 0.f;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  float *xw = xs + wave * SLAB * LDX;  // this wave's 32 rows (private: no cross-wave hazards)
  const int lane_row = RPI == 2 ? h : 0, lane_k = RPI == 2 ? i : lane;
  const long slab0 = (long)blockIdx.x * WAVES_PER_WG + wave, slab_stride = (long)gridDim.x * WAVES_PER_WG;
  float pf[NPF];  // next slab's rows, in flight while this slab computes
  // Branch-free on purpose: with `idx ? idx[j] : j` and a predicated load inside the unrolled loop, hipcc emitted a
  // uniform branch + exec-masked load per row and closed each with s_waitcnt vmcnt(0) -- the 16-32 row loads of a slab
  // were fully serialised (rocprofv3: waves 40 % of their time in s_waitcnt, MFMA pipe 49 % busy in the fused kernel).
  const int lane_kc = lane_k < D ? lane_k : 0;
  auto prefetch_rows = [&](long slab) {
    int rows[NPF];
    if (idx) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        long j = slab * SLAB + u * RPI + lane_row;
        rows[u] = (int)idx[j < M ? j : M - 1];
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        long j = slab * SLAB + u * RPI + lane_row;
        rows[u] = (int)(j < M ? j : M - 1);
      }
    }
    // no select on the loaded value (lanes >= D fetch column 0 and simply never store it to LDS): a dependent VALU op
    // right behind each load made hipcc wait for every load individually (s_waitcnt vmcnt(0) x NPF per slab)
#pragma unroll
    for (int u = 0; u < NPF; ++u) pf[u] = X[(long)rows[u] * ldx + lane_kc];
  };
  if (slab0 < n_slabs) prefetch_rows(slab0);
  for (long slab = slab0; slab < n_slabs; slab += slab_stride) {
#pragma unroll
    for (int u = 0; u < NPF; ++u)
      if (lane_k < D) xw[(u * RPI + lane_row) * LDX + lane_k] = pf[u];
    // wave-private LDS hand-off between lanes: LDS executes a wave's accesses in order, so all that is needed is a
    // compiler barrier + lgkmcnt(0) (a workgroup-scope release fence would also drain vmcnt, i.e. the stores)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (slab + slab_stride < n_slabs) prefetch_rows(slab + slab_stride);
    const float *xr = xw + i * LDX;  // my sample's row

    float xv[NCH][16];
    int ks[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int Dc = min(32, D - 32 * c);
      ks[c] = Dc > 0 ? (Dc + 1) >> 1 : 0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int kl = h * ks[c] + jj;
        xv[c][jj] = (jj < ks[c] && kl < Dc) ? xr[32 * c + kl] : 0.f;
      }
    }
    float mean = 0.f, rstd = 1.f;
    if (use_ln0) {
      float sm = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) sm += xv[c][jj];
      sm = wave_sum32(sm);
      mean = sm / (float)D;
      float vs = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int Dc = min(32, D - 32 * c);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int kl = h * ks[c] + jj;
          const float d = xv[c][jj] - mean;
          vs += (jj < ks[c] && kl < Dc) ? d * d : 0.f;
        }
      }
      vs = wave_sum32(vs);
      rstd = 1.0f / sqrtf(vs / (float)D + 1e-5f);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int Dc = min(32, D - 32 * c);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int kl = h * ks[c] + jj;
          xv[c][jj] = (jj < ks[c] && kl < Dc) ? (xv[c][jj] - mean) * rstd : 0.f;
        }
      }
    }

    f32x16 acc[HO / 32];
#pragma unroll
    for (int t = 0; t < HO / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = bl[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int KS = ks[c];
      const float *wt_lane = Wt + (32 * c + h * KS) * HO + i;
      float a_cur[HO / 32], a_nxt[HO / 32];
#pragma unroll
      for (int t = 0; t < HO / 32; ++t) a_cur[t] = wt_lane[32 * t];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int jn = jj + 1 < KS ? jj + 1 : jj;
#pragma unroll
        for (int t = 0; t < HO / 32; ++t) a_nxt[t] = wt_lane[jn * HO + 32 * t];
        if (jj < KS) {
#pragma unroll
          for (int t = 0; t < HO / 32; ++t) acc[t] = MFMA(a_cur[t], xv[c][jj], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < HO / 32; ++t) a_cur[t] = a_nxt[t];
      }
    }
    relu_norm_store<HO>(acc, lane, slab, xout, mask_out, rstd_out);
    if (lane < 32) {
      mu0_out[slab * SLAB + lane] = mean;
      rstd0_out[slab * SLAB + lane] = rstd;
    }
    if (x0n_out) x0n_store<NCH>(xr, D, mean, rstd, lane, slab, x0n_out);
    __builtin_amdgcn_wave_barrier();  // all lanes done reading xw before the next slab overwrites it
  }
}

// =============================================================================================
// Fused two-layer forward for narrow inputs (D <= 64) and a 128 -> 128 (or 64 -> 64) second layer -- the bench
// configuration.  x_hat_1 never leaves the registers between the layers: the accumulator image of layer 1 IS the B
// operand of layer 2 (common.h), so the second GEMM runs straight out of the register file; x_hat_1 is written to
// HBM only when a backward pass will need it (store1).  512-thread workgroups (8 waves) share ONE LDS copy of both
// weight matrices (W1'^T fp32 + the three bf16 images of W2' + per-wave row staging = 147 KiB), 1 workgroup per CU =
// 2 waves per SIMD.  Layer 1 (K <= 32) stays on the fp32 MFMA; layer 2 is split_gemm() fed from registers.
// =============================================================================================
constexpr int FUSED_WAVES = 8;

template <int H, int RPI, int NCH>
__global__ __launch_bounds__(64 * FUSED_WAVES, 2) void k_fwd_fused2(
    const float *__restrict__ X, long ldx, const int64_t *__restrict__ idx, long M, int D, const float *__restrict__ W1p,
    const float *__restrict__ b1p, int use_ln0, const float *__restrict__ W2p, const float *__restrict__ b2p, int store1,
    float *__restrict__ x1out, uint32_t *__restrict__ mask1, float *__restrict__ rstd1, float *__restrict__ mu0_out,
    float *__restrict__ rstd0_out, float *__restrict__ x2out, uint32_t *__restrict__ mask2, float *__restrict__ rstd2,
    float *__restrict__ x0n_out, long n_slabs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NTHR = 64 * FUSED_WAVES;
  constexpr int NT_ = H / 32, NJ = H / 16, NPF = SLAB / RPI;
  // compile-time row stride (odd) so every LDS address below is ONE per-lane base + an immediate offset; with a
  // run-time stride hipcc keeps ~30 loop-invariant address VGPRs alive and spills them (measured: 46 % s_waitcnt)
  constexpr int krows = NCH * 32, LDX = NCH * 32 + 1;
  float *xs = lds;                              // [8 waves][32 rows][LDX]
  float *b1l = xs + FUSED_WAVES * SLAB * LDX;   // [H]
  float *b2l = b1l + H;                         // [H]
  float *Wt = b2l + H;                          // [krows][H]   W1'^T
  u32x4 *w2img = reinterpret_cast<u32x4 *>(Wt + krows * H);  // three bf16 images of W2' (split_mfma.h)
  for (int e = threadIdx.x; e < FUSED_WAVES * SLAB * LDX; e += NTHR) xs[e] = 0.f;  // pad columns stay zero
  stage_split_matrix<H, H, false, NTHR>(w2img, W2p);
  for (int e = threadIdx.x; e < H; e += NTHR) {
    b2l[e] = b2p[e];
    b1l[e] = b1p[e];
  }
  for (int e = threadIdx.x; e < H * krows; e += NTHR) {
    int o = e / krows, k = e - o * krows;
    Wt[k * H + o] = k < D ? W1p[(long)o * D + k] : 0.f;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  float *xw = xs + wave * SLAB * LDX;
  const int lane_row = RPI == 2 ? h : 0, lane_k = RPI == 2 ? i : lane;
  const u32x4 *wl = w2img + lane;
  const long slab0 = (long)blockIdx.x * FUSED_WAVES + wave, slab_stride = (long)gridDim.x * FUSED_WAVES;

  float pf[NPF];  // next slab's rows, in flight while this slab computes
  // Branch-free on purpose: with `idx ? idx[j] : j` and a predicated load inside the unrolled loop, hipcc emitted a
  // uniform branch + exec-masked load per row and closed each with s_waitcnt vmcnt(0) -- the 16-32 row loads of a slab
  // were fully serialised (rocprofv3: waves 40 % of their time in s_waitcnt, MFMA pipe 49 % busy in the fused kernel).
  const int lane_kc = lane_k < D ? lane_k : 0;
  auto prefetch_rows = [&](long slab) {
    int rows[NPF];
    if (idx) {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        long j = slab * SLAB + u * RPI + lane_row;
        rows[u] = (int)idx[j < M ? j : M - 1];
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPF; ++u) {
        long j = slab * SLAB + u * RPI + lane_row;
        rows[u] = (int)(j < M ? j : M - 1);
      }
    }
    // no select on the loaded value (lanes >= D fetch column 0 and simply never store it to LDS): a dependent VALU op
    // right behind each load made hipcc wait for every load individually (s_waitcnt vmcnt(0) x NPF per slab)
#pragma unroll
    for (int u = 0; u < NPF; ++u) pf[u] = X[(long)rows[u] * ldx + lane_kc];
  };
  if (slab0 < n_slabs) prefetch_rows(slab0);

  for (long slab = slab0; slab < n_slabs; slab += slab_stride) {
    // ---- rows -> LDS (private to the wave), then prefetch the next slab's rows
#pragma unroll
    for (int u = 0; u < NPF; ++u)
      if (lane_k < D) xw[(u * RPI + lane_row) * LDX + lane_k] = pf[u];
    // wave-private LDS hand-off between lanes: LDS executes a wave's accesses in order, so all that is needed is a
    // compiler barrier + lgkmcnt(0).  (A workgroup-scope release fence would also wait for vmcnt(0), i.e. for the
    // previous slab's 16 KiB of activation stores -- ~2 us per slab.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const float *xr = xw + i * LDX;

    // ---- my sample's features (<= 16 per 32-feature chunk and lane half) -> registers in ONE sweep of independent
    // LDS reads; input-LayerNorm statistics (two-pass, from registers) and normalisation in place
    float xv[NCH][16];
    int ks[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int Dc = min(32, D - 32 * c);
      ks[c] = Dc > 0 ? (Dc + 1) >> 1 : 0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int kl = h * ks[c] + jj;
        xv[c][jj] = (jj < ks[c] && kl < Dc) ? xr[32 * c + kl] : 0.f;
      }
    }
    float mean = 0.f, rstd0 = 1.f;
    if (use_ln0) {
      float sm = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) sm += xv[c][jj];  // invalid slots hold 0
      sm = wave_sum32(sm);
      mean = sm / (float)D;
      float vs = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int Dc = min(32, D - 32 * c);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int kl = h * ks[c] + jj;
          const float d = xv[c][jj] - mean;
          vs += (jj < ks[c] && kl < Dc) ? d * d : 0.f;
        }
      }
      vs = wave_sum32(vs);
      rstd0 = 1.0f / sqrtf(vs / (float)D + 1e-5f);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int Dc = min(32, D - 32 * c);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int kl = h * ks[c] + jj;
          xv[c][jj] = (jj < ks[c] && kl < Dc) ? (xv[c][jj] - mean) * rstd0 : 0.f;
        }
      }
    }

    // ---- layer 1: weight fragments of step jj+1 are read (unconditionally, from a clamped row) while the MFMAs of
    // step jj run; steps beyond KS are skipped with a wave-uniform branch
    float x1[H / 2];
    uint32_t bits1[(H / 2 + 31) / 32];
    float r1;
    {
      f32x16 acc[NT_];
#pragma unroll
      for (int t = 0; t < NT_; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b1l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int KS = ks[c];
        const float *wt_lane = Wt + (32 * c + h * KS) * H + i;
        float a_cur[NT_], a_nxt[NT_];
#pragma unroll
        for (int t = 0; t < NT_; ++t) a_cur[t] = wt_lane[32 * t];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int jn = jj + 1 < KS ? jj + 1 : jj;  // clamped: always a valid LDS row
#pragma unroll
          for (int t = 0; t < NT_; ++t) a_nxt[t] = wt_lane[jn * H + 32 * t];
          if (jj < KS) {
#pragma unroll
            for (int t = 0; t < NT_; ++t) acc[t] = MFMA(a_cur[t], xv[c][jj], acc[t]);
          }
#pragma unroll
          for (int t = 0; t < NT_; ++t) a_cur[t] = a_nxt[t];
        }
      }
      relu_norm_regs<H>(acc, x1, bits1, r1);
    }
    if (store1) {
      act_store<H>(x1, bits1, r1, lane, slab, x1out, mask1, rstd1);
      if (lane < 32) {
        mu0_out[slab * SLAB + lane] = mean;
        rstd0_out[slab * SLAB + lane] = rstd0;
      }
      if (x0n_out) x0n_store<NCH>(xr, D, mean, rstd0, lane, slab, x0n_out);
    }
    __builtin_amdgcn_wave_barrier();  // all lanes done with xw before the next slab's rows overwrite it
    // next slab's rows: issued BEFORE the layer-2 GEMM (16 more live VGPRs) so that ~16k MFMA cycles cover the HBM
    // latency.  Issued after it, only the ~1.5k-cycle epilogue did, and rocprofv3 showed the waves 40 % of their time in
    // s_waitcnt with the MFMA pipe 49 % busy.
    if (slab + slab_stride < n_slabs) prefetch_rows(slab + slab_stride);
    __builtin_amdgcn_sched_barrier(0);

    // ---- layer 2 straight out of the register file, on the bf16 pipe: x_hat_1 is split once (exactly) into three
    // bf16 operands per k-step, the three weight images sit in LDS
    u32x4 y1[NJ], y2[NJ], y3[NJ];
    split_acts<H / 2>(x1, y1, y2, y3);
    f32x16 acc[NT_];
#pragma unroll
    for (int t = 0; t < NT_; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = b2l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    split_gemm<NT_, NJ>(wl, y1, y2, y3, acc, [](int) {});
    relu_norm_store<H>(acc, lane, slab, x2out, mask2, rstd2);
  }
}

// =============================================================================================
// backward through Linear(HI->HO) and the relu+norm in front of it:
//   dx_hat = Wp^T dz ;  da = rstd (dx_hat - mean_f(dx_hat) - x_hat mean_f(dx_hat x_hat)) ;  dz_prev = mask ? da : 0
// A operand = Wp^T: lane i -> input feature 32t+i, step R -> output feature f(R,h); Wp row-major
// in LDS is read with consecutive addresses by the 32 lanes of a half (conflict-free).
// =============================================================================================
// KT > 0: FIRST-layer variant.  dz_prev (= dz_1) is only ever consumed by the first layer's weight gradient
// dW_1' = dz_1^T x0n, so that GEMM is done right here -- dz_1 (registers, lane = sample) and the normalised inputs x0n
// (ATL(32*KT)) are transposed on the matrix pipe (mfma_transpose.h) and multiplied with 78 (KT = 1) / 132 (KT = 2) bf16
// MFMAs per slab; dz_1 is never written to HBM and the separate harl_mlp_dw_partials pass over it disappears.  One
// workgroup per CU, per-workgroup partials in the layout of harl_mlp_dw_partials.  (Round 1 staged both operands through a
// wave-private LDS transpose and used the fp32 MFMA with one LDS read per MFMA: the same speed, measured.)

// LEAN (round 5): the register diet that lets this kernel share a CU with a weight-gradient workgroup launched on a second
// stream (nets.backward_trunk, HARL_BWD_STREAMS): <= 344 of the 512 registers per SIMD lane and nothing in LDS but the weight
// images, so that one k_dw_tr workgroup (168 registers, 48 KiB) fits next to it and the two kernels' matrix / vector / memory
// phases overlap in hardware -- both read the same dz and x_hat_prev at about the same time.  What goes: the one-slab-ahead
// prefetch of dz and the early request of the LayerNorm operands (their latency is the co-resident wave's to fill).
template <int HO, int HI, int KT = 0, bool LEAN = false>
__global__ __launch_bounds__(WG_THREADS, (KT > 0 || split_one_wg(HI, HO)) ? 1 : 2) void k_bwd_dx(
    const float *__restrict__ dz, const float *__restrict__ xprev, const uint32_t *__restrict__ mask_prev,
    const float *__restrict__ rstd_prev, const float *__restrict__ Wp, float *__restrict__ dz_prev, long n_slabs,
    const float *__restrict__ x0n = nullptr, float *__restrict__ dw_part = nullptr, int n_part_rows = 0) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  PHASE_BEGIN();
  // dx_hat = Wp^T dz on the bf16 pipe (split_mfma.h): GEMM rows = input features, k = output features
  constexpr int MT = HI / 32, NJ = HO / 16, NRO = HO / 2;
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  constexpr int KPF = 32 * (KT > 0 ? KT : 1);
  stage_split_matrix<HO, HI, true, WG_THREADS>(img, Wp);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const long slab0 = (long)blockIdx.x * WAVES_PER_WG + wave, slab_stride = (long)gridDim.x * WAVES_PER_WG;
  const u32x4 *wl = img + lane;
  float raw[NRO];
  if constexpr (!LEAN) atl_load<HO>(dz, slab0 < n_slabs ? slab0 : 0, lane, raw);
  f32x16 acc1[KT > 0 ? HI / 32 : 1][KT > 0 ? KT : 1];  // fused first-layer weight gradient, persistent over the slabs
  float dbs[KT > 0 ? HI / 32 : 1];                     // ... and its bias gradient (per-lane sums over the lane's samples)
  const Ident ident = make_ident(lane);
#pragma unroll
  for (int a = 0; a < (KT > 0 ? HI / 32 : 1); ++a) dbs[a] = 0.f;
  if constexpr (KT > 0) {
#pragma unroll
    for (int mt = 0; mt < HI / 32; ++mt)
#pragma unroll
      for (int n = 0; n < KT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[mt][n][r] = 0.f;
  }
  PHASE(10);
  for (long slab = slab0; slab < n_slabs; slab += slab_stride) {
    u32x4 g1[NJ], g2[NJ], g3[NJ];
    if constexpr (LEAN) atl_load<HO>(dz, slab, lane, raw);
    split_acts<NRO>(raw, g1, g2, g3);
    PHASE(0);
    if constexpr (!LEAN) atl_load<HO>(dz, slab + slab_stride < n_slabs ? slab + slab_stride : slab, lane, raw);  // one slab ahead
    // operands of the LayerNorm backward: issued now, consumed after the MFMA loop (latency fully hidden); LEAN: behind the
    // GEMM (64 registers less across it)
    float xh[HI / 2];
    float rstd;
    constexpr int NWP = (HI / 2 + 31) / 32;
    uint32_t mbits[NWP];
    f32x4 x0r[KT > 0 ? KPF / 8 : 1];
    auto ln_operands = [&]() {
      atl_load<HI>(xprev, slab, lane, xh);
      rstd = rstd_prev[slab * SLAB + i];
#pragma unroll
      for (int w = 0; w < NWP; ++w) mbits[w] = mask_prev[(slab * NWP + w) * WAVE + lane];
      if constexpr (KT > 0) {
        const f32x4 *bp = reinterpret_cast<const f32x4 *>(x0n + slab * (long)(KPF * SLAB)) + lane;
#pragma unroll
        for (int q = 0; q < KPF / 8; ++q) x0r[q] = bp[q * WAVE];
      }
    };
    if constexpr (!LEAN) ln_operands();
    f32x16 acc[HI / 32];
#pragma unroll
    for (int t = 0; t < HI / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    PHASE(1);
    split_gemm<MT, NJ>(wl, g1, g2, g3, acc, [](int) {});
    if constexpr (LEAN) ln_operands();
    PHASE(2);
    float dx[HI / 2];
#pragma unroll
    for (int R = 0; R < HI / 2; ++R) dx[R] = acc[R >> 4][R & 15];
    if constexpr (KT == 0) {
      float out[HI / 2];
      ln_bwd_relu_mbits<HI>(dx, xh, mbits, rstd, out);
      atl_store<HI>(dz_prev, slab, lane, out);
      PHASE(3);
    } else {
      float out[HI / 2];
      ln_bwd_relu_mbits<HI>(dx, xh, mbits, rstd, out);
      if (dz_prev) atl_store<HI>(dz_prev, slab, lane, out);
      PHASE(3);
      // dW_1'[f][k] += sum_s dz_1[s][f] x0n[s][k] with both operands transposed on the matrix pipe (mfma_transpose.h): no LDS
      // round trip, bf16 MFMAs instead of the fp32 pipe with an LDS read per MFMA (same speed, measured; no staging area)
      float xr0[KPF / 2];
#pragma unroll
      for (int q = 0; q < KPF / 8; ++q) {
        xr0[4 * q + 0] = x0r[q][0];
        xr0[4 * q + 1] = x0r[q][1];
        xr0[4 * q + 2] = x0r[q][2];
        xr0[4 * q + 3] = x0r[q][3];
      }
      u32x4 a1[KPF / 16], a2[KPF / 16], a3[KPF / 16];
      split_acts<KPF / 2>(xr0, a1, a2, a3);
      u32x4 Bt[KT][3][2];
#pragma unroll
      for (int n = 0; n < KT; ++n)
        transpose_block<false>(a1[2 * n], a1[2 * n + 1], a2[2 * n], a2[2 * n + 1], a3[2 * n], a3[2 * n + 1], ident, Bt[n]);
#pragma unroll
      for (int a = 0; a < HI / 32; ++a) {
        u32x4 At[3][2];
        dbs[a] += split_transpose_block<true>(&out[16 * a], ident, At);
#pragma unroll
        for (int n = 0; n < KT; ++n) dw_tile(acc1[a][n], At, Bt[n]);
      }
      PHASE(4);
    }
  }
  if constexpr (KT > 0) {
    // the four waves' accumulators combined in fixed order -> ONE partial row dWp[HI][KPF] | dbp[HI]; the partial arena has
    // n_part_rows rows per layer (shared with the other gradient kernels) and this kernel runs one workgroup per CU, so it
    // clears the rest (mfma_transpose.h)
    finish_partials<HI / 32, KT>(acc1, dbs, reinterpret_cast<float *>(img), dw_part, n_part_rows);
  }
  PHASE(11);
  PHASE_END(KT > 0 ? 0 : 1);
}

// =============================================================================================
// weight-gradient partials  dWp[o][k] = sum_s dz[s][o] * x_hat[s][k],  dbp[o] = sum_s dz[s][o].
// The reduction index (samples) must be the MFMA k index, i.e. both operands are needed with
// lane <-> feature: each iteration stages 64 samples of both operands into LDS as [sample][feature]
// (ds_write_b128 from the ATL register image, row stride 32*tiles+4 floats: conflict-free for the
// 8-lane write groups and for the 32-lane ds_read_b32 fragment reads), then the 4 waves split the
// 32x32 output tiles and run 32 MFMA steps per tile.  Accumulators persist across the workgroup's
// iterations; the per-workgroup partial is written once and reduced in fixed order afterwards.
//   A_KIND 0: dz in ATL (HO = 32*MT)      1: row-major [M_pad][DHEAD_LD] head gradients (MT = 1)
//   B_KIND 0: x_hat in ATL (K = 32*NT)    1: raw X rows (gather + input-LayerNorm on the fly),
//                                            blockIdx.y selects a group of NT 32-wide k tiles
// =============================================================================================
constexpr int DW_S = 64;  // samples per staging round


template <int A_KIND, int B_KIND, int MT, int NT>
__global__ __launch_bounds__(WG_THREADS, 2) void k_dw(const float *__restrict__ a_src, const float *__restrict__ b_src,
                                                      long ldx, const int64_t *__restrict__ idx,
                                                      const float *__restrict__ mu0, const float *__restrict__ rstd0,
                                                      int K, long M, long n_slabs, float *__restrict__ part, int KP) {
  using SP = DwSplit<MT, NT>;
  constexpr int LDA = 32 * MT + 4, LDB = 32 * NT + 4;
  constexpr bool A_ATL = A_KIND == 0, B_ATL = B_KIND == 0;
  constexpr int HA = 32 * MT, HB = 32 * NT;
  // Staging roles.  Both operands ATL: waves 0,1 fetch the two A tiles, waves 2,3 the two B tiles.  Only one ATL
  // operand (first layer: B = raw rows; head: A = [M][32] rows): its two tiles are split in q-halves over all four
  // waves and the light operand is fetched element-wise by all 256 threads.
  constexpr int A_SPLIT = (A_ATL && !B_ATL) ? 2 : 1, B_SPLIT = (B_ATL && !A_ATL) ? 2 : 1;
  constexpr int A_NQ = A_ATL ? HA / 8 / A_SPLIT : 0, B_NQ = B_ATL ? HB / 8 / B_SPLIT : 0;
  constexpr int NQ = A_NQ > B_NQ ? A_NQ : B_NQ;       // float4 prefetch registers per lane for the ATL piece
  constexpr int BRAW = B_ATL ? 0 : (DW_S * HB) / WG_THREADS;  // raw-B elements per thread (= 8 NT)
  constexpr int BCOLS_LOG = NT == 1 ? 5 : (NT == 2 ? 6 : 7);

  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *As = lds;               // [DW_S][LDA]
  float *Bs = lds + DW_S * LDA;  // [DW_S][LDB]
  float *mul = Bs + DW_S * LDB;  // [DW_S] input-LayerNorm mean  (raw-B only)
  float *rsl = mul + DW_S;       // [DW_S] input-LayerNorm rstd
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave % SP::WM, wn = wave / SP::WM;
  const int k0 = blockIdx.y * (32 * NT);  // first input feature handled by this workgroup (raw B)
  const bool norm_b = !B_ATL && mu0 != nullptr;

  // which ATL piece this wave fetches
  bool piece_is_a;
  int piece_q0, piece_nq;
  const int sl = wave & 1;
  if (A_ATL && B_ATL) {
    piece_is_a = wave < 2;
    piece_q0 = 0;
    piece_nq = piece_is_a ? A_NQ : B_NQ;
  } else if (A_ATL) {
    piece_is_a = true;
    piece_q0 = (wave >> 1) * A_NQ;
    piece_nq = A_NQ;
  } else {
    piece_is_a = false;
    piece_q0 = (wave >> 1) * B_NQ;
    piece_nq = B_NQ;
  }
  const float *piece_src = piece_is_a ? a_src : b_src;
  const int piece_h = piece_is_a ? HA : HB;

  f32x4 pr[NQ > 0 ? NQ : 1];
  float xb[BRAW > 0 ? BRAW : 1];
  f32x4 ha[2];
  float pmu = 0.f, prs = 1.f;

  auto prefetch = [&](long it) {
    const long slab = 2 * it + sl;
    const bool ok = slab < n_slabs;
    if (NQ > 0) {
      const f32x4 *p = reinterpret_cast<const f32x4 *>(piece_src + slab * (long)(piece_h * SLAB)) + lane;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        pr[q] = (ok && q < piece_nq) ? p[(piece_q0 + q) * WAVE] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!A_ATL) {  // head gradients [M_pad][32]: 64 rows x 8 float4, two per thread
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f = threadIdx.x + WG_THREADS * u;
        const int row = f >> 3, c4 = f & 7;
        const long slab_r = 2 * it + (row >> 5);
        ha[u] = slab_r < n_slabs ? *reinterpret_cast<const f32x4 *>(a_src + (2 * it * SLAB + row) * (long)DHEAD_LD + 4 * c4)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (!B_ATL) {  // raw input rows (gathered): thread -> fixed column, 8*NT/... samples
#pragma unroll
      for (int u = 0; u < BRAW; ++u) {
        const int e = threadIdx.x + WG_THREADS * u;
        const int s = e >> BCOLS_LOG, kk = e & (HB - 1);
        long j = 2 * it * SLAB + s;
        const bool okr = j < M;
        if (j > M - 1) j = M - 1;
        const long row = idx ? idx[j] : j;
        const int k = k0 + kk;
        xb[u] = (okr && k < K) ? b_src[row * ldx + k] : 0.f;
      }
      if (norm_b && threadIdx.x < DW_S) {
        long j = 2 * it * SLAB + threadIdx.x;
        if (j > M - 1) j = M - 1;
        pmu = mu0[j];   // NB: mu0/rstd0 are indexed by minibatch position (as written by the forward pass)
        prs = rstd0[j];
      }
    }
  };

  f32x16 acc[SP::TM][SP::TN];
  float dbsum[SP::TM];
#pragma unroll
  for (int a = 0; a < SP::TM; ++a) {
    dbsum[a] = 0.f;
#pragma unroll
    for (int b = 0; b < SP::TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }

  const long n_iter = (n_slabs + 1) / 2;
  if ((long)blockIdx.x < n_iter) prefetch(blockIdx.x);
  for (long it = blockIdx.x; it < n_iter; it += gridDim.x) {
    __syncthreads();  // previous round's fragments fully read
    // ---- registers -> LDS as [sample][feature]
    if (NQ > 0) {
      float *dst = (piece_is_a ? As : Bs) + (sl * SLAB + i) * (piece_is_a ? LDA : LDB) + 4 * h;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (q < piece_nq) {
          const int qq = piece_q0 + q;
          *reinterpret_cast<f32x4 *>(dst + 32 * (qq >> 2) + 8 * (qq & 3)) = pr[q];
        }
      }
    }
    if (!A_ATL) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f = threadIdx.x + WG_THREADS * u;
        *reinterpret_cast<f32x4 *>(&As[(f >> 3) * LDA + 4 * (f & 7)]) = ha[u];
      }
    }
    if (!B_ATL) {
#pragma unroll
      for (int u = 0; u < BRAW; ++u) {
        const int e = threadIdx.x + WG_THREADS * u;
        Bs[(e >> BCOLS_LOG) * LDB + (e & (HB - 1))] = xb[u];
      }
      if (norm_b && threadIdx.x < DW_S) {
        mul[threadIdx.x] = pmu;
        rsl[threadIdx.x] = prs;
      }
    }
    __syncthreads();
    if (it + gridDim.x < n_iter) prefetch(it + gridDim.x);  // next round's loads fly during the MFMA phase
    __builtin_amdgcn_sched_barrier(0);  // pin the issue point: hipcc otherwise sinks the loads to their first use
    // ---- MFMA over the 64 staged samples (32 steps of 2)
#pragma unroll 4
    for (int kk = 0; kk < DW_S / 2; ++kk) {
      const int srow = 2 * kk + h;
      float av[SP::TM], bv[SP::TN];
#pragma unroll
      for (int a = 0; a < SP::TM; ++a) {
        av[a] = As[srow * LDA + 32 * (wm * SP::TM + a) + i];
        if (wn == 0) dbsum[a] += av[a];
      }
      float m_ = 0.f, r_ = 1.f;
      if (norm_b) {
        m_ = mul[srow];
        r_ = rsl[srow];
      }
#pragma unroll
      for (int b = 0; b < SP::TN; ++b) {
        const int nt = wn * SP::TN + b;
        float v = nt < NT ? Bs[srow * LDB + 32 * nt + i] : 0.f;
        // input LayerNorm applied at fragment-read time; zero-padded columns (k >= K) become -mu*rstd, which only
        // lands in dWp columns >= K that nobody reads
        bv[b] = norm_b ? (v - m_) * r_ : v;
      }
#pragma unroll
      for (int a = 0; a < SP::TM; ++a)
#pragma unroll
        for (int b = 0; b < SP::TN; ++b) acc[a][b] = MFMA(av[a], bv[b], acc[a][b]);
    }
  }

  // ---- write this workgroup's partial: dWp[32*MT][KP] then dbp[32*MT]
  float *mypart = part + (long)blockIdx.x * ((long)32 * MT * KP + 32 * MT);
#pragma unroll
  for (int a = 0; a < SP::TM; ++a) {
    const int mt = wm * SP::TM + a;
#pragma unroll
    for (int b = 0; b < SP::TN; ++b) {
      const int nt = wn * SP::TN + b;
      if (nt < NT) {
        const int kcol = k0 + 32 * nt + i;
        if (kcol < KP) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
            mypart[(long)o * KP + kcol] = acc[a][b][r];
          }
        }
      }
    }
    if (wn == 0 && blockIdx.y == 0) {
      float t = wave_sum32(dbsum[a]);
      if (h == 0) mypart[(long)32 * MT * KP + 32 * mt + i] = t;
    }
  }
}

// =============================================================================================
// k_dw_tr: weight-gradient partials of the hidden layers on the bf16 pipe (both operands ATL; widths 64 / 128 / 256 and the
// wide first layers in column groups).  The reduction index is the sample, so both operands are needed as "lane = feature,
// 8 consecutive samples per lane" while every producer holds "lane = sample".  The transposition is done by the LDS itself
// (gfx950 ds_read_b64_tr_b16): per round (ONE 32-sample slab) the four waves split the slab's (HA + HB)/8 float4 pieces
// exactly into three bf16 terms (split_mfma.h) and store them the way a lane holds them -- four consecutive features of its
// sample as ONE ds_write_b64 per term -- into [4 samples][16 features] blocks; the transpose read then hands every lane of a
// 16-lane group its feature column of a block: two ds_read_b64_tr_b16 per fragment.  Image per operand and term:
// [8 sample quads][H/16 feature tiles][4][16] bf16 with 8 B of padding per sample quad (the 16 lanes of a store group hit 32
// distinct banks; the 32 lanes of a read cycle take 256 contiguous bytes).  Each wave owns TM x TN output tiles: 2 k-steps x
// the 6 cross products per tile and round.  db' is summed in fp32 by the waves that stage dz (lane = sample) and reduced
// across lanes once at the end.  LDS 48 KiB for 128 x 128 -> 2 workgroups per CU.
// Round 2 scattered every term with its own ds_write_b16 ([feature][32 samples] images, 384 two-byte stores per slab at 4
// cycles each on the LDS store path): s_memtime showed 2.7k cycles per round in the scatter and 3.2k in a 48-MFMA phase that
// needs 1.55k -- bound by the LDS store path (0.207 -> 0.189 ms at 819 200 x 128 x 128 with the 96 wide stores).
//
// Wide first layers (x0n ATL(KP), KP up to 512): launched once per group of NT <= 6 column tiles; `b_slab_floats` = KP * 32
// is the slab stride of the B image, `tile0` the group's first 32-column tile, KP the row stride of dWp in the partial;
// db' is written by the tile0 == 0 launch only.
// =============================================================================================
__device__ __forceinline__ u32x2_t tr_read(const unsigned char *p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3))) *)(const __attribute__((address_space(3))) unsigned char *)p);
  return __builtin_bit_cast(u32x2_t, v);
}

template <int MT, int NT>
__device__ __forceinline__ void dw_tr_body(const float *__restrict__ a_src, const float *__restrict__ b_src, long n_slabs,
                                           float *__restrict__ part, long b_slab_floats, int tile0, int KP) {
  using SP = DwSplit<MT, NT>;
  constexpr int HA = 32 * MT, HB = 32 * NT;
  constexpr int NPA = HA / 8, NPB = HB / 8, PER = (NPA + NPB) / WAVES_PER_WG;  // float4 pieces per lane and wave
  static_assert((NPA + NPB) % WAVES_PER_WG == 0, "pieces divide evenly over the waves");
  constexpr int SQA = (HA / 16) * 128 + 8, SQB = (HB / 16) * 128 + 8;  // bytes per sample quad (4 samples x H features + pad)
  constexpr int IMG_A = 8 * SQA, IMG_B = 8 * SQB;                      // bytes per term image (32 samples)
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  PHASE_BEGIN();
  unsigned char *Ab = ldsb;               // [3 terms][IMG_A]
  unsigned char *Bb = ldsb + 3 * IMG_A;   // [3 terms][IMG_B]
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const int wm = wave % SP::WM, wn = wave / SP::WM;

  f32x4 pr[PER];
  float dbacc[PER][4];
#pragma unroll
  for (int u = 0; u < PER; ++u)
#pragma unroll
    for (int c = 0; c < 4; ++c) dbacc[u][c] = 0.f;
  auto prefetch = [&](long slab) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int gu = wave * PER + u;
      const bool is_a = gu < NPA;
      const float *src = is_a ? a_src + slab * (long)(HA * SLAB) : b_src + slab * b_slab_floats;
      const int q = is_a ? gu : gu - NPA + 4 * tile0;
      pr[u] = (reinterpret_cast<const f32x4 *>(src) + lane)[q * WAVE];
    }
  };

  f32x16 acc[SP::TM][SP::TN];
#pragma unroll
  for (int a = 0; a < SP::TM; ++a)
#pragma unroll
    for (int b = 0; b < SP::TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment address of this lane inside a term image: feature tile 2 t + ((lane >> 4) & 1), sample quads 4 ks + 2 h (+1),
  // block row (lane & 15) >> 2, column segment 4 (lane & 3)
  const int p16 = lane & 15, g1 = (lane >> 4) & 1;
  const int frag_lane = g1 * 128 + (p16 >> 2) * 32 + (p16 & 3) * 8;

  if ((long)blockIdx.x < n_slabs) prefetch(blockIdx.x);
  PHASE(10);
  for (long slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
    __syncthreads();  // previous round's fragments fully read
    PHASE(0);
    // ---- split + store: lane (sample i, half h) of piece q holds features 32 (q>>2) + 8 (q&3) + 4 h + c, c = 0..3
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int gu = wave * PER + u;
      const bool is_a = gu < NPA;
      const int q = is_a ? gu : gu - NPA;
      const int sqb = is_a ? SQA : SQB, tstride = is_a ? IMG_A : IMG_B;
      unsigned char *d = (is_a ? Ab : Bb) + (i >> 2) * sqb + (2 * (q >> 2) + ((q & 3) >> 1)) * 128 + (i & 3) * 32 +
                         (8 * (q & 1) + 4 * h) * 2;
      if (is_a) {
#pragma unroll
        for (int c = 0; c < 4; ++c) dbacc[u][c] += pr[u][c];
      }
      unsigned t1a, t2a, t3a, t1b, t2b, t3b;
      split3<false>(pr[u][0], pr[u][1], t1a, t2a, t3a);
      split3<false>(pr[u][2], pr[u][3], t1b, t2b, t3b);
      *reinterpret_cast<u32x2_t *>(d) = u32x2_t{t1a, t1b};
      *reinterpret_cast<u32x2_t *>(d + tstride) = u32x2_t{t2a, t2b};
      *reinterpret_cast<u32x2_t *>(d + 2 * tstride) = u32x2_t{t3a, t3b};
    }
    PHASE(1);
    __syncthreads();
    PHASE(2);
    if (slab + gridDim.x < n_slabs) prefetch(slab + gridDim.x);  // next round's loads fly during the MFMA phase
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2 k-steps of 16 samples: lane (feature, g = h) takes samples 16 ks + 8 h .. + 7 of its feature column
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 av[3][SP::TM], bv[3][SP::TN];
#pragma unroll
      for (int term = 0; term < 3; ++term) {
#pragma unroll
        for (int a = 0; a < SP::TM; ++a) {
          const unsigned char *fp = Ab + term * IMG_A + (4 * ks + 2 * h) * SQA + 2 * (wm * SP::TM + a) * 128 + frag_lane;
          const u32x2_t lo = tr_read(fp), hi = tr_read(fp + SQA);
          av[term][a] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
#pragma unroll
        for (int b = 0; b < SP::TN; ++b) {
          const int nt = wn * SP::TN + b < NT ? wn * SP::TN + b : NT - 1;  // surplus tile slots (NT = 1, 3) recompute the last tile
          const unsigned char *fp = Bb + term * IMG_B + (4 * ks + 2 * h) * SQB + 2 * nt * 128 + frag_lane;
          const u32x2_t lo = tr_read(fp), hi = tr_read(fp + SQB);
          bv[term][b] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
#pragma unroll
      for (int a = 0; a < SP::TM; ++a)
#pragma unroll
        for (int b = 0; b < SP::TN; ++b) {
          acc[a][b] = mfma_bf16(av[2][a], bv[0][b], acc[a][b]);
          acc[a][b] = mfma_bf16(av[0][a], bv[2][b], acc[a][b]);
          acc[a][b] = mfma_bf16(av[1][a], bv[1][b], acc[a][b]);
          acc[a][b] = mfma_bf16(av[1][a], bv[0][b], acc[a][b]);
          acc[a][b] = mfma_bf16(av[0][a], bv[1][b], acc[a][b]);
          acc[a][b] = mfma_bf16(av[0][a], bv[0][b], acc[a][b]);
        }
    }
    PHASE(3);
  }

  // ---- write this workgroup's partial: dWp[HA][KP] then dbp[HA]
  float *mypart = part + (long)blockIdx.x * ((long)HA * KP + HA);
#pragma unroll
  for (int a = 0; a < SP::TM; ++a) {
    const int mt = wm * SP::TM + a;
#pragma unroll
    for (int b = 0; b < SP::TN; ++b) {
      if (wn * SP::TN + b < NT) {
        const int kcol = 32 * (tile0 + wn * SP::TN + b) + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
          mypart[(long)o * KP + kcol] = acc[a][b][r];
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int gu = wave * PER + u;
    if (gu < NPA && tile0 == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float t = half_reduce_sum(dbacc[u][c]);
        if (i == 0) mypart[(long)HA * KP + 32 * (gu >> 2) + 8 * (gu & 3) + 4 * h + c] = t;
      }
    }
  }
  PHASE(11);
  PHASE_END(2);
}

template <int MT, int NT>
__global__ __launch_bounds__(WG_THREADS, 2) void k_dw_tr(const float *__restrict__ a_src, const float *__restrict__ b_src,
                                                         long n_slabs, float *__restrict__ part, long b_slab_floats,
                                                         int tile0, int KP) {
  dw_tr_body<MT, NT>(a_src, b_src, n_slabs, part, b_slab_floats, tile0, KP);
}

// several independent weight-gradient problems of one shape in ONE launch (blockIdx.y = problem): the six gate blocks of a
// GRU (d gi_g^T x_hat, d gh_g^T h~) were six launches of ~17 us each per optimiser step at the SMAC sizes, most of it launch
// latency and ramp-up (270 of them per 8-agent update)
constexpr int DW_MULTI_MAX = 8;
struct DwMulti {
  const float *a[DW_MULTI_MAX];
  const float *b[DW_MULTI_MAX];
  float *part[DW_MULTI_MAX];
};
template <int MT, int NT>
__global__ __launch_bounds__(WG_THREADS, 2) void k_dw_tr_multi(DwMulti P, long n_slabs, int KP) {
  dw_tr_body<MT, NT>(P.a[blockIdx.y], P.b[blockIdx.y], n_slabs, P.part[blockIdx.y], (long)KP * SLAB, 0, KP);
}

// ... and of DIFFERENT shapes (round 6): every weight gradient of a 64-wide recurrent network -- the three MLP layers (the first
// one against the wide x0n image, in groups of <= 4 column tiles) and the six gate blocks -- was four launches per optimiser step;
// blockIdx.y picks the problem, its column-tile count picks the body (all four instantiations live in the one kernel, LDS and
// registers sized for the widest).  Each problem is computed exactly as its own k_dw_tr<MT, nt> launch would.
constexpr int DW_MULTIV_MAX = 12;
struct DwMultiV {
  const float *a[DW_MULTIV_MAX];
  const float *b[DW_MULTIV_MAX];
  float *part[DW_MULTIV_MAX];
  int K[DW_MULTIV_MAX];      // width of the B image (row stride of dWp in the partial)
  int tile0[DW_MULTIV_MAX];  // first 32-column tile of this group
  int nt[DW_MULTIV_MAX];     // column tiles of this group (1..4)
};
template <int MT>
__global__ __launch_bounds__(WG_THREADS, 2) void k_dw_tr_multi_v(DwMultiV P, long n_slabs) {
  const int y = blockIdx.y;
  const float *a = P.a[y], *b = P.b[y];
  float *part = P.part[y];
  const int K = P.K[y], t0 = P.tile0[y];
  switch (P.nt[y]) {  // (workgroup-uniform)
    case 1: dw_tr_body<MT, 1>(a, b, n_slabs, part, (long)K * SLAB, t0, K); break;
    case 2: dw_tr_body<MT, 2>(a, b, n_slabs, part, (long)K * SLAB, t0, K); break;
    case 3: dw_tr_body<MT, 3>(a, b, n_slabs, part, (long)K * SLAB, t0, K); break;
    default: dw_tr_body<MT, 4>(a, b, n_slabs, part, (long)K * SLAB, t0, K); break;
  }
}

// The six gate blocks of a 64-wide GRU as ONE problem (round 6): dW_ih_g = d gi_g^T x_hat (g = r, z, n), dW_hh_g = d gh_g^T h~ with
// d gi = [dr, dz, dn], d gh = [dr, dz, dhn].  As six launches-in-one they read x_hat and h~ three times each and dr, dz twice:
// 3 072 B per row; here every image is staged ONCE per slab -- A = [dr | dz | dn | dhn] (256 features), B = [x_hat | h~] (128) --
// 1 536 B per row (the launch is bound by its row traffic: 4.4 TB/s at 655 360 rows).  Eight waves: all of them split and store
// their sixth of the 48 float4 pieces (k_dw_tr's transposing image, same addresses as an ATL(256) / ATL(128) operand would
// get), six of them own one 64 x 64 gate block each -- 2 x 2 tiles, two k-steps, the six cross products in k_dw_tr's order --
// and write it as that problem's partial row.  Same slabs per workgroup, same order of every sum: bit-identical to six k_dw_tr<2, 2>.
struct DwGru6 {
  const float *a[4];  // dr, dz, dn, dhn   (ATL(64) images)
  const float *b[2];  // x_hat of the last MLP layer, h~
  float *part[6];     // W_ih r, z, n ; W_hh r, z, n
};
constexpr int G6_WAVES = 8;
__global__ __launch_bounds__(64 * G6_WAVES, 1) void k_dw_gru6(DwGru6 P, long n_slabs) {
  constexpr int HA = 256, HB = 128, NPA = HA / 8, NPB = HB / 8, PER = (NPA + NPB) / G6_WAVES;
  static_assert((NPA + NPB) % G6_WAVES == 0, "pieces divide evenly over the waves");
  constexpr int SQA = (HA / 16) * 128 + 8, SQB = (HB / 16) * 128 + 8, IMG_A = 8 * SQA, IMG_B = 8 * SQB;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
  unsigned char *Ab = ldsb, *Bb = ldsb + 3 * IMG_A;
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  f32x4 pr[PER];
  float dbacc[PER][4];
#pragma unroll
  for (int u = 0; u < PER; ++u)
#pragma unroll
    for (int c = 0; c < 4; ++c) dbacc[u][c] = 0.f;
  auto prefetch = [&](long slab) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int gu = wave * PER + u;
      const bool is_a = gu < NPA;
      const int q = is_a ? gu : gu - NPA;
      const float *src = (is_a ? P.a[q >> 3] : P.b[q >> 3]) + slab * (long)(64 * SLAB);
      pr[u] = (reinterpret_cast<const f32x4 *>(src) + lane)[(q & 7) * WAVE];
    }
  };
  // wave p < 6 owns problem p: A block / B image
  const int ablk = wave == 5 ? 3 : (wave < 3 ? wave : wave - 3), bimg = wave < 3 ? 0 : 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int p16 = lane & 15, g1 = (lane >> 4) & 1;
  const int frag_lane = g1 * 128 + (p16 >> 2) * 32 + (p16 & 3) * 8;
  if ((long)blockIdx.x < n_slabs) prefetch(blockIdx.x);
  for (long slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
    __syncthreads();  // previous round's fragments fully read
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int gu = wave * PER + u;
      const bool is_a = gu < NPA;
      const int q = is_a ? gu : gu - NPA;
      const int sqb = is_a ? SQA : SQB, tstride = is_a ? IMG_A : IMG_B;
      unsigned char *d = (is_a ? Ab : Bb) + (i >> 2) * sqb + (2 * (q >> 2) + ((q & 3) >> 1)) * 128 + (i & 3) * 32 +
                         (8 * (q & 1) + 4 * h) * 2;
      if (is_a) {
#pragma unroll
        for (int c = 0; c < 4; ++c) dbacc[u][c] += pr[u][c];
      }
      unsigned t1a, t2a, t3a, t1b, t2b, t3b;
      split3<false>(pr[u][0], pr[u][1], t1a, t2a, t3a);
      split3<false>(pr[u][2], pr[u][3], t1b, t2b, t3b);
      *reinterpret_cast<u32x2_t *>(d) = u32x2_t{t1a, t1b};
      *reinterpret_cast<u32x2_t *>(d + tstride) = u32x2_t{t2a, t2b};
      *reinterpret_cast<u32x2_t *>(d + 2 * tstride) = u32x2_t{t3a, t3b};
    }
    __syncthreads();
    if (slab + gridDim.x < n_slabs) prefetch(slab + gridDim.x);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < 6) {  // (wave-uniform)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 av[3][2], bv[3][2];
#pragma unroll
        for (int term = 0; term < 3; ++term) {
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const unsigned char *fp = Ab + term * IMG_A + (4 * ks + 2 * h) * SQA + 2 * (2 * ablk + a) * 128 + frag_lane;
            const u32x2_t lo = tr_read(fp), hi = tr_read(fp + SQA);
            av[term][a] = u32x4{lo[0], lo[1], hi[0], hi[1]};
          }
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const unsigned char *fp = Bb + term * IMG_B + (4 * ks + 2 * h) * SQB + 2 * (2 * bimg + b) * 128 + frag_lane;
            const u32x2_t lo = tr_read(fp), hi = tr_read(fp + SQB);
            bv[term][b] = u32x4{lo[0], lo[1], hi[0], hi[1]};
          }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            acc[a][b] = mfma_bf16(av[2][a], bv[0][b], acc[a][b]);
            acc[a][b] = mfma_bf16(av[0][a], bv[2][b], acc[a][b]);
            acc[a][b] = mfma_bf16(av[1][a], bv[1][b], acc[a][b]);
            acc[a][b] = mfma_bf16(av[1][a], bv[0][b], acc[a][b]);
            acc[a][b] = mfma_bf16(av[0][a], bv[1][b], acc[a][b]);
            acc[a][b] = mfma_bf16(av[0][a], bv[0][b], acc[a][b]);
          }
      }
    }
  }
  constexpr long ROW = 64 * 64 + 64;  // dWp[64][64] | dbp[64]
  if (wave < 6) {
    float *mypart = P.part[wave] + (long)blockIdx.x * ROW;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
          mypart[(long)o * 64 + 32 * b + i] = acc[a][b][r];
        }
  }
  // db' of a gate block goes to every problem that uses it: dr -> W_ih r, W_hh r ; dz -> W_ih z, W_hh z ; dn -> W_ih n ; dhn -> W_hh n
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int gu = wave * PER + u;
    if (gu < NPA) {
      const int blk = gu >> 3, lq = gu & 7;
      float *p0 = P.part[blk == 3 ? 5 : blk] + (long)blockIdx.x * ROW + 64 * 64;
      float *p1 = blk < 2 ? P.part[blk + 3] + (long)blockIdx.x * ROW + 64 * 64 : nullptr;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float t = half_reduce_sum(dbacc[u][c]);
        if (i == 0) {
          const int f = 32 * (lq >> 2) + 8 * (lq & 3) + 4 * h + c;
          p0[f] = t;
          if (p1) p1[f] = t;
        }
      }
    }
  }
}

// =============================================================================================
// k_bwd_dx_dw: the WHOLE backward of one hidden Linear(128 -> 128) and the relu + LayerNorm in front of it in ONE persistent
// launch (round 5): dz_prev = LNrelu'(Wp^T dz), dW' += dz^T x_hat_prev, db' += sum dz, and -- first-layer variant, KT = 1 --
// dW_1' += dz_1^T x0n.  It replaces the pair harl_mlp_dw_partials(dz, x_hat_prev) + harl_mlp_bwd_dx(...), which streamed the same
// two operands (dz 512 B + x_hat_prev 512 B per sample) from HBM one after the other: half of the MPE step's time and 17 of its
// 58 GB.  Matches autograd through MLPLayer (harl/models/base/mlp.py:25-38, happo.py:93-100).
//
// One workgroup per CU, four waves, 144 KiB of LDS: the three split images of Wp^T (96 KiB, as k_bwd_dx) and TWO transposition
// buffers for x_hat_prev^T (k_dw_tr's image, 24 KiB each).  A super-round = four slabs, one per wave:
//   O part (the owner's slab, k_bwd_dx's body): dz already split (below) -> 192 MFMAs -> LayerNorm/ReLU backward -> dz_prev
//     (stored, KT = 0) or the first-layer weight gradient on the matrix-pipe transposes (KT = 1, mfma_transpose.h);
//   D part, four rounds, one per slab of the super-round: wave w owns row tile w of dW' (output features 32 w .. 32 w + 31).
//     Its A operand -- that 32-feature block of the slab's dz, transposed -- it makes itself on the matrix pipe (its four float4
//     pieces of the block, split exactly, times the permuted identity: 6 MFMAs, no LDS, and the transposition's row sums are
//     db'); the B operand x_hat_prev^T is shared: every wave splits a quarter of the slab's pieces and stores them into the
//     round's buffer, the fragments come back through ds_read_b64_tr_b16.  Both are fetched a second time microseconds after
//     their owners touched them -- which, measured (rocprofv3 --pmc FETCH_SIZE, profiles/r05_hbm_traffic.md), does NOT stay in the
//     L2: 1.56 GB leave it per launch against 0.96 GB of operands (a super-round's working set per XCD, 32 workgroups x 4 slabs x
//     32 KB, is the L2's 4 MB); the pair of layer kernels fetched 1.83 GB.
// Version 1 of this kernel (one buffer, both operands through LDS, two barriers per round) measured no faster than the pair it
// replaces: with one workgroup per CU nothing overlaps a barrier wait (profiles/r05_bwd_fused_ab.md).  Hence the software
// pipeline: everything round n+1 needs is prepared DURING round n's 48 product MFMAs -- a wave has one MFMA in flight for 32
// cycles but needs ~12 of issue for it, and up to five VALU instructions placed BETWEEN two MFMAs are free
// (profiles/r03_mfma_valu_overlap.md) -- as filler chunks of 5-8 VALU pinned behind each MFMA with sched_barrier (source order
// is honoured exactly): the split of the next A block (16 chunks), its 6 transposing MFMAs, the packing of their results, the
// split of the next B pieces (16 chunks), and a share of the split of the OWNER's next slab of dz (64 chunks per super-round:
// the 2.2k cycles k_bwd_dx spends in "split dz" per slab).  The B terms are stored at the END of round n into the buffer round
// n-1 read, so ONE barrier per round suffices (every wave has left round n-1 when any wave stores for n+1), and it finds the
// stores long complete.  FILL = false runs the same chunks as a block in front of the product MFMAs (A/B).
// Per-workgroup partial rows in the layout of harl_reduce_partials_multi: dW' by tiles straight from the accumulators (the
// waves own disjoint row tiles), dW_1' through finish_partials; rows gridDim.x .. n_part_rows-1 of both arenas are cleared.
// =============================================================================================
__device__ __forceinline__ void split_stage1(float f0, float f1, unsigned &p1, float &r0, float &r1) {
  const unsigned b0 = __float_as_uint(f0), b1 = __float_as_uint(f1);
  p1 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  r0 = f0 - __uint_as_float(b0 & 0xffff0000u);
  r1 = f1 - __uint_as_float(b1 & 0xffff0000u);
}
__device__ __forceinline__ void split_stage2(float r0, float r1, unsigned &p2, unsigned &p3) {
  const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
  p2 = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
  const float q0 = r0 - __uint_as_float(c0 & 0xffff0000u), q1 = r1 - __uint_as_float(c1 & 0xffff0000u);
  p3 = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
}

// compile-time loop: f(std::integral_constant<int, I>) for I = I0 .. N-1 -- every index inside is a constant expression, whatever
// the unroller thinks of the body's size (a `#pragma unroll` loop that stays rolled turns register arrays into scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int BDW_SQ = (128 / 16) * 128 + 8, BDW_IMG = 8 * BDW_SQ;  // k_dw_tr's image geometry at H = 128
constexpr int BDW_BUF = 3 * BDW_IMG;                                // one transposition buffer: three term images
constexpr size_t bdw_lds_bytes() { return split_image_bytes(128, 128) + (size_t)2 * BDW_BUF; }

#ifndef HARL_BWD_OSPLIT_IN_GEMM
#define HARL_BWD_OSPLIT_IN_GEMM 1
#endif
constexpr bool OSPLIT_IN_GEMM = HARL_BWD_OSPLIT_IN_GEMM != 0;  // (0: the owner's split in the weight-gradient rounds, rounds 5 - 6; A/B builds)
template <int KT, bool FILL>
__global__ __launch_bounds__(WG_THREADS, 1) void k_bwd_dx_dw(
    const float *__restrict__ dz, const float *__restrict__ xprev, const uint32_t *__restrict__ mask_prev,
    const float *__restrict__ rstd_prev, const float *__restrict__ Wp, float *__restrict__ dz_prev, long n_slabs,
    const float *__restrict__ x0n, float *__restrict__ dw1_part, float *__restrict__ dw2_part, int n_part_rows) {
  constexpr int H = 128, MT = 4, NJ = 8, NR = 64, KPF = 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  PHASE_BEGIN();
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  unsigned char *Bb = reinterpret_cast<unsigned char *>(img + 3 * MT * NJ * 64);  // [2][3 terms][BDW_IMG]: x_hat_prev^T staging
  stage_split_matrix<H, H, true, WG_THREADS>(img, Wp);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  const u32x4 *wl = img + lane;
  const long sr_stride = (long)gridDim.x * WAVES_PER_WG;
  const long base0 = (long)blockIdx.x * WAVES_PER_WG;
  const Ident ident = make_ident(lane);

  // ---- persistent accumulators
  f32x16 acc2[4];                 // dW' tiles (row tile = wave, column tiles 0..3)
  f32x16 acc1[KT > 0 ? 4 : 1];    // dW_1' tiles (KT = 1)
  float dbs[KT > 0 ? 4 : 1];      // db_1' (per-lane sums over the lane's samples)
  float db2 = 0.f;                // db' of feature 32 wave + (lane & 31): this lane half's 16 samples of every slab
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[b][r] = 0.f;
#pragma unroll
  for (int a = 0; a < (KT > 0 ? 4 : 1); ++a) {
    dbs[a] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[a][r] = 0.f;
  }

  // ---- D-part state.  prA: the four float4 pieces 4 wave + u of a slab's dz = registers 16 wave .. 16 wave + 15 of the
  // accumulator layout = the 32-feature block `wave`; prB: pieces 4 wave + u of x_hat_prev (this wave's quarter of the B operand)
  f32x4 prA[4], prB[4];
  u32x2_t spB[4][3];          // split terms of prB: one ds_write_b64 each
  u32x4 ya[3][2];             // split terms of prA by k-step (split_transpose_block's y1 / y2 / y3)
  f32x16 tc[3];               // the three transposed terms (lane = feature, 16 samples)
  u32x4 At[3][2], AtN[3][2];  // A operands [term][k-step] of this round / of the next one
  float rr0 = 0.f, rr1 = 0.f; // remainders between the two stages of a pair's split
  auto d_load = [&](long ds) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      prA[u] = (reinterpret_cast<const f32x4 *>(dz + ds * (long)(H * SLAB)) + lane)[(4 * wave + u) * WAVE];
      prB[u] = (reinterpret_cast<const f32x4 *>(xprev + ds * (long)(H * SLAB)) + lane)[(4 * wave + u) * WAVE];
    }
  };
  // Preparing a round that will not be executed (past the last slab) is harmless -- its operands are never multiplied -- except
  // for the bias sums of its A block: `vnext` (1 or 0) multiplies them.  The loads of such a round are clamped to a valid slab.
  float vnext = 1.f;
  auto d_load_next = [&](long ds, long fallback) { d_load(ds < n_slabs ? ds : fallback); };
  // A chunk k (0..15): pair p = k >> 1 = registers 2p, 2p + 1 of the block -> word p & 3 of k-step p >> 2; stage k & 1
  auto a_chunk = [&](int k) {
    const int pi = k >> 1, j = pi >> 2, c = pi & 3, u = pi >> 1, e = 2 * (pi & 1);
    if ((k & 1) == 0) {
      unsigned p1;
      split_stage1(prA[u][e], prA[u][e + 1], p1, rr0, rr1);
      ya[0][j][c] = p1;
    } else {
      unsigned p2, p3;
      split_stage2(rr0, rr1, p2, p3);
      ya[1][j][c] = p2;
      ya[2][j][c] = p3;
    }
  };
  // B chunk k (0..15): piece k >> 2, pair (k >> 1) & 1, stage k & 1
  auto b_chunk = [&](int k) {
    const int u = k >> 2, c2 = (k >> 1) & 1;
    if ((k & 1) == 0) {
      unsigned p1;
      split_stage1(prB[u][2 * c2], prB[u][2 * c2 + 1], p1, rr0, rr1);
      spB[u][0][c2] = p1;
    } else {
      unsigned p2, p3;
      split_stage2(rr0, rr1, p2, p3);
      spB[u][1][c2] = p2;
      spB[u][2][c2] = p3;
    }
  };
  // transposing MFMA k (0..5) of the A block: term k % 3, k-step k / 3 (mfma_transpose.h, transpose_block)
  auto t_mfma = [&](int k) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int t = k % 3;
    if (k < 3) tc[t] = mfma_bf16(ya[t][0], ident.j0, zero);
    else tc[t] = mfma_bf16(ya[t][1], ident.j1, tc[t]);
  };
  // pack chunk k (0..5): k < 3: term k -> the two k-step operands; k >= 3: its row sums into db'
  auto p_chunk = [&](int k) {
    if (k < 3) {
      pack_transposed(tc[k], AtN[k][0], AtN[k][1]);
    } else {
      f32x2 a = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 8; ++r) a += f32x2{tc[k - 3][2 * r], tc[k - 3][2 * r + 1]};
      db2 += vnext * (a[0] + a[1]);
    }
  };
  auto b_store = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = 4 * wave + u;
      unsigned char *d = Bb + buf * BDW_BUF + (i >> 2) * BDW_SQ + (2 * (q >> 2) + ((q & 3) >> 1)) * 128 + (i & 3) * 32 +
                         (8 * (q & 1) + 4 * h) * 2;
#pragma unroll
      for (int term = 0; term < 3; ++term) *reinterpret_cast<u32x2_t *>(d + term * BDW_IMG) = spB[u][term];
    }
  };
  // B fragment of column tile `tile`, k-step ks: the eight samples sigma(r, h) = (r & 3) + 8 (r >> 2) + 4 h + 16 ks, r = 0..7 --
  // the order the matrix-pipe transposition leaves the A operand in (mfma_transpose.h): sample quads 4 ks + h and 4 ks + 2 + h
  const int p16 = lane & 15, g1b = (lane >> 4) & 1;
  const int frag_lane = g1b * 128 + (p16 >> 2) * 32 + (p16 & 3) * 8;
  auto read_frag = [&](int buf, int ks, int tile, u32x4 (&f)[3]) {
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      const unsigned char *fp = Bb + buf * BDW_BUF + term * BDW_IMG + (4 * ks + h) * BDW_SQ + 2 * tile * 128 + frag_lane;
      const u32x2_t lo = tr_read(fp), hi = tr_read(fp + 2 * BDW_SQ);
      f[term] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  };

  // ---- owner state: the current slab's dz split into the B operands of the GEMM, the next slab's raw dz
  float raw[NR];
  u32x4 g1[NJ], g2[NJ], g3[NJ];
  {
    const long s0 = base0 + wave;
    atl_load<H>(dz, s0 < n_slabs ? s0 : 0, lane, raw);
    if constexpr (!OSPLIT_IN_GEMM) split_acts<NR, false>(raw, g1, g2, g3);
  }
  // O chunk k (0..63) of the NEXT slab's split: pair k>>1 (registers 2p, 2p+1 -> word p&3 of k-step p>>2), stage k&1
  // (two streams of them run side by side in a round: each keeps its own remainders between the stages of a pair)
  float rx0 = 0.f, rx1 = 0.f;
  auto o_chunk = [&](int k, bool second_stream) {
    const int pi = k >> 1, j = pi >> 2, c = pi & 3;
    float &q0 = second_stream ? rx0 : rr0, &q1 = second_stream ? rx1 : rr1;
    if ((k & 1) == 0) {
      unsigned p1;
      split_stage1(raw[2 * pi], raw[2 * pi + 1], p1, q0, q1);
      g1[j][c] = p1;
    } else {
      unsigned p2, p3;
      split_stage2(q0, q1, p2, p3);
      g2[j][c] = p2;
      g3[j][c] = p3;
    }
  };
  if constexpr (OSPLIT_IN_GEMM) {  // k-step 0 of the first slab (later slabs: the last weight-gradient round's slots 32..39)
#pragma unroll
    for (int k = 0; k < 8; ++k) o_chunk(k, false);
  }
  // ---- the first round of the first super-round is prepared in the open
  if (base0 < n_slabs) {
    d_load(base0);
#pragma unroll
    for (int k = 0; k < 16; ++k) a_chunk(k);
#pragma unroll
    for (int k = 0; k < 6; ++k) t_mfma(k);
#pragma unroll
    for (int k = 0; k < 16; ++k) b_chunk(k);
#pragma unroll
    for (int k = 0; k < 6; ++k) p_chunk(k);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      At[t][0] = AtN[t][0];
      At[t][1] = AtN[t][1];
    }
    b_store(0);
  }
  PHASE(10);

  for (long base = base0; base < n_slabs; base += sr_stride) {
    const long slab = base + wave;
    const bool own = slab < n_slabs;
    const long nxt = slab + sr_stride < n_slabs ? slab + sr_stride : (own ? slab : 0);
    // =========================== O part: this wave's own slab ===========================
    if (own) {
      float xh[NR];
      atl_load<H>(xprev, slab, lane, xh);
      const float rstd = rstd_prev[slab * SLAB + i];
      uint32_t mbits[2];
#pragma unroll
      for (int w = 0; w < 2; ++w) mbits[w] = mask_prev[(slab * 2 + w) * WAVE + lane];
      f32x4 x0r[KT > 0 ? KPF / 8 : 1];
      if constexpr (KT > 0) {
        const f32x4 *bp = reinterpret_cast<const f32x4 *>(x0n + slab * (long)(KPF * SLAB)) + lane;
#pragma unroll
        for (int q = 0; q < KPF / 8; ++q) x0r[q] = bp[q * WAVE];
      }
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      PHASE(0);
      if constexpr (OSPLIT_IN_GEMM) {
        // the split of k-step j + 1 of THIS slab's dz (raw) behind the 24 product MFMAs of k-step j: one chunk (one stage of
        // a pair: ~5.5 VALU) behind every third MFMA -- the 192 shadows of this GEMM were empty while the weight-gradient rounds carried
        // 6.7 VALU per MFMA (profiles/r05_bwd_fused_ab.md), 64 of their 216 chunks being exactly these
        split_gemm_fill<MT, NJ>(wl, g1, g2, g3, acc, [&](int s_, int k6) {
          const int j1 = s_ / MT + 1, q = (s_ % MT) * 6 + k6;  // q = 0..23: the MFMA's position inside k-step j
          if (j1 < NJ && q % 3 == 0) o_chunk(8 * j1 + q / 3, false);
        });
      } else {
        split_gemm<MT, NJ>(wl, g1, g2, g3, acc, [](int) {});
      }
      PHASE(1);
      float dx[NR];
#pragma unroll
      for (int R = 0; R < NR; ++R) dx[R] = acc[R >> 4][R & 15];
      float out[NR];
      ln_bwd_relu_mbits<H>(dx, xh, mbits, rstd, out);
      if (dz_prev) atl_store<H>(dz_prev, slab, lane, out);
      PHASE(2);
      if constexpr (KT > 0) {
        float xr0[KPF / 2];
#pragma unroll
        for (int q = 0; q < KPF / 8; ++q) {
          xr0[4 * q + 0] = x0r[q][0];
          xr0[4 * q + 1] = x0r[q][1];
          xr0[4 * q + 2] = x0r[q][2];
          xr0[4 * q + 3] = x0r[q][3];
        }
        u32x4 a1[KPF / 16], a2[KPF / 16], a3[KPF / 16];
        split_acts<KPF / 2>(xr0, a1, a2, a3);
        u32x4 Bt[3][2];
        transpose_block<false>(a1[0], a1[1], a2[0], a2[1], a3[0], a3[1], ident, Bt);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          u32x4 A1[3][2];
          dbs[a] += split_transpose_block<true>(&out[16 * a], ident, A1);
          dw_tile(acc1[a], A1, Bt);
        }
        PHASE(3);
      }
    }
    // =========================== D part: the four slabs of this super-round ===========================
    // the owner's NEXT slab of dz (g1..g3 are dead since the GEMM; first needed by the fillers of round 1) and this wave's
    // pieces of round 1 (round 0's operands were prepared by the previous super-round's last round / the prologue) -- not
    // earlier: the first-layer gradient above is the register peak of this kernel
    atl_load<H>(dz, nxt, lane, raw);
    d_load_next(base + 1, base);
    PHASE(4);
    static_for<0, 4>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if (base + r < n_slabs) {  // workgroup-uniform
        __syncthreads();  // every wave's B terms of this round are in buffer r & 1; buffer (r + 1) & 1 is free
        PHASE(5);
        // the round being prepared (r + 1, or round 0 of the next super-round) and the one whose pieces are requested (r + 2; none
        // from round 3: the next super-round fetches its round 1 itself)
        const long ds1 = r < 3 ? base + r + 1 : base + sr_stride;
        const long ds2 = r < 2 ? base + r + 2 : base + sr_stride;
        vnext = uniform_f(ds1 < n_slabs ? 1.f : 0.f);
        // the owner's 64 split chunks: 0 / 22 / 22 / 20 per round, as a main stream of 16 (slots 32..47) and a second stream of
        // 6 / 6 / 4 beside it (slots 32..); every range starts at an even chunk = the first stage of a pair
        constexpr int O_FIRST = r == 1 ? 0 : (r == 2 ? 22 : 44), O_EXTRA = r == 3 ? 4 : 6;
        // TSLOT: the product MFMA behind which the six transposing MFMAs are issued
        constexpr int TSLOT = r == 0 ? 31 : 15;
        auto slot = [&](auto sc) {  // filler work behind product MFMA s (0..47)
          constexpr int s_ = decltype(sc)::value;
          if constexpr (r == 0) {  // the pieces of round 1 were requested just above: nothing in the first 16 slots
            if constexpr (s_ >= 16 && s_ < 32) a_chunk(s_ - 16);
            else if constexpr (s_ >= 32 && s_ < 42) b_chunk(s_ - 32 + 6);
            else if constexpr (s_ >= 42) p_chunk(s_ - 42);
            if constexpr (s_ == 41) d_load_next(ds2, base);
          } else {
            if constexpr (s_ < 16) a_chunk(s_);
            else if constexpr (s_ < 26) b_chunk(s_ - 16 + 6);
            else if constexpr (s_ < 32) p_chunk(s_ - 26);
            else if constexpr (!OSPLIT_IN_GEMM) {
              o_chunk(O_FIRST + s_ - 32, false);
              if constexpr (s_ - 32 < O_EXTRA) o_chunk(O_FIRST + 16 + s_ - 32, true);
            } else if constexpr (r == 3 && s_ < 40) {
              o_chunk(s_ - 32, false);  // k-step 0 of the owner's next slab; the other seven ride inside its dX GEMM
            }
            if constexpr (s_ == 25 && r < 3) d_load_next(ds2, base);
          }
          if constexpr (s_ == TSLOT) {  // the six transposing MFMAs, each with one of the first six B chunks behind it
            static_for<0, 6>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              __builtin_amdgcn_sched_barrier(0);
              t_mfma(k);
              b_chunk(k);
            });
          }
        };
        u32x4 bv[2][3];
        read_frag(r & 1, 0, 0, bv[0]);
        if constexpr (!FILL) static_for<0, 48>(slot);
        // ---- 2 k-steps x 4 column tiles x the six cross products; B fragments one tile-step ahead
        static_for<0, 8>([&](auto nc) {
          constexpr int n = decltype(nc)::value, ks = n >> 2, b = n & 3;
          if constexpr (n + 1 < 8) read_frag(r & 1, (n + 1) >> 2, (n + 1) & 3, bv[(n + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int AT[6] = {2, 0, 1, 1, 0, 0}, BT[6] = {0, 2, 1, 0, 1, 0};  // the six cross products, smallest first
          static_for<0, 6>([&](auto kc) {
            constexpr int k6 = decltype(kc)::value;
            acc2[b] = mfma_bf16(At[AT[k6]][ks], bv[n & 1][BT[k6]], acc2[b]);
            if constexpr (FILL) slot(std::integral_constant<int, 6 * n + k6>{});
            __builtin_amdgcn_sched_barrier(0);
          });
        });
        PHASE(6);
        // the next round's B terms into the buffer the PREVIOUS round read (every wave has passed this round's barrier, i.e.
        // left the previous round); its A operands become current
        b_store((r + 1) & 1);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          At[t][0] = AtN[t][0];
          At[t][1] = AtN[t][1];
        }
        PHASE(7);
      }
    });
  }

  // ---- this workgroup's partial row of dW' | db' straight from the accumulators (disjoint tiles), the unused rows cleared
  {
    constexpr long ROW2 = (long)H * H + H;
    float *mypart = dw2_part + (long)blockIdx.x * ROW2;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
        mypart[(long)o * H + 32 * b + i] = acc2[b][r];
      }
    const float dbt = wave_sum32(db2);
    if (h == 0) mypart[(long)H * H + 32 * wave + i] = dbt;
    for (int row = blockIdx.x + gridDim.x; row < n_part_rows; row += gridDim.x) {
      float *z = dw2_part + (long)row * ROW2;
      for (int e = threadIdx.x; e < ROW2; e += WG_THREADS) z[e] = 0.f;
    }
  }
  if constexpr (KT > 0) {
    f32x16 a1x[4][1];
#pragma unroll
    for (int a = 0; a < 4; ++a) a1x[a][0] = acc1[a];
    finish_partials<4, 1>(a1x, dbs, reinterpret_cast<float *>(img), dw1_part, n_part_rows);
  }
  PHASE(11);
  PHASE_END(3);
}

HARL_PHASE_ACCESSOR(mlp)

static int bad(const char *m) {
  set_error(m);
  return -2;
}

extern "C" int harl_mlp_fwd_input(const float *X, long ldx, const int64_t *idx, long M, int D, const float *Wp,
                                  const float *bp, int use_ln0, int H, float *xout, uint32_t *relu_mask, float *rstd,
                                  float *mu0, float *rstd0, float *x0n, void *stream) {
  if (M <= 0) return 0;
  const long n_slabs = n_slabs_of(M);
  const int nch = (D + 31) / 32;
  const int grid = persistent_grid(n_slabs, 2);
  hipStream_t s = (hipStream_t)stream;
  if (H != 128 && H != 64) return bad("harl_mlp_fwd_input: hidden width must be 64 or 128");
  if (D <= 64) {
    const int kr = D <= 32 ? 32 : 64;
    const size_t shm = ((size_t)kr * H + H + (size_t)WAVES_PER_WG * SLAB * (kr + 1)) * sizeof(float);
#define LS(Hv, R)                                                                                                 \
  {                                                                                                               \
    allow_big_lds(k_fwd_input_staged<Hv, R>, shm);                                                                \
    hipLaunchKernelGGL((k_fwd_input_staged<Hv, R>), dim3(grid), dim3(WG_THREADS), shm, s, X, ldx, idx, M, D, Wp, bp, \
                       use_ln0, xout, relu_mask, rstd, mu0, rstd0, x0n, n_slabs);                                  \
  }
    if (H == 128) {
      if (D <= 32) LS(128, 2) else LS(128, 1)
    } else {
      if (D <= 32) LS(64, 2) else LS(64, 1)
    }
#undef LS
    return check_launch("harl_mlp_fwd_input");
  }
  const int resident = (long)nch * 32 * H * 4 <= 64 * 1024;
  const size_t shm = ((size_t)(resident ? nch * 32 : 32) * H + H) * sizeof(float);
  if (H == 128) {
    allow_big_lds(k_fwd_input<128, false>, shm);
    hipLaunchKernelGGL((k_fwd_input<128, false>), dim3(grid), dim3(WG_THREADS), shm, s, X, ldx, idx, M, D, Wp, bp,
                       use_ln0, xout, relu_mask, rstd, mu0, rstd0, n_slabs, nch, resident, nullptr, nullptr, nullptr);
  } else {
    allow_big_lds(k_fwd_input<64, false>, shm);
    hipLaunchKernelGGL((k_fwd_input<64, false>), dim3(grid), dim3(WG_THREADS), shm, s, X, ldx, idx, M, D, Wp, bp,
                       use_ln0, xout, relu_mask, rstd, mu0, rstd0, n_slabs, nch, resident, nullptr, nullptr, nullptr);
  }
  return check_launch("harl_mlp_fwd_input");
}

// ---- forward-mode (tangent) pass, first layer:  x_hat1_dot = LNjac(mask1 * (W1'_dot norm0(X) + b1'_dot))
extern "C" int harl_mlp_tangent_input(const float *X, long ldx, const int64_t *idx, long M, int D, const float *Wdp,
                                      const float *bdp, int use_ln0, int H, const float *x1, const uint32_t *mask1,
                                      const float *rstd1, float *x1dot, void *stream) {
  if (M <= 0) return 0;
  const long n_slabs = n_slabs_of(M);
  const int nch = (D + 31) / 32;
  const int grid = persistent_grid(n_slabs, 2);
  hipStream_t s = (hipStream_t)stream;
  const int resident = (long)nch * 32 * H * 4 <= 64 * 1024;
  const size_t shm = ((size_t)(resident ? nch * 32 : 32) * H + H) * sizeof(float);
  if (H == 128) {
    allow_big_lds(k_fwd_input<128, true>, shm);
    hipLaunchKernelGGL((k_fwd_input<128, true>), dim3(grid), dim3(WG_THREADS), shm, s, X, ldx, idx, M, D, Wdp, bdp,
                       use_ln0, x1dot, nullptr, nullptr, nullptr, nullptr, n_slabs, nch, resident, x1, mask1, rstd1);
  } else if (H == 64) {
    allow_big_lds(k_fwd_input<64, true>, shm);
    hipLaunchKernelGGL((k_fwd_input<64, true>), dim3(grid), dim3(WG_THREADS), shm, s, X, ldx, idx, M, D, Wdp, bdp,
                       use_ln0, x1dot, nullptr, nullptr, nullptr, nullptr, n_slabs, nch, resident, x1, mask1, rstd1);
  } else {
    return bad("harl_mlp_tangent_input: hidden width must be 64 or 128");
  }
  return check_launch("harl_mlp_tangent_input");
}

// ---- forward-mode pass, hidden layer:  z_dot = W' x_hat_in_dot + W'_dot x_hat_in + b'_dot ; x_hat_out_dot = LNjac(.)
// (k_fwd_hidden MODE 1 then MODE 2)
extern "C" int harl_mlp_tangent_hidden(const float *xin_dot, const float *xin, long M, int HI, int HO, const float *Wp,
                                       const float *Wdp, const float *bdp, const float *xprimal,
                                       const uint32_t *mask_in, const float *rstd_in, float *xout_dot, void *stream) {
  if (M <= 0) return 0;
  const long n_slabs = n_slabs_of(M);
  // two split-bf16 GEMMs through xout_dot:  z = Wdp x_hat_in + bdp (raw), then  z += Wp x_in_dot  and the LayerNorm Jacobian
  // (round 1 ran both products on the fp32 MFMA from two fp32 weight copies in LDS: 0.19 ms at 204 800 rows, a quarter of
  // the 17-agent HATRPO update)
  const size_t shm = split_image_bytes(HO, HI) + (size_t)HO * sizeof(float);
  const int grid = persistent_grid(n_slabs, split_one_wg(HO, HI) ? 1 : 2);
  hipStream_t s = (hipStream_t)stream;
#define L(a, b)                                                                                                      \
  {                                                                                                                  \
    allow_big_lds(k_fwd_hidden<a, b, 1>, shm);                                                                       \
    hipLaunchKernelGGL((k_fwd_hidden<a, b, 1>), dim3(grid), dim3(WG_THREADS), shm, s, xin, Wdp, bdp, xout_dot, nullptr, \
                       nullptr, n_slabs, nullptr, nullptr, nullptr);                                                 \
    allow_big_lds(k_fwd_hidden<a, b, 2>, shm);                                                                       \
    hipLaunchKernelGGL((k_fwd_hidden<a, b, 2>), dim3(grid), dim3(WG_THREADS), shm, s, xin_dot, Wp, nullptr, xout_dot,  \
                       nullptr, nullptr, n_slabs, xprimal, mask_in, rstd_in);                                        \
  }
  if (HI == 128 && HO == 128) L(128, 128)
  else if (HI == 64 && HO == 64) L(64, 64)
  else if (HI == 128 && HO == 64) L(128, 64)
  else if (HI == 64 && HO == 128) L(64, 128)
  else return bad("harl_mlp_tangent_hidden: widths must be 64 or 128");
#undef L
  return check_launch("harl_mlp_tangent_hidden");
}

// plain Linear, no epilogue: xout = Wp xin + bp as an ATL(HO) image (k_fwd_hidden MODE 1) -- the logits of the concatenated
// MultiDiscrete heads (csrc/multihead.hip); Wp is [HO][HI] with zero rows past the last head
// Three independent raw products  z_g = W_g x_g + b_g  (g = 0, 1, 2) in ONE launch: blockIdx.y picks the operand set, the body is
// k_fwd_hidden's MODE 1.  The composed 128-wide GRU (harl_amd/gru_wide.py) issues the three gate products of every time step --
// same rows, three weight blocks -- and was 16 698 launches of 9.6 us per 8-agent HATRPO update (24 147 launches in 375 ms, 18 %
// of it with the GPU idle: rocprofv3 trace, round 6); one launch with three times the workgroups takes the time of one.
struct Lin3 {
  const float *x[3], *W[3], *b[3];
  float *out[3];
};
template <int HI, int HO>
__global__ __launch_bounds__(WG_THREADS, split_one_wg(HO, HI) ? 1 : 2) void k_linear3(Lin3 P, long n_slabs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = HO / 32, NJ = HI / 16, NR = HI / 2;
  const int g = blockIdx.y;
  const float *__restrict__ xin = P.x[g];
  const float *__restrict__ Wp = P.W[g];
  const float *__restrict__ bp = P.b[g];
  float *__restrict__ xout = P.out[g];
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  float *bl = reinterpret_cast<float *>(img + 3 * MT * NJ * 64);
  stage_split_matrix<HO, HI, false, WG_THREADS>(img, Wp);
  for (int e = threadIdx.x; e < HO; e += WG_THREADS) bl[e] = bp[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5;
  const long slab0 = (long)blockIdx.x * WAVES_PER_WG + wave, slab_stride = (long)gridDim.x * WAVES_PER_WG;
  const u32x4 *wl = img + lane;
  float raw[NR];
  atl_load<HI>(xin, slab0 < n_slabs ? slab0 : 0, lane, raw);
  for (long slab = slab0; slab < n_slabs; slab += slab_stride) {
    u32x4 x1[NJ], x2[NJ], x3[NJ];
    split_acts<NR>(raw, x1, x2, x3);
    atl_load<HI>(xin, slab + slab_stride < n_slabs ? slab + slab_stride : slab, lane, raw);
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = bl[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    split_gemm<MT, NJ>(wl, x1, x2, x3, acc, [](int) {});
    float z[HO / 2];
#pragma unroll
    for (int R = 0; R < HO / 2; ++R) z[R] = acc[R >> 4][R & 15];
    atl_store<HO>(xout, slab, lane, z);
  }
}

extern "C" int harl_mlp_linear3(const float *x0, const float *x1, const float *x2, long M, int HI, int HO, const float *W0,
                                const float *W1, const float *W2, const float *b0, const float *b1, const float *b2, float *o0,
                                float *o1, float *o2, void *stream) {
  if (M <= 0) return 0;
  if (!x0 || !x1 || !x2 || !W0 || !W1 || !W2 || !b0 || !b1 || !b2 || !o0 || !o1 || !o2) return bad("harl_mlp_linear3: NULL operand");
  const long n_slabs = n_slabs_of(M);
  const size_t shm = split_image_bytes(HO, HI) + (size_t)HO * sizeof(float);
  // a third of the persistent grid per product (at least what the slabs need): the three share the chip
  int grid = persistent_grid(n_slabs, split_one_wg(HO, HI) ? 1 : 2);
  const long need = (n_slabs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  if (grid > need) grid = (int)need;
  if (grid < 1) grid = 1;
  const Lin3 P = {{x0, x1, x2}, {W0, W1, W2}, {b0, b1, b2}, {o0, o1, o2}};
  hipStream_t s = (hipStream_t)stream;
#define L3(a, b)                                                                                                \
  {                                                                                                             \
    allow_big_lds(k_linear3<a, b>, shm);                                                                        \
    hipLaunchKernelGGL((k_linear3<a, b>), dim3(grid, 3), dim3(WG_THREADS), shm, s, P, n_slabs);                 \
  }
  if (HI == 128 && HO == 128) L3(128, 128)
  else if (HI == 64 && HO == 64) L3(64, 64)
  else return bad("harl_mlp_linear3: widths must be 64 x 64 or 128 x 128");
#undef L3
  return check_launch("harl_mlp_linear3");
}

extern "C" int harl_mlp_linear(const float *xin, long M, int HI, int HO, const float *Wp, const float *bp, float *xout,
                               void *stream) {
  if (M <= 0) return 0;
  const long n_slabs = n_slabs_of(M);
  const size_t shm = split_image_bytes(HO, HI) + (size_t)HO * sizeof(float);
  const int grid = persistent_grid(n_slabs, split_one_wg(HO, HI) ? 1 : 2);
  hipStream_t s = (hipStream_t)stream;
#define L(a, b)                                                                                                      \
  {                                                                                                                  \
    allow_big_lds(k_fwd_hidden<a, b, 1>, shm);                                                                       \
    hipLaunchKernelGGL((k_fwd_hidden<a, b, 1>), dim3(grid), dim3(WG_THREADS), shm, s, xin, Wp, bp, xout, nullptr, nullptr, \
                       n_slabs, nullptr, nullptr, nullptr);                                                          \
  }
  if (HI == 128 && HO == 128) L(128, 128)
  else if (HI == 64 && HO == 64) L(64, 64)
  else if (HI == 128 && HO == 64) L(128, 64)
  else if (HI == 64 && HO == 128) L(64, 128)
  else return bad("harl_mlp_linear: widths must be 64 or 128");
#undef L
  return check_launch("harl_mlp_linear");
}

extern "C" int harl_mlp_fwd_fused2(const float *X, long ldx, const int64_t *idx, long M, int D, const float *W1p,
                                   const float *b1p, int use_ln0, const float *W2p, const float *b2p, int H, int store1,
                                   float *x1out, uint32_t *mask1, float *rstd1, float *mu0, float *rstd0, float *x2out,
                                   uint32_t *mask2, float *rstd2, float *x0n, void *stream) {
  if (M <= 0) return 0;
  if (D > 32) return bad("harl_mlp_fwd_fused2: input width must be <= 32 (LDS budget)");
  if (H != 128 && H != 64) return bad("harl_mlp_fwd_fused2: hidden width must be 64 or 128");
  const long n_slabs = n_slabs_of(M);
  const int nch = (D + 31) / 32;
  const size_t shm = split_image_bytes(H, H) + (2 * H + (size_t)nch * 32 * H + (size_t)FUSED_WAVES * SLAB * (nch * 32 + 1)) * sizeof(float);
  long wgs = (n_slabs + FUSED_WAVES - 1) / FUSED_WAVES;
  const int grid = (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
  hipStream_t s = (hipStream_t)stream;
#define LF(Hv, R, C)                                                                                                 \
  {                                                                                                                  \
    allow_big_lds(k_fwd_fused2<Hv, R, C>, shm);                                                                      \
    hipLaunchKernelGGL((k_fwd_fused2<Hv, R, C>), dim3(grid), dim3(64 * FUSED_WAVES), shm, s, X, ldx, idx, M, D, W1p,  \
                       b1p, use_ln0, W2p, b2p, store1, x1out, mask1, rstd1, mu0, rstd0, x2out, mask2, rstd2, x0n, n_slabs); \
  }
  if (H == 128) LF(128, 2, 1) else LF(64, 2, 1)
#undef LF
  return check_launch("harl_mlp_fwd_fused2");
}

extern "C" int harl_mlp_fwd_hidden(const float *xin, long M, int HI, int HO, const float *Wp, const float *bp,
                                   float *xout, uint32_t *relu_mask, float *rstd, void *stream) {
  if (M <= 0) return 0;
  const long n_slabs = n_slabs_of(M);
  const size_t shm = split_image_bytes(HO, HI) + (size_t)HO * sizeof(float);
  const int grid = persistent_grid(n_slabs, split_one_wg(HO, HI) ? 1 : 2);
  hipStream_t s = (hipStream_t)stream;
#define L(a, b)                                                                                                  \
  allow_big_lds(k_fwd_hidden<a, b>, shm);                                                                        \
  hipLaunchKernelGGL((k_fwd_hidden<a, b>), dim3(grid), dim3(WG_THREADS), shm, s, xin, Wp, bp, xout, relu_mask, rstd, \
                     n_slabs, nullptr, nullptr, nullptr)
  if (HI == 128 && HO == 128) { L(128, 128); }
  else if (HI == 64 && HO == 64) { L(64, 64); }
  else if (HI == 128 && HO == 64) { L(128, 64); }
  else if (HI == 64 && HO == 128) { L(64, 128); }
  else return bad("harl_mlp_fwd_hidden: widths must be 64 or 128");
#undef L
  return check_launch("harl_mlp_fwd_hidden");
}

// HARL_BWD_STREAMS=1 (read once): the register-lean instantiation of the fused first-layer variant, for co-residency with the
// weight-gradient launch of the same layer on a second stream (nets.backward_trunk)
static bool lean_bwd() {
  static const bool v = [] {
    const char *e = getenv("HARL_BWD_STREAMS");
    return e && e[0] == '1';
  }();
  return v;
}

extern "C" int harl_mlp_bwd_dx(const float *dz, const float *xprev, const uint32_t *relu_mask_prev,
                               const float *rstd_prev, long M, int HO, int HI, const float *Wp, float *dz_prev,
                               const float *x0n, int kp0, float *dw_part, int n_wg, void *stream) {
  if (M <= 0) return 0;
  const long n_slabs = n_slabs_of(M);
  hipStream_t s = (hipStream_t)stream;
  if (dw_part) {  // first-layer variant with the weight gradient fused in
    if (!x0n || (kp0 != 32 && kp0 != 64) || n_wg <= 0)
      return bad("harl_mlp_bwd_dx: fused first-layer gradient needs x0n, kp0 in {32, 64} and n_wg > 0");
  // LDS: the three weight images; the end-of-kernel combine of the fused first-layer gradient (finish_partials) works in
  // their place, so nothing else is needed -- unless one partial row [b][32 kt] | [b] is larger than the images (64 -> 64, kt = 2)
#define LFX(a, b, kt, lean)                                                                                       \
  {                                                                                                               \
    size_t shm = split_image_bytes(b, a);                                                                         \
    const size_t rowb = ((size_t)b * 32 * kt + b) * sizeof(float);                                                \
    if (shm < rowb) shm = rowb;                                                                                   \
    allow_big_lds(k_bwd_dx<a, b, kt, lean>, shm);                                                                 \
    hipLaunchKernelGGL((k_bwd_dx<a, b, kt, lean>), dim3(n_wg < 256 ? n_wg : 256), dim3(WG_THREADS), shm, s, dz,   \
                       xprev, relu_mask_prev, rstd_prev, Wp, dz_prev, n_slabs, x0n, dw_part, n_wg);               \
  }
#define LF(a, b, kt) LFX(a, b, kt, false)
    const int kt = kp0 / 32;
    if (HO == 128 && HI == 128 && kt == 1 && lean_bwd()) LFX(128, 128, 1, true)
    else if (HO == 128 && HI == 128) { if (kt == 1) LF(128, 128, 1) else LF(128, 128, 2) }
    else if (HO == 64 && HI == 64) { if (kt == 1) LF(64, 64, 1) else LF(64, 64, 2) }
    else if (HO == 128 && HI == 64) { if (kt == 1) LF(128, 64, 1) else LF(128, 64, 2) }
    else if (HO == 64 && HI == 128) { if (kt == 1) LF(64, 128, 1) else LF(64, 128, 2) }
    else return bad("harl_mlp_bwd_dx: widths must be 64 or 128");
#undef LF
#undef LFX
    return check_launch("harl_mlp_bwd_dx");
  }
  const size_t shm = split_image_bytes(HI, HO);
  const int grid = persistent_grid(n_slabs, split_one_wg(HI, HO) ? 1 : 2);
#define L(a, b)                                                                                                    \
  allow_big_lds(k_bwd_dx<a, b>, shm);                                                                              \
  hipLaunchKernelGGL((k_bwd_dx<a, b>), dim3(grid), dim3(WG_THREADS), shm, s, dz, xprev, relu_mask_prev, rstd_prev, Wp, \
                     dz_prev, n_slabs, nullptr, nullptr, 0)
  if (HO == 128 && HI == 128) { L(128, 128); }
  else if (HO == 64 && HI == 64) { L(64, 64); }
  else if (HO == 128 && HI == 64) { L(128, 64); }
  else if (HO == 64 && HI == 128) { L(64, 128); }
  else return bad("harl_mlp_bwd_dx: widths must be 64 or 128");
#undef L
  return check_launch("harl_mlp_bwd_dx");
}

extern "C" int harl_mlp_bwd_dx_dw(const float *dz, const float *xprev, const uint32_t *relu_mask_prev,
                                  const float *rstd_prev, long M, int HO, int HI, const float *Wp, float *dz_prev,
                                  const float *x0n, int kp0, float *dw1_part, float *dw2_part, int n_wg, int fill,
                                  void *stream) {
  if (M <= 0) return 0;
  if (HO != 128 || HI != 128) return bad("harl_mlp_bwd_dx_dw: 128 x 128 layers only (the layer kernels take the other widths)");
  if (!dw2_part || n_wg <= 0) return bad("harl_mlp_bwd_dx_dw: needs the partial arena of dW' and n_wg > 0");
  const bool first = dw1_part != nullptr;
  if (first && (!x0n || kp0 != 32)) return bad("harl_mlp_bwd_dx_dw: the fused first-layer gradient needs x0n with kp0 = 32");
  if (!first && !dz_prev) return bad("harl_mlp_bwd_dx_dw: dz_prev is required when no first-layer gradient is fused");
  const long n_slabs = n_slabs_of(M);
  const long wgs = (n_slabs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  int grid = n_wg < 256 ? n_wg : 256;
  if (wgs < grid) grid = (int)wgs;
  const size_t shm = bdw_lds_bytes();
  hipStream_t s = (hipStream_t)stream;
#define LB(kt, fl)                                                                                                    \
  {                                                                                                                   \
    allow_big_lds(k_bwd_dx_dw<kt, fl>, shm);                                                                          \
    hipLaunchKernelGGL((k_bwd_dx_dw<kt, fl>), dim3(grid), dim3(WG_THREADS), shm, s, dz, xprev, relu_mask_prev,       \
                       rstd_prev, Wp, dz_prev, n_slabs, x0n, dw1_part, dw2_part, n_wg);                              \
  }
  if (first) { if (fill) LB(1, true) else LB(1, false) }
  else { if (fill) LB(0, true) else LB(0, false) }
#undef LB
  return check_launch("harl_mlp_bwd_dx_dw");
}

template <int MT, int NT>
static void launch_dw_tr(const float *a, const float *b, long n_slabs, float *part, int K, int tile0, int n_wg, hipStream_t s) {
  const size_t shm = (size_t)3 * 8 * ((2 * MT * 128 + 8) + (2 * NT * 128 + 8));
  allow_big_lds(k_dw_tr<MT, NT>, shm);
  hipLaunchKernelGGL((k_dw_tr<MT, NT>), dim3(n_wg), dim3(WG_THREADS), shm, s, a, b, n_slabs, part, (long)K * SLAB, tile0, K);
}

template <int A_KIND, int B_KIND, int MT, int NT>
static void launch_dw(const float *a, const float *b, long ldx, const int64_t *idx, const float *mu0,
                      const float *rstd0, int K, long M, long n_slabs, float *part, int KP, int n_wg, int ny,
                      hipStream_t s) {
  const size_t shm = ((size_t)DW_S * ((32 * MT + 4) + (32 * NT + 4)) + 2 * DW_S) * sizeof(float);
  hipLaunchKernelGGL((k_dw<A_KIND, B_KIND, MT, NT>), dim3(n_wg, ny), dim3(WG_THREADS), shm, s, a, b, ldx, idx, mu0,
                     rstd0, K, M, n_slabs, part, KP);
}

extern "C" int harl_mlp_dw_partials(const float *a, int a_kind, int lda, int HO, const float *b, int b_kind, long ldx,
                                    const int64_t *idx, const float *mu0, const float *rstd0, int K, long M,
                                    float *part, int n_wg, void *stream) {
  if (M <= 0 || n_wg <= 0) return 0;
  const long n_slabs = n_slabs_of(M);
  const int KP = ((K + 31) / 32) * 32;
  hipStream_t s = (hipStream_t)stream;
  if (a_kind == 1 && lda != DHEAD_LD) return bad("harl_mlp_dw_partials: head gradient matrix must have row stride 32");
  const int MT = a_kind == 1 ? 1 : HO / 32;
  if (a_kind == 0 && HO != 64 && HO != 128 && HO != 256) return bad("harl_mlp_dw_partials: HO must be 64, 128 or 256");
  if (a_kind == 1 && HO > 32) return bad("harl_mlp_dw_partials: head width must be <= 32");
#define DW(AK, BK, MTv, NTv, ny) launch_dw<AK, BK, MTv, NTv>(a, b, ldx, idx, mu0, rstd0, K, M, n_slabs, part, KP, n_wg, ny, s)
  if (b_kind == 0) {
    const int NT = K / 32;
    if (K % 32 != 0 || K < 32) return bad("harl_mlp_dw_partials: ATL input width must be a multiple of 32");
#define DWS(MTv, NTv) launch_dw_tr<MTv, NTv>(a, b, n_slabs, part, K, tile0, n_wg, s);
    if (K > 128 || K == 96 || MT == 8) {  // wide first layer (and every 256-row operand): x0n ATL(K), K a multiple of 32 up to 512, in groups of <= 4 column tiles
      if (K % 32 != 0 || K > 512 || (MT != 8 && MT != 4 && MT != 2)) return bad("harl_mlp_dw_partials: wide ATL input must be a multiple of 32, <= 512");
      // every launch re-reads and re-splits dz: as few column groups as the register file allows (8 x 2 tiles for 256-wide
      // layers; up to 6 tiles next to 4 row tiles: a 416-wide first layer is 5 + 5 + 3 instead of 4 + 4 + 4 + 1)
      const int ntiles = K / 32;
      const int gstep = MT == 8 ? 2 : (MT == 4 && ntiles > 4 ? (ntiles + (ntiles + 5) / 6 - 1) / ((ntiles + 5) / 6) : 4);
      for (int tile0 = 0; tile0 < ntiles; tile0 += gstep) {
        const int nt = ntiles - tile0 < gstep ? ntiles - tile0 : gstep;
        if (MT == 8) {
          if (nt == 2) DWS(8, 2) else DWS(8, 1)
        } else if (MT == 4) {
          if (nt == 6) DWS(4, 6) else if (nt == 5) DWS(4, 5) else
          if (nt == 4) DWS(4, 4) else if (nt == 3) DWS(4, 3) else if (nt == 2) DWS(4, 2) else DWS(4, 1)
        } else {
          if (nt == 4) DWS(2, 4) else if (nt == 3) DWS(2, 3) else if (nt == 2) DWS(2, 2) else DWS(2, 1)
        }
      }
      return check_launch("harl_mlp_dw_partials");
    }
    const int tile0 = 0;
    if (MT == 4 && NT == 1) DW(0, 0, 4, 1, 1);
    else if (MT == 2 && NT == 1) DW(0, 0, 2, 1, 1);
    else if (MT == 4 && NT == 4) DWS(4, 4)
    else if (MT == 4 && NT == 2) DWS(4, 2)
    else if (MT == 2 && NT == 4) DWS(2, 4)
    else if (MT == 2 && NT == 2) DWS(2, 2)
#undef DWS
    else if (MT == 1 && NT == 4) DW(1, 0, 1, 4, 1);
    else if (MT == 1 && NT == 2) DW(1, 0, 1, 2, 1);
    else return bad("harl_mlp_dw_partials: unsupported tile shape");
  } else {
    const int ktiles = KP / 32;
    if (a_kind != 0) return bad("harl_mlp_dw_partials: raw-input B operand needs an ATL A operand");
    if (ktiles == 1) {
      if (MT == 4) DW(0, 1, 4, 1, 1); else DW(0, 1, 2, 1, 1);
    } else if (ktiles == 2) {
      if (MT == 4) DW(0, 1, 4, 2, 1); else DW(0, 1, 2, 2, 1);
    } else {
      const int ny = (ktiles + 3) / 4;
      if (MT == 4) DW(0, 1, 4, 4, ny); else DW(0, 1, 2, 4, ny);
    }
  }
#undef DW
  return check_launch("harl_mlp_dw_partials");
}

extern "C" int harl_mlp_dw_partials_multi(int n, const float *const *a, const float *const *b, float *const *part, int HO,
                                          int K, long M, int n_wg, void *stream) {
  if (M <= 0 || n_wg <= 0 || n <= 0) return 0;
  if (n > DW_MULTI_MAX) return bad("harl_mlp_dw_partials_multi: at most 8 problems per launch");
  const long n_slabs = n_slabs_of(M);
  DwMulti P;
  for (int k = 0; k < DW_MULTI_MAX; ++k) {
    P.a[k] = a[k < n ? k : 0];
    P.b[k] = b[k < n ? k : 0];
    P.part[k] = part[k < n ? k : 0];
  }
  hipStream_t s = (hipStream_t)stream;
#define DWM(MTv, NTv)                                                                                      \
  {                                                                                                        \
    const size_t shm = (size_t)3 * 8 * ((2 * MTv * 128 + 8) + (2 * NTv * 128 + 8));                        \
    allow_big_lds(k_dw_tr_multi<MTv, NTv>, shm);                                                           \
    hipLaunchKernelGGL((k_dw_tr_multi<MTv, NTv>), dim3(n_wg, n), dim3(WG_THREADS), shm, s, P, n_slabs, K);  \
  }
  if (HO == 64 && K == 64) DWM(2, 2)
  else if (HO == 128 && K == 128) DWM(4, 4)
  else return bad("harl_mlp_dw_partials_multi: square 64 / 128 blocks only");
#undef DWM
  return check_launch("harl_mlp_dw_partials_multi");
}

extern "C" int harl_mlp_dw_partials_multi_v(int n, const float *const *a, const float *const *b, float *const *part, int HO,
                                            const int *K, const int *tile0, const int *nt, long M, int n_wg, void *stream) {
  if (M <= 0 || n_wg <= 0 || n <= 0) return 0;
  if (n > DW_MULTIV_MAX) return bad("harl_mlp_dw_partials_multi_v: at most 12 problems per launch");
  if (HO != 64) return bad("harl_mlp_dw_partials_multi_v: 64-row operands only");
  const long n_slabs = n_slabs_of(M);
  DwMultiV P;
  int ntmax = 1;
  for (int k = 0; k < DW_MULTIV_MAX; ++k) {
    const int j = k < n ? k : 0;
    if (K[j] % 32 != 0 || K[j] < 32 || K[j] > 512 || nt[j] < 1 || nt[j] > 4 || tile0[j] < 0 || 32 * (tile0[j] + nt[j]) > K[j])
      return bad("harl_mlp_dw_partials_multi_v: K must be a multiple of 32 up to 512, 1 <= nt <= 4, the group inside the image");
    P.a[k] = a[j];
    P.b[k] = b[j];
    P.part[k] = part[j];
    P.K[k] = K[j];
    P.tile0[k] = tile0[j];
    P.nt[k] = nt[j];
    if (nt[j] > ntmax) ntmax = nt[j];
  }
  const size_t shm = (size_t)3 * 8 * ((2 * 2 * 128 + 8) + (2 * 4 * 128 + 8));  // k_dw_tr<2, 4>'s staging area
  allow_big_lds(k_dw_tr_multi_v<2>, shm);
  hipLaunchKernelGGL((k_dw_tr_multi_v<2>), dim3(n_wg, n), dim3(WG_THREADS), shm, (hipStream_t)stream, P, n_slabs);
  return check_launch("harl_mlp_dw_partials_multi_v");
}

extern "C" int harl_gru_dw6(const float *dr, const float *dz, const float *dn, const float *dhn, const float *xhat, const float *hpm,
                            float *const *part, long M, int n_wg, void *stream) {
  if (M <= 0 || n_wg <= 0) return 0;
  DwGru6 P;
  P.a[0] = dr;
  P.a[1] = dz;
  P.a[2] = dn;
  P.a[3] = dhn;
  P.b[0] = xhat;
  P.b[1] = hpm;
  for (int k = 0; k < 6; ++k) P.part[k] = part[k];
  const size_t shm = (size_t)3 * 8 * ((2 * 8 * 128 + 8) + (2 * 4 * 128 + 8));
  allow_big_lds(k_dw_gru6, shm);
  hipLaunchKernelGGL(k_dw_gru6, dim3(n_wg), dim3(64 * G6_WAVES), shm, (hipStream_t)stream, P, n_slabs_of(M));
  return check_launch("harl_gru_dw6");
}
