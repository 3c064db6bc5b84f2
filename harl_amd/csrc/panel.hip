// panel.hip -- hidden width 256 (harl/models/base/mlp.py:7-70 with hidden_sizes = [256, 256, 256]: the reference's dexhands
// HAPPO configurations), gfx950.
//
// The 64/128-wide layer kernels (mlp.hip) keep the three bf16 images of the whole weight matrix in LDS for the lifetime of a
// persistent workgroup.  For 256 outputs that is 3 x 256 x K x 2 B = 393 KB at K = 256: it does not fit.  Here the K
// dimension is walked in PANELS of 32 columns: per panel the workgroup stages the images of W[:, 32 p .. 32 p + 31]
// (3 x 8 row tiles x 2 k-steps x 1 KiB = 48 KiB), every wave multiplies it into the 8 accumulator tiles of ITS slab, and the
// next panel replaces it.  A workgroup-iteration is 4 slabs (128 samples), so the matrix streams from L2 once per 128
// samples; the dexhands batches are 19 200 - 32 000 rows per agent, i.e. this path is about coverage, not about the roof.
//   forward  : x_hat_out = norm(relu(W' x_in + b'))            x_in  = ATL(KP) image (normalised inputs x0n, or x_hat of the layer before)
//   backward : dz_in = relu' . LNbwd(W'^T dz_out)               (the layer kernels' harl_mlp_bwd_dx for 256-wide layers)
// Weight gradients are harl_mlp_dw_partials (k_dw_tr<8, NT>, mlp.hip) over the same ATL images.
#include "common.h"
#include "split_mfma.h"
#include "../../include/harl_hip.h"

using namespace harl;

namespace {
int bad(const char *m) {
  set_error(m);
  return -2;
}

constexpr int PH = 256, PMT = PH / 32, PNJ = 2;            // 8 row tiles, 2 k-steps (32 columns) per panel
constexpr int PANEL_IMG = PMT * PNJ * 64;                  // u32x4 fragments per term

// images of rows 0..255, k columns 32 p .. 32 p + 31:
//   TRANSPOSED = false: A[row][k] = Wp[row * ldw + k]  (k < kvalid, else 0)       forward, Wp = [256][ldw]
//   TRANSPOSED = true : A[row][k] = Wp[k * ldw + row]                              backward dX, Wp = [K][256]
template <bool TRANSPOSED>
__device__ __forceinline__ void stage_panel(u32x4 *__restrict__ img, const float *__restrict__ Wp, int ldw, int kvalid, int p) {
  for (int e = threadIdx.x; e < PANEL_IMG; e += WG_THREADS) {
    const int ln = e & 63, j = (e >> 6) % PNJ, t = (e >> 6) / PNJ, m = 32 * t + (ln & 31), g = ln >> 5;
    unsigned q[3][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k0 = 32 * p + feat_base(8 * j + 2 * c) + 4 * g, k1 = 32 * p + feat_base(8 * j + 2 * c + 1) + 4 * g;
      float w0, w1;
      if (TRANSPOSED) {
        w0 = k0 < kvalid ? Wp[(long)k0 * ldw + m] : 0.f;
        w1 = k1 < kvalid ? Wp[(long)k1 * ldw + m] : 0.f;
      } else {
        w0 = k0 < kvalid ? Wp[(long)m * ldw + k0] : 0.f;
        w1 = k1 < kvalid ? Wp[(long)m * ldw + k1] : 0.f;
      }
      split3_rne(w0, w1, q[0][c], q[1][c], q[2][c]);
    }
#pragma unroll
    for (int term = 0; term < 3; ++term) img[term * PANEL_IMG + e] = u32x4{q[term][0], q[term][1], q[term][2], q[term][3]};
  }
}

// acc[0..7] += panel p of the matrix x the slab's k-steps 2p, 2p+1 (pieces 4p .. 4p+3 of its ATL(KP) image)
__device__ __forceinline__ void panel_gemm(const u32x4 *__restrict__ wl, const f32x4 *__restrict__ xp, int p, f32x16 (&acc)[PMT]) {
  float xr[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = xp[(4 * p + q) * WAVE];
    xr[4 * q + 0] = v[0];
    xr[4 * q + 1] = v[1];
    xr[4 * q + 2] = v[2];
    xr[4 * q + 3] = v[3];
  }
  u32x4 x1[PNJ], x2[PNJ], x3[PNJ];
  split_acts<16>(xr, x1, x2, x3);
  split_gemm<PMT, PNJ>(wl, x1, x2, x3, acc, [](int) {});
}

// MODE 0 forward, 1 backward, 2 forward-mode tangent (HATRPO's Fisher-vector product on 256-wide layers, round 4):
//   z_dot = Wp (= W'_dot) x_in + bp (= b'_dot) [+ Wp2 (= W') x_in2 (= x_in_dot)],  x_out_dot = LNjac(mask . z_dot) with the PRIMAL
//   x_hat / mask / rstd of this layer in xprev / mask_prev / rstd_prev (the epilogue of mlp.hip's ln_jac_store)
template <int MODE>
__global__ __launch_bounds__(WG_THREADS, 1) void k_panel(const float *__restrict__ xin, int KP, const float *__restrict__ Wp,
                                                         int ldw, int kvalid, const float *__restrict__ bp,
                                                         float *__restrict__ xout, uint32_t *__restrict__ mask_out,
                                                         float *__restrict__ rstd_out, const float *__restrict__ xprev,
                                                         const uint32_t *__restrict__ mask_prev,
                                                         const float *__restrict__ rstd_prev, long n_slabs,
                                                         const float *__restrict__ xin2 = nullptr, int KP2 = 0,
                                                         const float *__restrict__ Wp2 = nullptr, int ldw2 = 0) {
  constexpr bool BWD = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31, h = lane >> 5;
  const u32x4 *wl = img + lane;
  const int n_panels = KP / 32;
  const long n_iter = (n_slabs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  for (long it = blockIdx.x; it < n_iter; it += gridDim.x) {  // uniform trip count: barriers inside
    const long s_raw = it * WAVES_PER_WG + wave;
    const bool live = s_raw < n_slabs;
    const long slab = live ? s_raw : n_slabs - 1;
    const f32x4 *xp = reinterpret_cast<const f32x4 *>(xin + slab * (long)KP * SLAB) + lane;
    f32x16 acc[PMT];
#pragma unroll
    for (int t = 0; t < PMT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = BWD ? 0.f : bp[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    for (int p = 0; p < n_panels; ++p) {
      __syncthreads();  // the previous panel's fragments are no longer being read
      stage_panel<BWD>(img, Wp, ldw, kvalid, p);
      __syncthreads();
      panel_gemm(wl, xp, p, acc);
    }
    if constexpr (MODE == 2) {
      if (xin2) {  // (kernel-uniform) the second half of the pre-activation's tangent: W' x_in_dot
        const f32x4 *xp2 = reinterpret_cast<const f32x4 *>(xin2 + slab * (long)KP2 * SLAB) + lane;
        for (int p = 0; p < KP2 / 32; ++p) {
          __syncthreads();
          stage_panel<false>(img, Wp2, ldw2, ldw2, p);
          __syncthreads();
          panel_gemm(wl, xp2, p, acc);
        }
      }
    }
    if (!live) continue;
    if constexpr (MODE == 2) {
      constexpr int NR = PH / 2, NW = (NR + 31) / 32;
      float xh[NR], ad[NR];
      atl_load<PH>(xprev, slab, lane, xh);
      uint32_t bits[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) bits[w] = mask_prev[(slab * NW + w) * WAVE + lane];
      const float rstd = rstd_prev[slab * SLAB + i];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int R = 0; R < NR; ++R) {
        ad[R] = mask_pop(acc[R >> 4][R & 15], bits[R >> 5]);
        s1 += ad[R];
        s2 += ad[R] * xh[R];
      }
      s1 = wave_sum32(s1) * (1.0f / PH);
      s2 = wave_sum32(s2) * (1.0f / PH);
#pragma unroll
      for (int R = 0; R < NR; ++R) ad[R] = rstd * (ad[R] - s1 - xh[R] * s2);
      atl_store<PH>(xout, slab, lane, ad);
    } else if constexpr (!BWD) {
      constexpr int NR = PH / 2, NW = (NR + 31) / 32;
      uint32_t bits[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) bits[w] = 0u;
      float v[NR];
      float sum = 0.f;
#pragma unroll
      for (int R = 0; R < NR; ++R) {
        v[R] = relu_push(acc[R >> 4][R & 15], bits[R >> 5]);
        sum += v[R];
      }
      sum = wave_sum32(sum);
      const float mean = sum * (1.0f / PH);
      float vs = 0.f;
#pragma unroll
      for (int R = 0; R < NR; ++R) {
        v[R] -= mean;
        vs += v[R] * v[R];
      }
      vs = wave_sum32(vs);
      const float rstd = 1.0f / sqrtf(vs * (1.0f / PH) + 1e-5f);
#pragma unroll
      for (int R = 0; R < NR; ++R) v[R] *= rstd;
      atl_store<PH>(xout, slab, lane, v);
#pragma unroll
      for (int w = 0; w < NW; ++w) mask_out[(slab * NW + w) * WAVE + lane] = bits[w];
      if (lane < 32) rstd_out[slab * SLAB + lane] = rstd;
    } else {
      float dx[PH / 2], xh[PH / 2], out[PH / 2];
#pragma unroll
      for (int R = 0; R < PH / 2; ++R) dx[R] = acc[R >> 4][R & 15];
      atl_load<PH>(xprev, slab, lane, xh);
      ln_bwd_relu_regs<PH>(dx, xh, mask_prev, rstd_prev[slab * SLAB + i], lane, slab, out);
      atl_store<PH>(xout, slab, lane, out);
    }
  }
}

int panel_grid(long n_slabs) {
  const long n_iter = (n_slabs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  return (int)(n_iter < 256 ? (n_iter < 1 ? 1 : n_iter) : 256);
}
}  // namespace

extern "C" int harl_mlp_panel_fwd(const float *xin, long M, int KP, const float *Wp, int D, const float *bp, int HO,
                                  float *xout, uint32_t *relu_mask, float *rstd, void *stream) {
  if (M <= 0) return 0;
  if (HO != PH) return bad("harl_mlp_panel_fwd: output width must be 256");
  if (KP % 32 != 0 || KP < D || KP > 512) return bad("harl_mlp_panel_fwd: KP must be a multiple of 32, >= D and <= 512");
  const long n_slabs = n_slabs_of(M);
  const size_t shm = (size_t)3 * PANEL_IMG * sizeof(u32x4);
  allow_big_lds(k_panel<0>, shm);
  hipLaunchKernelGGL(k_panel<0>, dim3(panel_grid(n_slabs)), dim3(WG_THREADS), shm, (hipStream_t)stream, xin, KP, Wp, D, D,
                     bp, xout, relu_mask, rstd, nullptr, nullptr, nullptr, n_slabs);
  return check_launch("harl_mlp_panel_fwd");
}

extern "C" int harl_mlp_panel_bwd(const float *dz, const float *xprev, const uint32_t *relu_mask_prev,
                                  const float *rstd_prev, long M, int HO, int HI, const float *Wp, float *dz_prev,
                                  void *stream) {
  if (M <= 0) return 0;
  if (HI != PH || HO != PH) return bad("harl_mlp_panel_bwd: both widths must be 256");
  const long n_slabs = n_slabs_of(M);
  const size_t shm = (size_t)3 * PANEL_IMG * sizeof(u32x4);
  allow_big_lds(k_panel<1>, shm);
  // dx_hat[i] = sum_o Wp[o][i] dz[o]: rows = input features, k = output features, Wp = [HO][HI] (row stride HI)
  hipLaunchKernelGGL(k_panel<1>, dim3(panel_grid(n_slabs)), dim3(WG_THREADS), shm, (hipStream_t)stream, dz, HO, Wp, HI, HO,
                     nullptr, dz_prev, nullptr, nullptr, xprev, relu_mask_prev, rstd_prev, n_slabs);
  return check_launch("harl_mlp_panel_bwd");
}

// ---- head weight gradient of a 256-wide trunk from the ROW-MAJOR head gradients [M_pad][32] (the separate pass HATRPO's
// surrogate gradient and Fisher-vector product use; HAPPO's loss kernel fuses it):  dWp[d][f] = sum_s dhead[s][d] x_hat[s][f],
// dbp[d] = sum_s dhead[s][d].  One thread per feature, D accumulators in registers, every workgroup a contiguous range of
// samples and ONE partial row dWp[32][256] | dbp[32] (rows >= D zero), the layout k_dw leaves for narrower trunks.  Coverage
// path: 0.3 GFLOP at the dexhands batch sizes.
__global__ __launch_bounds__(PH) void k_head_dw_rows256(const float *__restrict__ dhead, long M, int D,
                                                        const float *__restrict__ xhat, float *__restrict__ part, int n_wg) {
  const int f = threadIdx.x;  // feature
  const int t = f >> 5, w = f & 31, hh = (w >> 2) & 1, r = (w & 3) + 4 * (w >> 3), R = 16 * t + r;
  const long per = (M + n_wg - 1) / n_wg;
  const long s0 = (long)blockIdx.x * per, s1 = s0 + per < M ? s0 + per : M;
  float acc[32], db = 0.f;
#pragma unroll
  for (int d = 0; d < 32; ++d) acc[d] = 0.f;
  __shared__ float dh[32];
  for (long s = s0; s < s1; ++s) {
    __syncthreads();
    if (f < 32) dh[f] = f < D ? dhead[s * DHEAD_LD + f] : 0.f;
    __syncthreads();
    const long slab = s >> 5;
    const int lane = (int)(s & 31) + 32 * hh;
    const float x = xhat[slab * (long)(PH * SLAB) + ((long)((R >> 2) * WAVE + lane)) * 4 + (R & 3)];
#pragma unroll
    for (int d = 0; d < 32; ++d) acc[d] += dh[d] * x;
    if (f < 32) db += dh[f];
  }
  float *out = part + (long)blockIdx.x * (32 * PH + 32);
#pragma unroll
  for (int d = 0; d < 32; ++d) out[d * PH + f] = acc[d];
  if (f < 32) out[32 * PH + f] = db;
}

extern "C" int harl_head_dw_rows256(const float *dhead, long M, int act_dim, const float *xhat, float *part, int n_wg,
                                    void *stream) {
  if (M <= 0 || n_wg <= 0) return 0;
  if (act_dim < 1 || act_dim > 32) return bad("harl_head_dw_rows256: head width must be in [1, 32]");
  hipLaunchKernelGGL(k_head_dw_rows256, dim3(n_wg), dim3(PH), 0, (hipStream_t)stream, dhead, M, act_dim, xhat, part, n_wg);
  return check_launch("harl_head_dw_rows256");
}

extern "C" int harl_mlp_panel_tangent(const float *xin_dot, const float *xin, long M, int KP, const float *Wp, const float *Wdp,
                                      int D, const float *bdp, const float *xprimal, const uint32_t *mask_in,
                                      const float *rstd_in, float *xout_dot, void *stream) {
  if (M <= 0) return 0;
  if (KP % 32 != 0 || KP < D || KP > 512) return bad("harl_mlp_panel_tangent: KP must be a multiple of 32, >= D and <= 512");
  if (xin_dot && (!Wp || KP != PH || D != PH)) return bad("harl_mlp_panel_tangent: a hidden layer (x_in_dot given) is 256 -> 256 and needs W'");
  const long n_slabs = n_slabs_of(M);
  const size_t shm = (size_t)3 * PANEL_IMG * sizeof(u32x4);
  allow_big_lds(k_panel<2>, shm);
  hipLaunchKernelGGL(k_panel<2>, dim3(panel_grid(n_slabs)), dim3(WG_THREADS), shm, (hipStream_t)stream, xin, KP, Wdp, D, D, bdp,
                     xout_dot, nullptr, nullptr, xprimal, mask_in, rstd_in, n_slabs, xin_dot, xin_dot ? PH : 0, Wp, PH);
  return check_launch("harl_mlp_panel_tangent");
}
