// trunk.hip -- the whole 64-wide MLP trunk of a policy / critic in ONE launch per direction (round 6), gfx950.
//
// Replaces, launch for launch and operation for operation, the layer kernels behind harl/models/base/mlp.py:7-70 (+ the input half
// of nn.GRU's gates, harl/models/base/rnn.py:23-81) for networks with a wide first layer and 64-wide hidden layers -- the SMAC
// shapes (obs 128 / 216 -> [64, 64, 64] -> GRU 64):
//   forward : harl_mlp_fwd_wide + (L - 1) x harl_mlp_fwd_hidden + the gate launch inside harl_gru_fwd  ->  harl_mlp_fwd_trunk
//   backward: the dx launch inside harl_gru_bwd + (L - 1) x harl_mlp_bwd_dx                           ->  harl_mlp_bwd_trunk
// Why: at 81 920 rows per minibatch (2 560 slabs) each of those launches runs 12 - 25 us, most of it ramp-up, weight staging and
// tail; an 8-agent recurrent update issued 1 378 of them (profiles/r06_smac_timeline.md).  Here the activations of a slab stay in
// the accumulator layout from layer to layer (it IS the next GEMM's B operand, common.h), the hidden layers' split weight images
// (24 KiB each) and the gates' (72 KiB) sit in LDS together, and only what a later kernel reads is written.
//
// Every stage calls the SAME device functions in the same order as the kernel it replaces (fwd_epilogue.h, split_mfma.h,
// common.h): results are bit-identical to the layer-by-layer composition (tests/gpu_checks.py::check_trunk_fused compares them
// with torch.equal), so every golden recorded from the reference holds for both.
#include <stdlib.h>
#include "common.h"
#include "split_mfma.h"
#include "fwd_epilogue.h"
#include "../../include/harl_hip.h"

using namespace harl;

namespace {
int bad(const char *m) {
  set_error(m);
  return -2;
}

constexpr int TR_MAXH = 2;  // hidden layers behind the first one

struct TrunkFwd {
  const float *x0n;
  const u32x4 *img1;
  const float *b1;
  long n_slabs;
  int KP;
  const float *Wp[TR_MAXH];
  const float *bp[TR_MAXH];
  float *xout[TR_MAXH + 1];      // NULL: this layer's activation record is not written (forward-only passes)
  uint32_t *mask[TR_MAXH + 1];
  float *rstd[TR_MAXH + 1];
  const float *Wih, *bih, *bhh;  // GATES: the folded input matrix [3H][H] and both bias vectors of the GRU
  float *gi_r, *gi_z, *gi_n;
};

// ---------------------------------------------------------------------------------------------
// forward.  Layer 1 is k_fwd_wide's loop (wide.hip): a wave owns two slabs at a time and streams the A fragments of the global
// split image from L2, one k-step ahead.  Then, slab by slab: wide epilogue -> [split -> GEMM against the LDS images -> ReLU /
// LayerNorm epilogue] x NLH -> (GATES) split -> the 6-tile gate GEMM -> the three gate images k_gru_fwd_q reads.
// ---------------------------------------------------------------------------------------------
template <int H, int NLH, bool GATES>
__global__ __launch_bounds__(WG_THREADS, 1) void k_fwd_trunk(TrunkFwd A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = H / 32, NJ = H / 16, NR = H / 2, NW = (NR + 31) / 32;
  constexpr int IMG = 3 * MT * NJ * 64;  // u32x4 fragments of one hidden layer (three terms)
  constexpr int MTG = 3 * MT, IMGG = 3 * MTG * NJ * 64;
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  u32x4 *imgg = img + NLH * IMG;
  float *bl = reinterpret_cast<float *>(imgg + (GATES ? IMGG : 0));  // [NLH][H], then [3 H]
  float *blg = bl + NLH * H;
#pragma unroll
  for (int l = 0; l < NLH; ++l) {
    stage_split_matrix<H, H, false, WG_THREADS>(img + l * IMG, A.Wp[l]);
    for (int e = threadIdx.x; e < H; e += WG_THREADS) bl[l * H + e] = A.bp[l][e];
  }
  if constexpr (GATES) {
    stage_split_matrix<3 * H, H, false, WG_THREADS>(imgg, A.Wih);
    for (int e = threadIdx.x; e < 3 * H; e += WG_THREADS) blg[e] = A.bih[e] + (e < 2 * H ? A.bhh[e] : 0.f);  // as k_gru_gates_xs
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5;
  const int KP = A.KP, NJ1 = KP / 16, TS = MT * NJ1 * 64;
  const long n_slabs = A.n_slabs, n_pairs = (n_slabs + 1) / 2;
  const u32x4 *wl1 = A.img1 + lane;
  const float *__restrict__ b1 = A.b1;

  auto tail = [&](f32x16(&acc)[MT], long slab) {
    float v[NR];
    uint32_t bits[NW];
    float rstd;
    wide_relu_norm_regs<H>(acc, v, bits, rstd);
    if (A.xout[0]) act_store<H>(v, bits, rstd, lane, slab, A.xout[0], A.mask[0], A.rstd[0]);
#pragma unroll
    for (int l = 0; l < NLH; ++l) {
      u32x4 x1[NJ], x2[NJ], x3[NJ];
      split_acts<NR>(v, x1, x2, x3);
      f32x16 a[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[t][r] = bl[l * H + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
      split_gemm<MT, NJ>(img + l * IMG + lane, x1, x2, x3, a, [](int) {});
      relu_norm_regs<H>(a, v, bits, rstd);
      if (A.xout[l + 1]) act_store<H>(v, bits, rstd, lane, slab, A.xout[l + 1], A.mask[l + 1], A.rstd[l + 1]);
    }
    if constexpr (GATES) {
      u32x4 x1[NJ], x2[NJ], x3[NJ];
      split_acts<NR>(v, x1, x2, x3);
      f32x16 a6[MTG];
#pragma unroll
      for (int t6 = 0; t6 < MTG; ++t6)
#pragma unroll
        for (int r = 0; r < 16; ++r) a6[t6][r] = blg[32 * t6 + (r & 3) + 8 * (r >> 2) + 4 * h];
      split_gemm<MTG, NJ>(imgg + lane, x1, x2, x3, a6, [](int) {});
      float o[NR];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int R = 0; R < NR; ++R) o[R] = a6[g * MT + (R >> 4)][R & 15];
        atl_store<H>(g == 0 ? A.gi_r : (g == 1 ? A.gi_z : A.gi_n), slab, lane, o);
      }
    }
  };

  for (long pair = (long)blockIdx.x * WAVES_PER_WG + wave; pair < n_pairs; pair += (long)gridDim.x * WAVES_PER_WG) {
    const long s0 = 2 * pair, s1 = s0 + 1 < n_slabs ? s0 + 1 : s0;
    const f32x4 *xp0 = reinterpret_cast<const f32x4 *>(A.x0n + s0 * (long)KP * SLAB) + lane;
    const f32x4 *xp1 = reinterpret_cast<const f32x4 *>(A.x0n + s1 * (long)KP * SLAB) + lane;
    f32x16 acc0[MT], acc1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[t][r] = acc1[t][r] = b1[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    u32x4 an[3][MT];
    f32x4 bn[2][2];
    auto fetch = [&](int j) {
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int t = 0; t < MT; ++t) an[term][t] = wl1[(long)term * TS + (t * NJ1 + j) * 64];
      const f32x4 *q0 = xp0 + (2 * j) * WAVE, *q1 = xp1 + (2 * j) * WAVE;
      bn[0][0] = q0[0];
      bn[0][1] = q0[WAVE];
      bn[1][0] = q1[0];
      bn[1][1] = q1[WAVE];
    };
    fetch(0);
    for (int j = 0; j < NJ1; ++j) {
      u32x4 a[3][MT];
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int t = 0; t < MT; ++t) a[term][t] = an[term][t];
      u32x4 b[2][3];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 &src = bn[sl][c >> 1];
          unsigned p1, p2, p3;
          split3(src[2 * (c & 1)], src[2 * (c & 1) + 1], p1, p2, p3);
          b[sl][0][c] = p1;
          b[sl][1][c] = p2;
          b[sl][2][c] = p3;
        }
      if (j + 1 < NJ1) fetch(j + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < MT; ++t) {  // the six cross products, smallest first, as k_fwd_wide issues them
        acc0[t] = mfma_bf16(a[2][t], b[0][0], acc0[t]);
        acc1[t] = mfma_bf16(a[2][t], b[1][0], acc1[t]);
        acc0[t] = mfma_bf16(a[0][t], b[0][2], acc0[t]);
        acc1[t] = mfma_bf16(a[0][t], b[1][2], acc1[t]);
        acc0[t] = mfma_bf16(a[1][t], b[0][1], acc0[t]);
        acc1[t] = mfma_bf16(a[1][t], b[1][1], acc1[t]);
        acc0[t] = mfma_bf16(a[1][t], b[0][0], acc0[t]);
        acc1[t] = mfma_bf16(a[1][t], b[1][0], acc1[t]);
        acc0[t] = mfma_bf16(a[0][t], b[0][1], acc0[t]);
        acc1[t] = mfma_bf16(a[0][t], b[1][1], acc1[t]);
        acc0[t] = mfma_bf16(a[0][t], b[0][0], acc0[t]);
        acc1[t] = mfma_bf16(a[0][t], b[1][0], acc1[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    tail(acc0, s0);
    if (s1 != s0) tail(acc1, s1);
  }
}

// ---------------------------------------------------------------------------------------------
// backward.  Per slab: (GATES) d gates -> W_ih'^T GEMM -> LayerNorm / ReLU backward of the last MLP layer (k_gru_dx), then for
// every hidden Linear from the top: split dz -> Wp^T GEMM -> LayerNorm / ReLU backward of the layer below (k_bwd_dx<64, 64>).
// Every dz is written (the weight-gradient launch reads them all); none is read back here.
// ---------------------------------------------------------------------------------------------
struct TrunkBwd {
  long n_slabs;
  const float *Wih, *dr, *dzg, *dn;     // GATES
  const float *dz_in;                   // !GATES: dz of the top MLP layer
  const float *Wp[TR_MAXH];             // hidden Linears from the top
  const float *xh[TR_MAXH + 1];         // activation records: [0] the top MLP layer (stage 0), [k + 1] the layer below Linear k
  const uint32_t *mask[TR_MAXH + 1];
  const float *rstd[TR_MAXH + 1];
  float *dz_out[TR_MAXH + 1];           // [0]: dz of the top layer (GATES), [k + 1]: dz of the layer below Linear k
};

template <int H, int NLH, bool GATES>
__global__ __launch_bounds__(WG_THREADS, 1) void k_bwd_trunk(TrunkBwd A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = H / 32, NJ = H / 16, NR = H / 2, NW = (NR + 31) / 32, NJG = 3 * H / 16;
  constexpr int IMG = 3 * MT * NJ * 64, IMGG = 3 * MT * NJG * 64;
  u32x4 *img = reinterpret_cast<u32x4 *>(lds);
  u32x4 *imgg = img + NLH * IMG;
#pragma unroll
  for (int k = 0; k < NLH; ++k) stage_split_matrix<H, H, true, WG_THREADS>(img + k * IMG, A.Wp[k]);
  if constexpr (GATES) stage_split_matrix<3 * H, H, true, WG_THREADS>(imgg, A.Wih);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31;
  const long n_slabs = A.n_slabs;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float cur[NR];
    if constexpr (GATES) {
      float gates[3 * NR];
      {
        float t[NR];
        atl_load<H>(A.dr, slab, lane, t);
#pragma unroll
        for (int R = 0; R < NR; ++R) gates[R] = t[R];
        atl_load<H>(A.dzg, slab, lane, t);
#pragma unroll
        for (int R = 0; R < NR; ++R) gates[NR + R] = t[R];
        atl_load<H>(A.dn, slab, lane, t);
#pragma unroll
        for (int R = 0; R < NR; ++R) gates[2 * NR + R] = t[R];
      }
      float xh[NR];
      atl_load<H>(A.xh[0], slab, lane, xh);
      const float rstd = A.rstd[0][slab * SLAB + i];
      uint32_t mb[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) mb[w] = A.mask[0][(slab * NW + w) * WAVE + lane];
      u32x4 g1[NJG], g2[NJG], g3[NJG];
      split_acts<3 * NR>(gates, g1, g2, g3);
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      split_gemm<MT, NJG>(imgg + lane, g1, g2, g3, acc, [](int) {});
      float dx[NR];
#pragma unroll
      for (int R = 0; R < NR; ++R) dx[R] = acc[R >> 4][R & 15];
      ln_bwd_relu_mbits<H>(dx, xh, mb, rstd, cur);
      atl_store<H>(A.dz_out[0], slab, lane, cur);
    } else {
      atl_load<H>(A.dz_in, slab, lane, cur);
    }
#pragma unroll
    for (int k = 0; k < NLH; ++k) {
      u32x4 g1[NJ], g2[NJ], g3[NJ];
      split_acts<NR>(cur, g1, g2, g3);
      float xh[NR];
      atl_load<H>(A.xh[k + 1], slab, lane, xh);
      const float rstd = A.rstd[k + 1][slab * SLAB + i];
      uint32_t mb[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) mb[w] = A.mask[k + 1][(slab * NW + w) * WAVE + lane];
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      split_gemm<MT, NJ>(img + k * IMG + lane, g1, g2, g3, acc, [](int) {});
      float dx[NR];
#pragma unroll
      for (int R = 0; R < NR; ++R) dx[R] = acc[R >> 4][R & 15];
      ln_bwd_relu_mbits<H>(dx, xh, mb, rstd, cur);
      atl_store<H>(A.dz_out[k + 1], slab, lane, cur);
    }
  }
}

template <int NLH, bool GATES>
void launch_fwd(const TrunkFwd &A, hipStream_t s) {
  constexpr int H = 64;
  const size_t shm = (size_t)NLH * split_image_bytes(H, H) + (GATES ? split_image_bytes(3 * H, H) : 0) +
                     ((size_t)NLH * H + 3 * H) * sizeof(float);
  allow_big_lds(k_fwd_trunk<H, NLH, GATES>, shm);
  const long pairs = (A.n_slabs + 1) / 2, wgs = (pairs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  const int grid = (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
  hipLaunchKernelGGL((k_fwd_trunk<H, NLH, GATES>), dim3(grid), dim3(WG_THREADS), shm, s, A);
}

template <int NLH, bool GATES>
void launch_bwd(const TrunkBwd &A, hipStream_t s) {
  constexpr int H = 64;
  const size_t shm = (size_t)NLH * split_image_bytes(H, H) + (GATES ? split_image_bytes(3 * H, H) : 0);
  allow_big_lds(k_bwd_trunk<H, NLH, GATES>, shm);
  hipLaunchKernelGGL((k_bwd_trunk<H, NLH, GATES>), dim3(persistent_grid(A.n_slabs, 1)), dim3(WG_THREADS), shm, s, A);
}
}  // namespace

extern "C" int harl_mlp_fwd_trunk(const float *x0n, long M, int KP, const float *W1p, int D, const float *b1p, int H, void *w_img,
                                  int n_hidden, const float *const *Wp, const float *const *bp, float *const *xout,
                                  uint32_t *const *relu_mask, float *const *rstd, const float *Wih, const float *bih,
                                  const float *bhh, float *gi_ws, void *stream) {
  if (M <= 0) return 0;
  if (H != 64) return bad("harl_mlp_fwd_trunk: hidden width must be 64");
  if (n_hidden < 1 || n_hidden > TR_MAXH) return bad("harl_mlp_fwd_trunk: one or two hidden layers behind the first one");
  if (KP % 32 != 0 || KP < D || KP > 512) return bad("harl_mlp_fwd_trunk: KP must be a multiple of 32, >= D and <= 512");
  if (!w_img || !x0n) return bad("harl_mlp_fwd_trunk: x0n and the weight-image scratch are required");
  if (!gi_ws && !xout[n_hidden]) return bad("harl_mlp_fwd_trunk: nothing to write (no gate workspace and no last-layer output)");
  if (gi_ws && !(Wih && bih && bhh)) return bad("harl_mlp_fwd_trunk: the gate product needs W_ih', b_ih and b_hh");
  hipStream_t s = (hipStream_t)stream;
  TrunkFwd A{};
  A.x0n = x0n;
  A.img1 = reinterpret_cast<const u32x4 *>(w_img);
  A.b1 = b1p;
  A.n_slabs = n_slabs_of(M);
  A.KP = KP;
  for (int l = 0; l < n_hidden; ++l) {
    A.Wp[l] = Wp[l];
    A.bp[l] = bp[l];
  }
  for (int l = 0; l <= n_hidden; ++l) {
    A.xout[l] = xout[l];
    A.mask[l] = relu_mask[l];
    A.rstd[l] = rstd[l];
    if (xout[l] && !(relu_mask[l] && rstd[l])) return bad("harl_mlp_fwd_trunk: a stored layer needs its mask and rstd arrays");
  }
  A.Wih = Wih;
  A.bih = bih;
  A.bhh = bhh;
  if (gi_ws) {
    const long Mp = A.n_slabs * SLAB;
    A.gi_r = gi_ws;
    A.gi_z = gi_ws + Mp * H;
    A.gi_n = gi_ws + 2 * Mp * H;
  }
  launch_split_image(W1p, H, D, KP, w_img, s);
  if (n_hidden == 1) {
    if (gi_ws) launch_fwd<1, true>(A, s); else launch_fwd<1, false>(A, s);
  } else {
    if (gi_ws) launch_fwd<2, true>(A, s); else launch_fwd<2, false>(A, s);
  }
  return check_launch("harl_mlp_fwd_trunk");
}

extern "C" int harl_mlp_bwd_trunk(long M, int H, int n_hidden, const float *Wih, const float *dr, const float *dzg, const float *dn,
                                  const float *dz_in, const float *const *Wp, const float *const *xh,
                                  const uint32_t *const *relu_mask, const float *const *rstd, float *const *dz_out, void *stream) {
  if (M <= 0) return 0;
  if (H != 64) return bad("harl_mlp_bwd_trunk: hidden width must be 64");
  if (n_hidden < 1 || n_hidden > TR_MAXH) return bad("harl_mlp_bwd_trunk: one or two hidden Linears");
  const bool gates = Wih != nullptr;
  if (gates && !(dr && dzg && dn && dz_out[0] && xh[0] && relu_mask[0] && rstd[0]))
    return bad("harl_mlp_bwd_trunk: the gate stage needs dr, dz, dn, the top layer's record and dz_out[0]");
  if (!gates && !dz_in) return bad("harl_mlp_bwd_trunk: dz of the top layer is required without the gate stage");
  TrunkBwd A{};
  A.n_slabs = n_slabs_of(M);
  A.Wih = Wih;
  A.dr = dr;
  A.dzg = dzg;
  A.dn = dn;
  A.dz_in = dz_in;
  for (int k = 0; k < n_hidden; ++k) A.Wp[k] = Wp[k];
  for (int k = 0; k <= n_hidden; ++k) {
    A.xh[k] = xh[k];
    A.mask[k] = relu_mask[k];
    A.rstd[k] = rstd[k];
    A.dz_out[k] = dz_out[k];
    if (k > 0 && !(xh[k] && relu_mask[k] && rstd[k] && dz_out[k])) return bad("harl_mlp_bwd_trunk: missing layer record");
  }
  hipStream_t s = (hipStream_t)stream;
  if (n_hidden == 1) {
    if (gates) launch_bwd<1, true>(A, s); else launch_bwd<1, false>(A, s);
  } else {
    if (gates) launch_bwd<2, true>(A, s); else launch_bwd<2, false>(A, s);
  }
  return check_launch("harl_mlp_bwd_trunk");
}
