// update.hip -- the optimiser step of a two-hidden-layer MLP actor / critic with narrow inputs (D <= 64), with as many
// activations as possible kept on chip (gfx950).
//
//   k_upd_fwd : x0n -> layer 1 -> layer 2 -> head -> loss -> head backward -> LayerNorm/ReLU backward -> dz_2   (+ head dW)
//               HYBRID step (default): also writes layer 1's activation record (x_hat_1, ReLU mask, 1/sigma); the backward is
//               the layer kernels' (harl_mlp_bwd_dx + harl_mlp_dw_partials, mlp.hip).  x_hat_2, its mask and statistic never
//               cross HBM: ~3.5 KB per sample and update against ~4.5 KB for the layer-by-layer step, one launch less.
//               KP0 = 0 instantiations: the LAST hidden layer + head of deeper networks (input = x_hat_{L-1} from HBM).
//               !TRAIN instantiations: log-prob / factor-product / value passes (forward only).
//   k_upd_dw2 : x0n, dz_2 -> x_hat_1 (recomputed) -> dW_2' = dz_2^T x_hat_1, db_2'                       } HARL_FUSED_UPDATE=1:
//   k_upd_dx  : x0n, dz_2 -> x_hat_1 (recomputed), dx_hat_1 = W_2'^T dz_2 -> dz_1 -> dW_1' = dz_1^T x0n, db_1' } the fully fused step
//
// The fully fused step replaces, for this network shape, harl_mlp_fwd_fused2x + harl_actor_head_loss / harl_critic_head_loss +
// harl_mlp_dw_partials(hidden) + harl_mlp_bwd_dx (reference: autograd through MLPBase + ACTLayer / v_out,
// harl/algorithms/actors/happo.py:28-102, harl/algorithms/critics/v_critic.py:116-157).  HBM traffic per sample and
// optimiser step: x0n 3 x 128 B (256 B for D > 32) + the loss row inputs in, dz_2 512 B out and 2 x 512 B in -- about
// 1.9 KB -- with x_hat_1 recomputed from the 128-byte normalised-input image (48 MFMAs) instead of being stored.  Measured on
// MI355X its backward half is SLOWER than the layer kernels (more instructions for fewer bytes), so the
// default keeps only the forward half.
//
// Transposes on the matrix pipe.  A weight gradient contracts over SAMPLES, but activations live as "lane = sample"
// (common.h).  Multiplying a split operand (as the A operand, M = sample) by a permuted identity (B) yields the block in the
// C layout: lane = feature, 16 samples per lane -- exactly, because every bf16 term times 1.0 is exact and each output
// receives a single non-zero product.  Both operands of a weight-gradient GEMM are transposed the same way, so their
// sample order (sigma(r, h) = (r & 3) + 8 (r >> 2) + 4 h) agrees and the k order of a dot product is free.  Per 32-feature
// block: 6 MFMAs + 24 v_perm, no LDS, no barrier, and the weight-gradient accumulators stay wave-private (they are combined
// once, in fixed order, at the end of the kernel).
#include <type_traits>
#include "common.h"
#include "split_mfma.h"
#include "heads_common.h"
#include "mfma_transpose.h"
#include "../../include/harl_hip.h"

using namespace harl;

namespace {

int bad(const char *m) {
  set_error(m);
  return -2;
}

// ---- ReLU (+ bit mask, MSB-first as in common.h) + LayerNorm over the H features of a sample, from the accumulators.
// The SAME routine serves the forward pass and both recomputations, so the recomputed x_hat_1 / mask are bit-identical
// to what the forward pass saw (same staging, same GEMM order, same epilogue).
template <int H, bool MASK>
__device__ __forceinline__ void relu_ln(const f32x16 (&acc)[H / 32], float (&v)[H / 2], uint32_t (&bits)[(H / 2 + 31) / 32],
                                        float &rstd_out) {
  constexpr int NR = H / 2;
#pragma unroll
  for (int w = 0; w < (NR + 31) / 32; ++w) bits[w] = 0u;
#pragma unroll
  for (int R = 0; R < NR; ++R) {
    if constexpr (MASK) v[R] = relu_push(acc[R >> 4][R & 15], bits[R >> 5]);
    else v[R] = relu_plain(acc[R >> 4][R & 15]);
  }
  f32x2 s2v = {0.f, 0.f};
#pragma unroll
  for (int P = 0; P < NR / 2; ++P) s2v += f32x2{v[2 * P], v[2 * P + 1]};
  float sum = s2v[0] + s2v[1];
  sum = wave_sum32(sum);
  const float mean = sum * (1.0f / H);
  const f32x2 mv = {mean, mean};
  f32x2 vsv = {0.f, 0.f};
#pragma unroll
  for (int P = 0; P < NR / 2; ++P) {
    const f32x2 d = f32x2{v[2 * P], v[2 * P + 1]} - mv;
    vsv = fma2(d, d, vsv);
    v[2 * P] = d[0];
    v[2 * P + 1] = d[1];
  }
  float vs = vsv[0] + vsv[1];
  vs = wave_sum32(vs);
  const float rstd = 1.0f / sqrtf(vs * (1.0f / H) + 1e-5f);
  const f32x2 rv = {rstd, rstd};
#pragma unroll
  for (int P = 0; P < NR / 2; ++P) {
    const f32x2 o = f32x2{v[2 * P], v[2 * P + 1]} * rv;
    v[2 * P] = o[0];
    v[2 * P + 1] = o[1];
  }
  rstd_out = rstd;
}

// backward of x_hat = norm(relu(z)) with the ReLU mask in registers (cf. ln_bwd_relu_regs in common.h)
template <int H>
__device__ __forceinline__ void ln_bwd_relu_bits(const float (&dx)[H / 2], const float (&xh)[H / 2],
                                                 const uint32_t (&bits_in)[(H / 2 + 31) / 32], float rstd,
                                                 float (&out)[H / 2]) {
  constexpr int NR = H / 2, NW = (NR + 31) / 32;
  f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
  for (int P = 0; P < NR / 2; ++P) {
    const f32x2 d = {dx[2 * P], dx[2 * P + 1]}, x = {xh[2 * P], xh[2 * P + 1]};
    a1 += d;
    a2 += d * x;
  }
  float s1 = a1[0] + a1[1], s2 = a2[0] + a2[1];
  s1 = wave_sum32(s1);
  s2 = wave_sum32(s2);
  const float c1 = -(s1 * (1.0f / H)) * rstd, c2 = -(s2 * (1.0f / H)) * rstd;
  const f32x2 c1v = {c1, c1}, c2v = {c2, c2}, rv = {rstd, rstd};
  uint32_t bits[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) bits[w] = bits_in[w];
#pragma unroll
  for (int P = 0; P < NR / 2; ++P) {
    const f32x2 d = {dx[2 * P], dx[2 * P + 1]}, x = {xh[2 * P], xh[2 * P + 1]};
    const f32x2 da = fma2(x, c2v, fma2(d, rv, c1v));
    out[2 * P] = mask_pop(da[0], bits[(2 * P) >> 5]);
    out[2 * P + 1] = mask_pop(da[1], bits[(2 * P + 1) >> 5]);
  }
}

// three bf16 images of W1' [H][D] (row stride D), K zero-padded to KP0 (the staging of k_fwd_fused2x, wide.hip)
template <int H, int KP0, int NTHR>
__device__ __forceinline__ void stage_w1_images(u32x4 *__restrict__ w1img, const float *__restrict__ W1p, int D) {
  constexpr int MT = H / 32, NJ1 = KP0 / 16;
  for (int e = threadIdx.x; e < MT * NJ1 * 64; e += NTHR) {
    const int ln = e & 63, j = (e >> 6) % NJ1, t = (e >> 6) / NJ1, m = 32 * t + (ln & 31), g = ln >> 5;
    unsigned p[3][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int f0 = feat_base(8 * j + 2 * c) + 4 * g, f1 = feat_base(8 * j + 2 * c + 1) + 4 * g;
      split3_rne(f0 < D ? W1p[(long)m * D + f0] : 0.f, f1 < D ? W1p[(long)m * D + f1] : 0.f, p[0][c], p[1][c], p[2][c]);
    }
#pragma unroll
    for (int term = 0; term < 3; ++term) w1img[term * (MT * NJ1 * 64) + e] = u32x4{p[term][0], p[term][1], p[term][2], p[term][3]};
  }
}

// =============================================================================================
// k_upd_dw2:  dW_2'[o][k] = sum_s dz_2[s][o] x_hat_1[s][k],  db_2'[o] = sum_s dz_2[s][o],  x_hat_1 recomputed from x0n.
// All 16 output tiles live in the wave's accumulators (256 registers: one wave per SIMD with the whole 512-entry file);
// per slab 48 (layer-1 recompute) + 48 (transposes) + 192 (gradient) MFMAs.  The transposed x_hat_1 operands of the slab
// (24 KiB per wave) are parked in a wave-private LDS area between their production and the gradient MFMAs -- every lane
// reads back exactly what it wrote (lane-linear ds_write_b128 / ds_read_b128, no barrier) -- which keeps the rest of the
// slab's state inside the 256 architectural VGPRs next to the 256 accumulator registers.
// =============================================================================================
template <int H, int KP0>
__global__ __launch_bounds__(WG_THREADS, 1) void k_upd_dw2(const float *__restrict__ x0n, const float *__restrict__ dz2,
                                                           const float *__restrict__ W1p, int D,
                                                           const float *__restrict__ b1p, long n_slabs,
                                                           float *__restrict__ part, int n_part_rows) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = H / 32, NJ1 = KP0 / 16, NR = H / 2, NW = (NR + 31) / 32;
  u32x4 *w1img = reinterpret_cast<u32x4 *>(lds);
  float *b1l = reinterpret_cast<float *>(w1img + 3 * MT * NJ1 * 64);
  u32x4 *park = reinterpret_cast<u32x4 *>(b1l + H);  // [4 waves][MT blocks][3 terms][2 k-steps][64 lanes]; combine buffer at the end
  stage_w1_images<H, KP0, WG_THREADS>(w1img, W1p, D);
  for (int e = threadIdx.x; e < H; e += WG_THREADS) b1l[e] = b1p[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5;
  const long slab0 = (long)blockIdx.x * WAVES_PER_WG + wave, slab_stride = (long)gridDim.x * WAVES_PER_WG;
  const u32x4 *wl1 = w1img + lane;
  u32x4 *pk = park + (long)wave * (MT * 6 * 64) + lane;
  const Ident I = make_ident(lane);
  f32x16 acc2[MT][MT];
  float dbs[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    dbs[a] = 0.f;
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[a][b][r] = 0.f;
  }
  float xr[KP0 / 2];
  atl_load<KP0>(x0n, slab0 < n_slabs ? slab0 : 0, lane, xr);
  for (long slab = slab0; slab < n_slabs; slab += slab_stride) {
    float raw[NR];
    {
      u32x4 a1[NJ1], a2[NJ1], a3[NJ1];
      split_acts<KP0 / 2, false>(xr, a1, a2, a3);
      atl_load<H>(dz2, slab, lane, raw);  // consumed after the layer-1 recompute: > 2000 cycles to land
      atl_load<KP0>(x0n, slab + slab_stride < n_slabs ? slab + slab_stride : slab, lane, xr);  // one slab ahead
      float x1[NR];
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b1l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
      split_gemm<MT, NJ1>(wl1, a1, a2, a3, acc, [](int) {});
      uint32_t bits[NW];
      float r1;
      relu_ln<H, false>(acc, x1, bits, r1);
      // x_hat_1, block by block: split (exact), transposed, parked
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        u32x4 B[3][2];
        split_transpose_block<false, false>(&x1[16 * b], I, B);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) pk[((b * 3 + term) * 2 + ks) * 64] = B[term][ks];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // dz_2, block by block: split and transposed -> A operands, bias sums on the way
    u32x4 A[MT][3][2];
#pragma unroll
    for (int a = 0; a < MT; ++a) dbs[a] += split_transpose_block<true, false>(&raw[16 * a], I, A[a]);
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      u32x4 B[3][2];
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) B[term][ks] = pk[((b * 3 + term) * 2 + ks) * 64];
#pragma unroll
      for (int a = 0; a < MT; ++a) dw_tile(acc2[a][b], A[a], B);
    }
  }
  finish_partials<MT, MT>(acc2, dbs, reinterpret_cast<float *>(park), part, n_part_rows);
}

// =============================================================================================
// k_upd_dx:  dx_hat_1 = W_2'^T dz_2 ; x_hat_1 / mask_1 / rstd_1 recomputed from x0n ; dz_1 = LayerNorm/ReLU backward ;
// dW_1'[o][k] = sum_s dz_1[s][o] x0n[s][k], db_1'[o] = sum_s dz_1[s][o].  Nothing is written but the gradient partials.
// LDS: the three images of W_2'^T (96 KiB at H = 128) + those of W_1'.
// =============================================================================================
template <int H, int KP0>
__global__ __launch_bounds__(WG_THREADS, 1) void k_upd_dx(const float *__restrict__ x0n, const float *__restrict__ dz2,
                                                          const float *__restrict__ W1p, int D,
                                                          const float *__restrict__ b1p, const float *__restrict__ W2p,
                                                          long n_slabs, float *__restrict__ part, int n_part_rows) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int MT = H / 32, KT = KP0 / 32, NJ1 = KP0 / 16, NJ2 = H / 16, NR = H / 2, NW = (NR + 31) / 32;
  u32x4 *img2 = reinterpret_cast<u32x4 *>(lds);     // W2'^T images; reused as the combine buffer at the end
  u32x4 *w1img = img2 + 3 * MT * NJ2 * 64;
  float *b1l = reinterpret_cast<float *>(w1img + 3 * MT * NJ1 * 64);
  stage_split_matrix<H, H, true, WG_THREADS>(img2, W2p);
  stage_w1_images<H, KP0, WG_THREADS>(w1img, W1p, D);
  for (int e = threadIdx.x; e < H; e += WG_THREADS) b1l[e] = b1p[e];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5;
  const long slab0 = (long)blockIdx.x * WAVES_PER_WG + wave, slab_stride = (long)gridDim.x * WAVES_PER_WG;
  const u32x4 *wl1 = w1img + lane, *wl2 = img2 + lane;
  const Ident I = make_ident(lane);
  f32x16 acc1[MT][KT];
  float dbs[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    dbs[a] = 0.f;
#pragma unroll
    for (int n = 0; n < KT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[a][n][r] = 0.f;
  }
  float raw[NR], xr[KP0 / 2];
  atl_load<H>(dz2, slab0 < n_slabs ? slab0 : 0, lane, raw);
  atl_load<KP0>(x0n, slab0 < n_slabs ? slab0 : 0, lane, xr);
  for (long slab = slab0; slab < n_slabs; slab += slab_stride) {
    u32x4 a1[NJ1], a2[NJ1], a3[NJ1];
    split_acts<KP0 / 2>(xr, a1, a2, a3);
    float dx[NR];
    {
      u32x4 g1[NJ2], g2[NJ2], g3[NJ2];
      split_acts<NR>(raw, g1, g2, g3);
      const long nxt = slab + slab_stride < n_slabs ? slab + slab_stride : slab;  // one slab ahead
      atl_load<H>(dz2, nxt, lane, raw);
      atl_load<KP0>(x0n, nxt, lane, xr);
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
      split_gemm<MT, NJ2>(wl2, g1, g2, g3, acc, [](int) {});
#pragma unroll
      for (int R = 0; R < NR; ++R) dx[R] = acc[R >> 4][R & 15];
    }
    float dz1[NR];
    {
      float x1[NR];
      uint32_t bits1[NW];
      float r1;
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b1l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
      split_gemm<MT, NJ1>(wl1, a1, a2, a3, acc, [](int) {});
      relu_ln<H, true>(acc, x1, bits1, r1);
      ln_bwd_relu_bits<H>(dx, x1, bits1, r1, dz1);
    }
    u32x4 B[KT][3][2];
#pragma unroll
    for (int n = 0; n < KT; ++n)
      transpose_block<false>(a1[2 * n], a1[2 * n + 1], a2[2 * n], a2[2 * n + 1], a3[2 * n], a3[2 * n + 1], I, B[n]);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      u32x4 A[3][2];
      dbs[a] += split_transpose_block<true>(&dz1[16 * a], I, A);
#pragma unroll
      for (int n = 0; n < KT; ++n) dw_tile(acc1[a][n], A, B[n]);
    }
  }
  finish_partials<MT, KT>(acc1, dbs, reinterpret_cast<float *>(img2), part, n_part_rows);
}

// =============================================================================================
// k_upd_fwd:  the whole forward pass and the loss of one optimiser step (TRAIN), or a log-prob / value pass (!TRAIN).
//   TRAIN : x0n -> x_hat_1 -> x_hat_2 -> head -> ratio / clipped surrogate x factor / entropy (actor) or clipped Huber value
//           loss (critic) -> d loss / d head -> head weight gradient (transposes on the matrix pipe) -> LayerNorm / ReLU
//           backward -> dz_2 (ATL).  x_hat_1, x_hat_2, their masks and statistics never leave the registers.
//   !TRAIN: ... -> head -> log-probs (+ the sequential-update factor product) / values.
// LDS: images of W_1' and W_2' (24-48 + 96 KiB), biases, the head weights as the transposed image whlT[h][d][R]
// (heads_common.h: forward FMAs and the backward MFMA read the same one).
// =============================================================================================
struct UpdFwdArgs {
  const float *x0n, *W1p, *b1p, *W2p, *b2p;
  int D;
  long n_slabs;
  float *dz2;
  int n_part_rows;
  // hybrid optimiser step (TRAIN): also leave x_hat_1 (ATL), its ReLU mask and LayerNorm statistic in HBM, so that the
  // LAYER-BY-LAYER backward kernels (harl_mlp_bwd_dx, harl_mlp_dw_partials) run behind this launch; NULL = keep them on chip
  float *xh1;
  uint32_t *mask1;
  float *rstd1;
};

// EIGHT waves per workgroup = two per SIMD, 256 registers each: one wave's VALU / LDS / scalar work issues under the other's
// MFMAs (a lone wave issues one instruction of ANY kind per four cycles -- measured: the four-wave version of this kernel
// spent 42 % of its time issuing VALU, 31 % in s_waitcnt and 21 % blocked behind its own MFMAs, matrix pipe 28 % busy).
// To fit, the head weight gradient is accumulated in a wave-private LDS tile (HROWS rows x H floats; ds_add_f32 from the
// transient MFMA tile) instead of 64 persistent accumulator registers.
constexpr int UF_WAVES = 8, UF_THREADS = 64 * UF_WAVES;

template <int NV>
__device__ __forceinline__ void block_reduce_store8(float (&v)[NV], float *red /*[8][PS_STRIDE]*/, float *out_row) {
  const int lane = threadIdx.x & 63, wave = wave_id();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float t = wave_reduce_sum(v[k]);
    if (lane == 0) red[wave * PS_STRIDE + k] = t;
  }
  __syncthreads();
  if (threadIdx.x < PS_STRIDE) {
    float t = 0.f;
    if (threadIdx.x < NV) {
#pragma unroll
      for (int w = 0; w < UF_WAVES; ++w) t += red[w * PS_STRIDE + threadIdx.x];
    }
    out_row[threadIdx.x] = t;
  }
}

// DA = the number of head outputs (1 for the critic): one instantiation per action width, so that the per-dimension loops of
// the head and of the loss arithmetic have no run-time bounds; DAP = its padding to 4 | 8 = the layout of the head's LDS image
template <int H, int KP0, int DA, bool DISCRETE, bool TRAIN, typename ARGS>
__global__ __launch_bounds__(UF_THREADS, 2) void k_upd_fwd(UpdFwdArgs U, ARGS A) {
  constexpr bool CRITIC = std::is_same<ARGS, CriticArgs>::value;
  static_assert(DA >= 1 && DA <= 8 && (!CRITIC || DA == 1), "the LDS head-gradient tile covers 8 head outputs");
  constexpr int DAP = DA <= 4 ? 4 : 8;
  constexpr int HROWS = DA;  // rows of the head weight gradient that can be non-zero
  // KP0 == 0: the LAST hidden layer + head of a deeper network -- the input is x_hat_{L-1} as the previous layer kernel left
  // it in HBM (an ATL(H) image, U.x0n), "layer 2" is layer L, and the backward that follows is the layer kernels' from
  // dz_L on: x_hat_L, its mask and statistic never leave the chip (TRAIN only)
  constexpr bool HID = KP0 == 0;
  // (round 4, MPE actor launch: 0.353 -> 0.350 ms; the other instantiations would spill registers with the load moved up)
  constexpr bool EARLY_X0N = H == 128 && KP0 == 32 && DA <= 5 && !DISCRETE;
  // log-prob / value passes: right behind the split of this slab's inputs (the instantiations left out would spill)
  constexpr bool EARLY_X0N_LP = !TRAIN && !HID && (H == 64 || (DA != 4 && !(KP0 == 64 && DA >= 7 && !DISCRETE)));
  static_assert(!HID || TRAIN, "the last-layer variant exists for optimiser steps only");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  PHASE_BEGIN();
  constexpr int MT = H / 32, NJ1 = KP0 / 16, NJ2 = H / 16, NR = H / 2, NW = (NR + 31) / 32;
  u32x4 *w2img = reinterpret_cast<u32x4 *>(lds);   // reused as the head-gradient combine buffer at the end
  u32x4 *w1img = w2img + 3 * MT * NJ2 * 64;
  float *b1l = reinterpret_cast<float *>(w1img + 3 * MT * NJ1 * 64);
  float *b2l = b1l + H;
  float *whlT = b2l + H;                   // [2][DAE][H/2] (+ padding): the head weights, heads_common.h head_fwd_regs_t
  float *cst = whlT + 2 * head_t_block(H, DA);  // bias, sigma, logsigma, dsigma_dlogstd, rowsum, 1/sigma, 1/sigma^2
  float *red = cst + 7 * DAP;              // [8][PS_STRIDE]
  float *hacc = red + UF_WAVES * PS_STRIDE;  // [8 waves][HROWS][H] head weight gradient, wave-private
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31, h = lane >> 5;
  // slab -> (round, wave, workgroup), WAVE-major inside a round: the remainder of n_slabs over the 8 x gridDim.x waves lands on the
  // low wave indices of EVERY workgroup instead of on all waves of the low workgroups.  819 200 rows are 12.5 slabs per wave:
  // block-major, workgroups 0..127 ran 13 slabs on all eight waves and 128..255 twelve (per-workgroup loop times 227..269 us,
  // half the chip idle for the last slab: tools/phase_cycles.py --wg, round 4); now every SIMD holds one wave with 13 and one
  // with 12 (waves w and w + 4 share a SIMD), and the last slab of a SIMD runs alone, i.e. faster
  const long slab0 = (long)wave * gridDim.x + blockIdx.x, slab_stride = (long)gridDim.x * UF_WAVES;
  // ---- the first slab's inputs are requested BEFORE the weights are staged (their latency hides behind the staging) and are
  // complete when the loop starts (explicit s_waitcnt below): with loads still in flight at loop entry, the compiler's vmcnt
  // at the top of the loop is the minimum over the entry path and the back edge, and the entry path's "nothing younger than
  // the inputs" made every iteration wait for ALL stores of the previous one (ISA, round 4: s_waitcnt vmcnt(8) behind the
  // eight row loads of the loop header, i.e. until the 16 dz_2 stores had completed)
  constexpr int NXR = HID ? NR : KP0 / 2;  // the next slab's input, one slab ahead: x0n (KP0 wide) or x_hat_{L-1} (H wide)
  float xr[NXR];
  atl_load<2 * NXR>(U.x0n, slab0 < U.n_slabs ? slab0 : 0, lane, xr);
  // the gathered row index runs one slab ahead of the rows, branch-free (heads_common.h, row_index_request); slabs past the end
  // clamp to row M - 1 inside, so the request of the last iteration reads a mapped row that is never used
  constexpr bool ROWS = !CRITIC || TRAIN;  // (value passes read no per-row inputs)
  long rawn = 0;
  if constexpr (ROWS) rawn = row_index_request(A.idx, A.Whp, slab0, lane, A.M);
  stage_split_matrix<H, H, false, UF_THREADS>(w2img, U.W2p);
  if constexpr (!HID) stage_w1_images<H, KP0, UF_THREADS>(w1img, U.W1p, U.D);
  for (int e = threadIdx.x; e < H; e += UF_THREADS) {
    if constexpr (!HID) b1l[e] = U.b1p[e];
    b2l[e] = U.b2p[e];
  }
  if (TRAIN)
    for (int e = threadIdx.x; e < UF_WAVES * HROWS * H; e += UF_THREADS) hacc[e] = 0.f;
  if (threadIdx.x < WG_THREADS) {  // (the staging helpers of heads_common.h stride by WG_THREADS)
    stage_head_t<H, DAP, DA>(whlT, cst, A.Whp, A.bhp, WG_THREADS);
    if constexpr (!CRITIC) {
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
    }
  }
  __syncthreads();

  const u32x4 *wl1 = w1img + lane, *wl2 = w2img + lane;
  const float *whlT_h = whlT + h * head_t_block(H, DA);
  float *hw = hacc + wave * (HROWS * H);

  float adv_mean = 0.f, adv_den = 1.f, vmean = 0.f, vsd = 1.f;
  if constexpr (CRITIC) {
    if (TRAIN && A.vn_stats) {  // valuenorm.py:38-45
      const float d = fmaxf(A.vn_stats[2], 1e-5f);
      vmean = A.vn_stats[0] / d;
      const float msq = A.vn_stats[1] / d;
      vsd = sqrtf(fmaxf(msq - vmean * vmean, 1e-2f));
    }
  } else {
    if (TRAIN && A.adv_moments) {  // happo.py:122-127
      const double cnt = A.adv_moments[2];
      const double m = A.adv_moments[0] / cnt;
      const double var = A.adv_moments[1] / cnt - m * m;
      adv_mean = (float)m;
      adv_den = 1.0f / ((float)sqrt(var > 0 ? var : 0.0) + 1e-5f);  // reciprocal (actor_sample multiplies)
    }
  }
  adv_mean = uniform_f(adv_mean);  // uniform scalars belong in scalar registers (they were being spilled as vector registers)
  adv_den = uniform_f(adv_den);
  vmean = uniform_f(vmean);
  vsd = uniform_f(vsd);
  constexpr int NSC = CRITIC ? 8 : 8 + DAP;
  float sc[NSC];
#pragma unroll
  for (int k = 0; k < NSC; ++k) sc[k] = 0.f;
  // head bias gradient sums.  Actor: they share registers with the log-std gradient sums sc[8 + d], which only lane half 0
  // accumulates (actor_sample) -- half 1 (same samples) adds d loss / d head[d] into its copy; unpacked in the epilogue.
  // Eight accumulator registers less in a loop that sits at its 256-register budget: spilled loop-carried sums cost a
  // scratch round trip EACH per slab (in-order issue waits out the reload; phase timers, round 3)
  float dbacc[CRITIC ? DAP : 1];
#pragma unroll
  for (int d = 0; d < (CRITIC ? DAP : 1); ++d) dbacc[d] = 0.f;

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) (expcnt / lgkmcnt untouched): see the prologue loads above
  PHASE(10);
  for (long slab = slab0; slab < U.n_slabs; slab += slab_stride) {
    // ---- this slab's per-row loss inputs are requested at the TOP of its iteration (consumed behind the two GEMMs, ~15k cycles
    // later), the gathered row index of the next slab with them.  Rounds 2-3 carried them across the back edge (loaded at the
    // end of the previous iteration): the loads merge into dwordx4 tuples, the loop-carried copies do not coalesce with the
    // tuple, and the back edge became `s_waitcnt vmcnt(7); v_mov x3` -- a full drain of the memory pipeline (the 16 dz_2
    // stores of the slab included) once per slab (ISA, round 4).  Nothing is loop-carried now but the index (one register pair)
    // and the next slab's input image.  The sched_barrier keeps the scheduler from sinking the loads to their first use.
    ActorRow<DAP> rcur;
    float cvold = 0.f, cret = 0.f;
    if constexpr (ROWS) {
      const long row = row_index_resolve(A.idx, rawn, slab, lane, A.M);
      if constexpr (!CRITIC) actor_row_load_at<DAP, DISCRETE, TRAIN, DA>(A, slab, lane, row, rcur);
      else critic_row_load_at(A, row, cvold, cret);
      rawn = row_index_request(A.idx, A.Whp, slab + slab_stride, lane, A.M);
      __builtin_amdgcn_sched_barrier(0);
    }
    float v[NR];        // x_hat_2
    uint32_t bits2[NW];
    float r2;
    {
      float x1[NR];
      if constexpr (HID) {
#pragma unroll
        for (int R = 0; R < NR; ++R) x1[R] = xr[R];
        PHASE(0);
        PHASE(1);
      } else {
        u32x4 a1[NJ1], a2[NJ1], a3[NJ1];
        split_acts<KP0 / 2>(xr, a1, a2, a3);
        // log-prob / value passes: the next slab's inputs are requested as soon as this slab's are split -- the whole body
        // covers their latency (requested in the loss phase, the top of the next iteration waited ~7k cycles per slab for them:
        // phase timers, round 4, 26.6 % of the log-prob launch).  Optimiser steps hold too many registers across the GEMMs.
        if constexpr (EARLY_X0N_LP) atl_load<2 * NXR>(U.x0n, slab + slab_stride < U.n_slabs ? slab + slab_stride : slab, lane, xr);
        f32x16 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = b1l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
        PHASE(0);
        split_gemm<MT, NJ1>(wl1, a1, a2, a3, acc, [](int) {});
        PHASE(1);
        uint32_t bits1[NW];
        float r1;
        relu_ln<H, TRAIN>(acc, x1, bits1, r1);
        if constexpr (TRAIN) {
          if (U.xh1) {  // (wave-uniform) the layer kernels' activation record of layer 1, cf. k_fwd_fused2x (wide.hip)
            atl_store<H>(U.xh1, slab, lane, x1);
#pragma unroll
            for (int w = 0; w < NW; ++w) U.mask1[(slab * NW + w) * WAVE + lane] = bits1[w];
            if (lane < 32) U.rstd1[slab * SLAB + lane] = r1;
          }
        }
      }
      u32x4 y1[NJ2], y2[NJ2], y3[NJ2];
      split_acts<NR>(x1, y1, y2, y3);
      PHASE(2);
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b2l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
      split_gemm<MT, NJ2>(wl2, y1, y2, y3, acc, [](int) {});
      PHASE(3);
      relu_ln<H, TRAIN>(acc, v, bits2, r2);
    }
    PHASE(4);
    f32x4 xs[H / 8];
#pragma unroll
    for (int q = 0; q < H / 8; ++q) xs[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    float z[DAP];
    head_fwd_regs_t<H, DAP, DA>(xs, whlT_h, cst, z);  // (round 4: 0.353 -> 0.341 ms per MPE actor launch against head_fwd_regs)
    PHASE(5);
    const long sn = slab + slab_stride < U.n_slabs ? slab + slab_stride : slab;
    if constexpr (!TRAIN && !EARLY_X0N_LP) atl_load<2 * NXR>(U.x0n, sn, lane, xr);
    float dzh[DAP];
    float s1, s2;
    if constexpr (CRITIC) {
      float dv;
      if (!critic_sample<TRAIN>(A, z[0], slab, lane, vmean, vsd, sc, dv, cvold, cret)) continue;
#pragma unroll
      for (int d = 0; d < DAP; ++d) dzh[d] = d == 0 ? dv : 0.f;
      s1 = dv * cst[4 * DAP];
      s2 = dv * (z[0] - cst[0]);
    } else {
      if (!actor_sample<DAP, DISCRETE, TRAIN, DA>(A, cst, z, slab, lane, adv_mean, adv_den, sc, dzh, s1, s2, rcur)) continue;
    }
    PHASE(6);
    if constexpr (TRAIN) {
      // the next slab's normalised inputs are requested HERE, in front of the head weight gradient, where the registers allow
      // it: issued behind it they had only the head backward (~5k cycles) to land and the top of the next iteration waited for
      // them (phase timers, round 4: 8.6 % of the kernel in "rows + split x0n")
      if constexpr (EARLY_X0N) atl_load<2 * NXR>(U.x0n, sn, lane, xr);
      // ---- head weight gradient dW_head'[d][f] += sum_s dzh[s][d] x_hat_2[s][f]: both operands transposed on the matrix
      // pipe, one 32-feature tile at a time; rows d < HROWS of the tile are added to the wave's LDS accumulator
      // the permuted identities are rebuilt per slab (~40 VALU) instead of occupying 12 registers across the two GEMMs, where
      // the kernel sits at its 256-register budget; the empty asm keeps the compiler from hoisting them back out of the loop
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const Ident I = make_ident(lane_o);
      const u32x4 I16 = make_ident16(lane_o);
      u32x4 Ah[3][2];
      {
        float e8[8];  // lane half 0 carries head-gradient entries 0..7 (entries >= DAP are zero), half 1 zeros
#pragma unroll
        for (int c = 0; c < 8; ++c) e8[c] = (c < DAP && h == 0) ? dzh[c < DAP ? c : 0] : 0.f;
        u32x4 t1, t2, t3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          unsigned a, b, d;
          split3(e8[2 * c], e8[2 * c + 1], a, b, d);
          t1[c] = a;
          t2[c] = b;
          t3[c] = d;
        }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f32x16 c1 = mfma_bf16(t1, I16, zero), c2 = mfma_bf16(t2, I16, zero), c3 = mfma_bf16(t3, I16, zero);
        pack_transposed(c1, Ah[0][0], Ah[0][1]);
        pack_transposed(c2, Ah[1][0], Ah[1][1]);
        pack_transposed(c3, Ah[2][0], Ah[2][1]);
      }
#pragma unroll
      for (int n = 0; n < MT; ++n) {
        u32x4 B[3][2];
        split_transpose_block<false>(&v[16 * n], I, B);
        f32x16 tile = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        dw_tile(tile, Ah, B);
        // tile[r] of lane (i, h) = row d = (r & 3) + 8 (r >> 2) + 4 h, column 32 n + i
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // plain read-modify-write (every (row, column) of the wave's tile belongs to one lane; ds_add_f32 runs at a
          // fraction of the ds_write rate: heads_common.h, head_dw_step_lds)
          if (r < HROWS && h == 0) hw[r * H + 32 * n + i] += tile[r];
          if (4 + r < HROWS && h == 1) hw[(4 + r) * H + 32 * n + i] += tile[r];
        }
      }
      if constexpr (CRITIC) {
        if (h == 0) dbacc[0] += dzh[0];
      } else {
        if (h == 1) {
#pragma unroll
          for (int d = 0; d < DAP; ++d) sc[8 + d] += dzh[d];
        }
      }
      PHASE(7);
      if constexpr (!EARLY_X0N) atl_load<2 * NXR>(U.x0n, sn, lane, xr);
      // ---- head backward (W_head'^T dzh on the fp32 MFMA) + LayerNorm / ReLU backward -> dz_2
      head_bwd_regs_bits<H, DAP, DA, true, true>(xs, bits2[0], bits2[NW - 1], r2, slab, lane, whlT, dzh, s1, s2, U.dz2);
      PHASE(8);
    }
  }
  if constexpr (TRAIN) {
    float dbv[DAP];
#pragma unroll
    for (int d = 0; d < DAP; ++d) {
      if constexpr (CRITIC) {
        dbv[d] = dbacc[d];
      } else {
        dbv[d] = h == 1 ? sc[8 + d] : 0.f;
        sc[8 + d] = h == 0 ? sc[8 + d] : 0.f;
      }
    }
    block_reduce_store8<NSC>(sc, red, A.part_scalars + (long)blockIdx.x * PS_STRIDE);
    for (int row = blockIdx.x + gridDim.x; row < U.n_part_rows; row += gridDim.x)
      if (threadIdx.x < PS_STRIDE) A.part_scalars[(long)row * PS_STRIDE + threadIdx.x] = 0.f;
    // ---- head weight gradient: bias sums across lanes, then the eight waves' tiles in fixed order -> ONE partial row
    // dWp[32][H] | dbp[32] (rows >= HROWS are zero)
    float dbs[DAP];
#pragma unroll
    for (int d = 0; d < DAP; ++d) dbs[d] = wave_reduce_sum(dbv[d]);
    float *dbl = red;  // [8][PS_STRIDE] reused (block_reduce_store8 is done with it after the barrier below)
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < DAP; ++d) dbl[wave * PS_STRIDE + d] = dbs[d];
    }
    __syncthreads();
    float *out = A.dw_part + (long)blockIdx.x * HeadDw<H>::OUT_FLOATS;
    for (int e = threadIdx.x; e < HeadDw<H>::OUT_FLOATS; e += UF_THREADS) {
      float t = 0.f;
      if (e < 32 * H) {
        const int row = e / H;
        if (row < HROWS) {
#pragma unroll
          for (int w = 0; w < UF_WAVES; ++w) t += hacc[(w * HROWS + row) * H + (e - row * H)];
        }
      } else {
        const int d = e - 32 * H;
        if (d < DAP) {
#pragma unroll
          for (int w = 0; w < UF_WAVES; ++w) t += dbl[w * PS_STRIDE + d];
        }
      }
      out[e] = t;
    }
    for (int row = blockIdx.x + gridDim.x; row < U.n_part_rows; row += gridDim.x) {
      float *zr = A.dw_part + (long)row * HeadDw<H>::OUT_FLOATS;
      for (int e = threadIdx.x; e < HeadDw<H>::OUT_FLOATS; e += UF_THREADS) zr[e] = 0.f;
    }
  }
  PHASE(11);
  PHASE_END((TRAIN ? 0 : 2) + (CRITIC ? 1 : 0));
}

int fwd_grid(long n_slabs) {
  const long wgs = (n_slabs + UF_WAVES - 1) / UF_WAVES;
  return (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
}

int upd_grid(long n_slabs) {
  const long wgs = (n_slabs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  return (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
}

size_t fwd_lds_bytes(int H, int kp0, int dap, int hrows, int da) {
  return split_image_bytes(H, H) + split_image_bytes(H, kp0) +
         ((size_t)2 * H + (size_t)2 * head_t_block(H, da) + 7 * dap + UF_WAVES * PS_STRIDE + (size_t)UF_WAVES * hrows * H) *
             sizeof(float);
}

constexpr size_t LDS_PER_WG_MAX = 160 * 1024;  // gfx950: 160 KiB per CU

template <int H, int KP0, int DA, bool DISC, bool TRAIN, typename ARGS>
int launch_fwd(const UpdFwdArgs &U, const ARGS &A, hipStream_t s) {
  // (the wave-private head-gradient tiles exist in optimiser steps only: one row per head output)
  const size_t shm = fwd_lds_bytes(H, KP0, DA <= 4 ? 4 : 8, TRAIN ? DA : 0, DA);
  if (shm > LDS_PER_WG_MAX) {  // harl_update_supported() says so beforehand; never launch a kernel that cannot be resident
    return bad("harl_update_*: this (D, H, act_dim) does not fit the LDS of one workgroup");
  }
  allow_big_lds(k_upd_fwd<H, KP0, DA, DISC, TRAIN, ARGS>, shm);
  hipLaunchKernelGGL((k_upd_fwd<H, KP0, DA, DISC, TRAIN, ARGS>), dim3(fwd_grid(U.n_slabs)), dim3(UF_THREADS), shm, s, U, A);
  return 0;
}

// one instantiation per action width 1..8 (the per-dimension loops of the head and the loss are compile-time bounded)
template <int H, int KP0, bool TRAIN, int DA = 8>
int launch_actor_width(const UpdFwdArgs &U, const ActorArgs &A, int discrete, hipStream_t s) {
  if (A.act_dim == DA)
    return discrete ? launch_fwd<H, KP0, DA, true, TRAIN, ActorArgs>(U, A, s) : launch_fwd<H, KP0, DA, false, TRAIN, ActorArgs>(U, A, s);
  if constexpr (DA > 1) return launch_actor_width<H, KP0, TRAIN, DA - 1>(U, A, discrete, s);
  return bad("harl_update_*: act_dim must be in [1, 8]");
}

template <bool TRAIN>
int dispatch_fwd_actor(const UpdFwdArgs &U, const ActorArgs &A, int H, int discrete, hipStream_t s) {
  const int D = A.act_dim;
  if (D < 1 || D > 8) return bad("harl_update_fwd: act_dim must be in [1, 8]");
  const int kp0 = U.D <= 32 ? 32 : 64;
#define CASE(Hv, Kv)                                                                \
  if (H == Hv && kp0 == Kv) {                                                       \
    const int rc = launch_actor_width<Hv, Kv, TRAIN>(U, A, discrete, s);             \
    return rc ? rc : check_launch("harl_update_fwd");                               \
  }
  CASE(128, 32) CASE(128, 64) CASE(64, 32) CASE(64, 64)
#undef CASE
  return bad("harl_update_fwd: hidden width must be 64 or 128");
}

// last hidden layer + head of a deeper network (KP0 = 0 instantiations)
int dispatch_last_actor(const UpdFwdArgs &U, const ActorArgs &A, int H, int discrete, hipStream_t s) {
  const int D = A.act_dim;
  if (D < 1 || D > 8) return bad("harl_update_last_actor: act_dim must be in [1, 8]");
  int rc;
  if (H == 128) rc = launch_actor_width<128, 0, true>(U, A, discrete, s);
  else if (H == 64) rc = launch_actor_width<64, 0, true>(U, A, discrete, s);
  else return bad("harl_update_last_actor: hidden width must be 64 or 128");
  return rc ? rc : check_launch("harl_update_last_actor");
}

int dispatch_last_critic(const UpdFwdArgs &U, const CriticArgs &A, int H, hipStream_t s) {
  int rc;
  if (H == 128) rc = launch_fwd<128, 0, 1, false, true, CriticArgs>(U, A, s);
  else if (H == 64) rc = launch_fwd<64, 0, 1, false, true, CriticArgs>(U, A, s);
  else return bad("harl_update_last_critic: hidden width must be 64 or 128");
  return rc ? rc : check_launch("harl_update_last_critic");
}

template <bool TRAIN>
int dispatch_fwd_critic(const UpdFwdArgs &U, const CriticArgs &A, int H, hipStream_t s) {
  const int kp0 = U.D <= 32 ? 32 : 64;
#define CASE(Hv, Kv)                                                     \
  if (H == Hv && kp0 == Kv) {                                            \
    const int rc = launch_fwd<Hv, Kv, 1, false, TRAIN, CriticArgs>(U, A, s); \
    return rc ? rc : check_launch("harl_update_fwd");                        \
  }
  CASE(128, 32) CASE(128, 64) CASE(64, 32) CASE(64, 64)
#undef CASE
  return bad("harl_update_fwd: hidden width must be 64 or 128");
}

}  // namespace

HARL_PHASE_ACCESSOR(update)

extern "C" int harl_update_supported(int D, int H, int act_dim, int kind) {
  if (!(D >= 0 && D <= 64 && (H == 64 || H == 128) && act_dim >= 1 && act_dim <= 8)) return 0;
  // D = 0: the last-layer variant (harl_update_last_*).  An optimiser step also holds the wave-private head-gradient tiles
  // in LDS (one row of H floats per head output and wave): actors with 128-wide layers and 33..64 inputs fit up to 2 outputs
  const int dap = act_dim <= 4 ? 4 : 8;
  const int hrows = kind == 0 ? 0 : (kind == 2 ? 1 : act_dim);  // forward-only pass / critic step (one head output) / actor step
  return fwd_lds_bytes(H, D == 0 ? 0 : (D <= 32 ? 32 : 64), dap, hrows, kind == 2 ? 1 : act_dim) <= LDS_PER_WG_MAX ? 1 : 0;
}

extern "C" int harl_update_fwd_actor(const float *x0n, long M, int D, int H, const float *W1p, const float *b1p,
                                     const float *W2p, const float *b2p, const float *Whp, const float *bhp,
                                     const float *log_std, float std_x_coef, float std_y_coef, int discrete, int act_dim,
                                     const float *actions, const float *avail, const float *old_logp, const float *adv,
                                     const double *adv_moments, const float *factor, const float *active,
                                     double clip_param, float entropy_coef, int agg_mean, int trpo, float *logp_out,
                                     float *dz2, float *part_scalars, float *dw_part_head, int n_part_rows, float *xh1,
                                     uint32_t *rmask1, float *rstd1, void *stream) {
  if (M <= 0) return 0;
  if (D < 1 || D > 64) return bad("harl_update_fwd_actor: input width must be <= 64");
  if (!dz2 || !part_scalars || !dw_part_head || n_part_rows <= 0) return bad("harl_update_fwd_actor: missing outputs");
  if (xh1 && (!rmask1 || !rstd1)) return bad("harl_update_fwd_actor: xh1 needs rmask1 and rstd1");
  UpdFwdArgs U{x0n, W1p, b1p, W2p, b2p, D, n_slabs_of(M), dz2, n_part_rows, xh1, rmask1, rstd1};
  if (fwd_grid(U.n_slabs) > n_part_rows) return bad("harl_update_fwd_actor: n_part_rows smaller than the launch grid");
  ActorArgs A{};
  A.trpo = trpo;
  A.logp_out = logp_out;
  A.dw_part = dw_part_head;
  A.M = M; A.Whp = Whp; A.bhp = bhp; A.log_std = log_std;
  A.std_x_coef = std_x_coef; A.std_y_coef = std_y_coef; A.act_dim = act_dim;
  A.actions = actions; A.avail = avail; A.old_logp = old_logp; A.adv = adv; A.adv_moments = adv_moments;
  A.factor_in = factor; A.active = active; A.entropy_coef = entropy_coef;
  A.clip_lo = (float)(1.0 - clip_param); A.clip_hi = (float)(1.0 + clip_param);  // torch.clamp(imp, 1 - c, 1 + c): Python doubles
  A.agg_mean = agg_mean; A.part_scalars = part_scalars; A.n_slabs = U.n_slabs;
  return dispatch_fwd_actor<true>(U, A, H, discrete, (hipStream_t)stream);
}

extern "C" int harl_update_logp(const float *x0n, long M, int D, int H, const float *W1p, const float *b1p,
                                const float *W2p, const float *b2p, const float *Whp, const float *bhp,
                                const float *log_std, float std_x_coef, float std_y_coef, int discrete, int act_dim,
                                const float *actions, const float *avail, float *logp_out, const float *old_logp,
                                float *factor, int agg_mean, float *head_out, void *stream) {
  if (M <= 0) return 0;
  if (D < 1 || D > 64) return bad("harl_update_logp: input width must be <= 64");
  if (factor && !old_logp) return bad("harl_update_logp: factor update needs old_logp");
  UpdFwdArgs U{x0n, W1p, b1p, W2p, b2p, D, n_slabs_of(M), nullptr, 0, nullptr, nullptr, nullptr};
  ActorArgs A{};
  A.head_out = head_out;
  A.M = M; A.Whp = Whp; A.bhp = bhp; A.log_std = log_std;
  A.std_x_coef = std_x_coef; A.std_y_coef = std_y_coef; A.act_dim = act_dim;
  A.actions = actions; A.avail = avail; A.old_logp = old_logp; A.agg_mean = agg_mean;
  A.logp_out = logp_out; A.factor_out = factor; A.n_slabs = U.n_slabs;
  return dispatch_fwd_actor<false>(U, A, H, discrete, (hipStream_t)stream);
}

extern "C" int harl_update_fwd_critic(const float *x0n, long M, int D, int H, const float *W1p, const float *b1p,
                                      const float *W2p, const float *b2p, const float *Whp, const float *bhp,
                                      const float *value_preds, const float *returns, const float *vn_stats,
                                      float clip_param, int use_clipped, int use_huber, float huber_delta, float *dz2,
                                      float *part_scalars, float *dw_part_head, int n_part_rows, float *xh1,
                                      uint32_t *rmask1, float *rstd1, void *stream) {
  if (M <= 0) return 0;
  if (D < 1 || D > 64) return bad("harl_update_fwd_critic: input width must be <= 64");
  if (!dz2 || !part_scalars || !dw_part_head || n_part_rows <= 0) return bad("harl_update_fwd_critic: missing outputs");
  if (xh1 && (!rmask1 || !rstd1)) return bad("harl_update_fwd_critic: xh1 needs rmask1 and rstd1");
  UpdFwdArgs U{x0n, W1p, b1p, W2p, b2p, D, n_slabs_of(M), dz2, n_part_rows, xh1, rmask1, rstd1};
  if (fwd_grid(U.n_slabs) > n_part_rows) return bad("harl_update_fwd_critic: n_part_rows smaller than the launch grid");
  CriticArgs A{};
  A.M = M; A.Whp = Whp; A.bhp = bhp;
  A.value_preds = value_preds; A.returns = returns; A.vn_stats = vn_stats; A.clip_param = clip_param;
  A.huber_delta = huber_delta; A.use_clipped = use_clipped; A.use_huber = use_huber;
  A.part_scalars = part_scalars; A.n_slabs = U.n_slabs;
  A.dw_part = dw_part_head;
  return dispatch_fwd_critic<true>(U, A, H, (hipStream_t)stream);
}

extern "C" int harl_update_last_actor(const float *xin, long M, int H, const float *Wp, const float *bp, const float *Whp,
                                      const float *bhp, const float *log_std, float std_x_coef, float std_y_coef,
                                      int discrete, int act_dim, const int64_t *idx, const float *actions,
                                      const float *avail, const float *old_logp, const float *adv,
                                      const double *adv_moments, const float *factor, const float *active,
                                      double clip_param, float entropy_coef, int agg_mean, int trpo, float *logp_out,
                                      float *dz, float *part_scalars, float *dw_part_head, int n_part_rows, void *stream) {
  if (M <= 0) return 0;
  if (!xin || !dz || !part_scalars || !dw_part_head || n_part_rows <= 0) return bad("harl_update_last_actor: missing arguments");
  UpdFwdArgs U{xin, nullptr, nullptr, Wp, bp, 0, n_slabs_of(M), dz, n_part_rows, nullptr, nullptr, nullptr};
  if (fwd_grid(U.n_slabs) > n_part_rows) return bad("harl_update_last_actor: n_part_rows smaller than the launch grid");
  ActorArgs A{};
  A.trpo = trpo;
  A.logp_out = logp_out;
  A.dw_part = dw_part_head;
  A.M = M; A.Whp = Whp; A.bhp = bhp; A.log_std = log_std;
  A.std_x_coef = std_x_coef; A.std_y_coef = std_y_coef; A.act_dim = act_dim; A.idx = idx;
  A.actions = actions; A.avail = avail; A.old_logp = old_logp; A.adv = adv; A.adv_moments = adv_moments;
  A.factor_in = factor; A.active = active; A.entropy_coef = entropy_coef;
  A.clip_lo = (float)(1.0 - clip_param); A.clip_hi = (float)(1.0 + clip_param);
  A.agg_mean = agg_mean; A.part_scalars = part_scalars; A.n_slabs = U.n_slabs;
  return dispatch_last_actor(U, A, H, discrete, (hipStream_t)stream);
}

extern "C" int harl_update_last_critic(const float *xin, long M, int H, const float *Wp, const float *bp, const float *Whp,
                                       const float *bhp, const int64_t *idx, const float *value_preds, const float *returns,
                                       const float *vn_stats, float clip_param, int use_clipped, int use_huber,
                                       float huber_delta, float *dz, float *part_scalars, float *dw_part_head,
                                       int n_part_rows, void *stream) {
  if (M <= 0) return 0;
  if (!xin || !dz || !part_scalars || !dw_part_head || n_part_rows <= 0) return bad("harl_update_last_critic: missing arguments");
  UpdFwdArgs U{xin, nullptr, nullptr, Wp, bp, 0, n_slabs_of(M), dz, n_part_rows, nullptr, nullptr, nullptr};
  if (fwd_grid(U.n_slabs) > n_part_rows) return bad("harl_update_last_critic: n_part_rows smaller than the launch grid");
  CriticArgs A{};
  A.M = M; A.Whp = Whp; A.bhp = bhp; A.idx = idx;
  A.value_preds = value_preds; A.returns = returns; A.vn_stats = vn_stats; A.clip_param = clip_param;
  A.huber_delta = huber_delta; A.use_clipped = use_clipped; A.use_huber = use_huber;
  A.part_scalars = part_scalars; A.n_slabs = U.n_slabs;
  A.dw_part = dw_part_head;
  return dispatch_last_critic(U, A, H, (hipStream_t)stream);
}

extern "C" int harl_update_values(const float *x0n, long M, int D, int H, const float *W1p, const float *b1p,
                                  const float *W2p, const float *b2p, const float *Whp, const float *bhp, float *values,
                                  void *stream) {
  if (M <= 0) return 0;
  if (D < 1 || D > 64) return bad("harl_update_values: input width must be <= 64");
  UpdFwdArgs U{x0n, W1p, b1p, W2p, b2p, D, n_slabs_of(M), nullptr, 0, nullptr, nullptr, nullptr};
  CriticArgs A{};
  A.M = M; A.Whp = Whp; A.bhp = bhp; A.values_out = values; A.n_slabs = U.n_slabs;
  return dispatch_fwd_critic<false>(U, A, H, (hipStream_t)stream);
}

extern "C" int harl_update_bwd(const float *x0n, const float *dz2, long M, int D, int H, const float *W1p,
                               const float *b1p, const float *W2p, float *dw_part1, float *dw_part2, int n_part_rows,
                               void *stream) {
  if (M <= 0) return 0;
  if (D < 1 || D > 64) return bad("harl_update_bwd: input width must be <= 64");
  if (!dw_part1 || !dw_part2 || n_part_rows <= 0) return bad("harl_update_bwd: missing outputs");
  const long n_slabs = n_slabs_of(M);
  const int grid = upd_grid(n_slabs), kp0 = D <= 32 ? 32 : 64;
  if (grid > n_part_rows) return bad("harl_update_bwd: n_part_rows smaller than the launch grid");
  hipStream_t s = (hipStream_t)stream;
#define CASE(Hv, Kv)                                                                                                    \
  if (H == Hv && kp0 == Kv) {                                                                                           \
    size_t shm2 = (size_t)WAVES_PER_WG * (Hv / 32) * 6 * 64 * 16;              /* parked operands */                    \
    if (shm2 < ((size_t)Hv * Hv + Hv) * sizeof(float)) shm2 = ((size_t)Hv * Hv + Hv) * sizeof(float);                  \
    shm2 += split_image_bytes(Hv, Kv) + (size_t)Hv * sizeof(float);                                                     \
    allow_big_lds(k_upd_dw2<Hv, Kv>, shm2);                                                                             \
    hipLaunchKernelGGL((k_upd_dw2<Hv, Kv>), dim3(grid), dim3(WG_THREADS), shm2, s, x0n, dz2, W1p, D, b1p, n_slabs,      \
                       dw_part2, n_part_rows);                                                                          \
    size_t shm1 = split_image_bytes(Hv, Hv) + split_image_bytes(Hv, Kv) + (size_t)Hv * sizeof(float);                   \
    const size_t need = ((size_t)Hv * Kv + Hv) * sizeof(float);                                                         \
    if (shm1 < need) shm1 = need;                                                                                       \
    allow_big_lds(k_upd_dx<Hv, Kv>, shm1);                                                                              \
    hipLaunchKernelGGL((k_upd_dx<Hv, Kv>), dim3(grid), dim3(WG_THREADS), shm1, s, x0n, dz2, W1p, D, b1p, W2p, n_slabs,  \
                       dw_part1, n_part_rows);                                                                          \
    return check_launch("harl_update_bwd");                                                                             \
  }
  CASE(128, 32) CASE(128, 64) CASE(64, 32) CASE(64, 64)
#undef CASE
  return bad("harl_update_bwd: hidden width must be 64 or 128");
}
