// wide.hip -- first layer for observations wider than 32 (32 < D <= 512: MPE share_obs 54, Humanoid 393 / 376, SMAC 128+), gfx950.
//
// The narrow kernels of mlp.hip park a slab's rows in LDS next to the whole first-layer matrix; neither fits for wide
// rows.  Here the layer is split in two streaming kernels:
//   1. harl_mlp_x0n_wide: rows X[idx] -> input LayerNorm statistics (exact two-pass, in registers) -> the normalised
//      inputs as an ATL(KP) image (KP = D rounded up to 32, zero padded).  Rows are read ONCE, coalesced (lane = column);
//      the transposition to "lane = sample" goes through two 32x32 LDS tiles per 64 columns.  x0n depends on the inputs
//      only, not on the weights: every later consumer (first-layer GEMM of each epoch / line-search step, its tangent in
//      HATRPO's Fisher-vector products, the first-layer weight gradient) streams this image instead of gathering rows.
//   2. harl_mlp_fwd_wide / harl_mlp_tangent_wide: x0n ATL(KP) -> H on the bf16 matrix pipe with the exact three-way
//      operand split (split_mfma.h).  3 * H * KP bf16 (320 KiB for 128 x 416) do not fit the LDS, so the three weight
//      images are built in a global scratch buffer by a small kernel (they stay L2 resident) and every wave streams its
//      A fragments from there, amortised over TWO slabs per wave.
// The first-layer weight gradient of wide inputs is k_dw_tr (mlp.hip) over column groups of the same x0n image.
#include <type_traits>
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

// ---------------------------------------------------------------------------------------------
// rows -> x0n ATL(KP).  One wave per slab; NC = ceil(D / 64) column chunks of 64 (lane = column).
// ---------------------------------------------------------------------------------------------
// sum over the 64 lanes on the VALU only (4 DPP adds inside each row of 16, then the 4 row sums through SGPRs); the
// result is wave-uniform.  __shfl_xor lowers to ds_bpermute here: 6 dependent LDS round trips per reduction.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm:[1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm:[2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror: every lane of a row holds the row's sum
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

constexpr int XT_LD = 36;  // floats per feature row of a transposition tile [32 features][32 samples + 4]

template <int NC>
__global__ __launch_bounds__(WG_THREADS, NC <= 3 ? 2 : 1) void k_x0n_wide(const float *__restrict__ X, long ldx,
                                                                           const int64_t *__restrict__ idx, long M, int D,
                                                                           int use_ln0, float *__restrict__ x0n,
                                                                           float *__restrict__ mu0_out,
                                                                           float *__restrict__ rstd0_out, long n_slabs, int KP) {
  __shared__ __attribute__((aligned(16))) float tiles[WAVES_PER_WG][2][32 * XT_LD];
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  float *tw = &tiles[wave][0][0];
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    // ---- the slab's 32 rows, lane = column (coalesced 256-byte runs), all loads in flight at once
    // (row indices first and branch-free: with `idx ? idx[j] : j` inside the load loop hipcc closes every row's loads
    // with s_waitcnt vmcnt(0) -- 32 serialised HBM round trips per slab)
    long rows[32];
    if (idx) {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const long j = slab * SLAB + r;
        rows[r] = idx[j < M ? j : M - 1];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const long j = slab * SLAB + r;
        rows[r] = j < M ? j : M - 1;
      }
    }
    int kc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) kc[c] = 64 * c + lane < D ? 64 * c + lane : 0;
    float v[32][NC];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const float *xr = X + rows[r] * ldx;
#pragma unroll
      for (int c = 0; c < NC; ++c) v[r][c] = xr[kc[c]];
    }
    // ---- input LayerNorm statistics per row: exact two-pass on the register image
    const float invD = 1.0f / (float)D;
    float my_mean = 0.f, my_rstd = 1.f;  // lane n < 32 keeps the statistics of sample n
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      float mean = 0.f, rstd = 1.f;
      if (use_ln0) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) s += (64 * c + lane < D) ? v[r][c] : 0.f;
        mean = wave_sum_dpp(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float d = v[r][c] - mean;
          q += (64 * c + lane < D) ? d * d : 0.f;
        }
        rstd = 1.0f / sqrtf(wave_sum_dpp(q) * invD + 1e-5f);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int k = 64 * c + lane;
        // pad columns are zero except the LAST one (k = KP - 1 > D - 1), a column of ones: the weight images are zero
        // there, and the fused first-layer weight gradient of k_bwd_dx reads db' out of it (mlp.hip, x0n_store)
        v[r][c] = k < D ? (v[r][c] - mean) * rstd : ((k == KP - 1 && D < KP) ? 1.0f : 0.f);
      }
      if (lane == r) {
        my_mean = mean;
        my_rstd = rstd;
      }
    }
    if (lane < 32) {
      mu0_out[slab * SLAB + lane] = my_mean;
      rstd0_out[slab * SLAB + lane] = my_rstd;
    }
    // ---- transposition, 64 columns (two 32-feature tiles) at a time: lane (tile h, feature i) writes its 32 samples,
    // lane (sample i, half h) reads the 16 features of each tile it owns in the ATL image
    f32x4 *op = reinterpret_cast<f32x4 *>(x0n + slab * (long)KP * SLAB) + lane;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (64 * c < KP) {
        float *tl = tw + h * (32 * XT_LD) + i * XT_LD;
#pragma unroll
        for (int r4 = 0; r4 < 8; ++r4)
          *reinterpret_cast<f32x4 *>(tl + 4 * r4) = f32x4{v[4 * r4][c], v[4 * r4 + 1][c], v[4 * r4 + 2][c], v[4 * r4 + 3][c]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private hand-off between lanes
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          if (64 * c + 32 * t2 < KP) {
            const float *ts = tw + t2 * (32 * XT_LD);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {  // piece q = 4 (2c + t2) + qq: features 8 qq + 4 h + e of this tile
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = ts[(8 * qq + 4 * h + e) * XT_LD + i];
              op[(4 * (2 * c + t2) + qq) * WAVE] = o;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();  // all lanes done reading before the next chunk overwrites the tiles
      }
    }
  }
}

// rows up to 64 wide, identity order, dense (ldx == D): the 32 rows of a slab are ONE contiguous run of 32 D floats -- copied
// to LDS with fully coalesced loads, then every lane (sample i, half h) reads its sample's D values (in-lane two-pass
// statistics, no cross-lane traffic) and writes its feature slots of the ATL(32 / 64) image.
template <int KPV>  // KPV = 32 or 64: D <= KPV
__global__ __launch_bounds__(WG_THREADS, 4) void k_x0n_contig(const float *__restrict__ X, long M, int D, int use_ln0,
                                                              float *__restrict__ x0n, float *__restrict__ mu0_out,
                                                              float *__restrict__ rstd0_out, long n_slabs) {
  __shared__ float rowsl[WAVES_PER_WG][32 * KPV + 32];
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  float *rw = &rowsl[wave][0];
  const int per = 32 * D;  // floats per slab
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    const long base = slab * SLAB * (long)D, lim = M * (long)D;
#pragma unroll
    for (int u = 0; u < KPV / 2; ++u) {  // KPV/2 x 64 = 32 x KPV
      const int e = u * 64 + lane;
      if (e < per) rw[e] = base + e < lim ? X[base + e] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const float *xr = rw + i * D;
    float mean = 0.f, rstd = 1.f;
    if (use_ln0) {
      float sm = 0.f;
      for (int k = 0; k < D; ++k) sm += xr[k];
      mean = sm / (float)D;
      float vs = 0.f;
      for (int k = 0; k < D; ++k) {
        const float d = xr[k] - mean;
        vs += d * d;
      }
      rstd = 1.0f / sqrtf(vs / (float)D + 1e-5f);
    }
    f32x4 *op = reinterpret_cast<f32x4 *>(x0n + slab * (long)KPV * SLAB) + lane;
#pragma unroll
    for (int q = 0; q < KPV / 8; ++q) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = 32 * (q >> 2) + 8 * (q & 3) + 4 * h + e;
        o[e] = f < D ? (xr[f < D ? f : 0] - mean) * rstd : ((f == KPV - 1 && D < KPV) ? 1.0f : 0.f);
      }
      op[q * WAVE] = o;
    }
    if (h == 0) {
      mu0_out[slab * SLAB + i] = mean;
      rstd0_out[slab * SLAB + i] = rstd;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();  // all lanes done reading before the next slab overwrites the rows
  }
}

// narrow rows (D <= 32): two rows per load instruction (lane = (row parity, column)), statistics per half-wave
__global__ __launch_bounds__(WG_THREADS, 2) void k_x0n_narrow(const float *__restrict__ X, long ldx,
                                                              const int64_t *__restrict__ idx, long M, int D, int use_ln0,
                                                              float *__restrict__ x0n, float *__restrict__ mu0_out,
                                                              float *__restrict__ rstd0_out, long n_slabs) {
  __shared__ __attribute__((aligned(16))) float tiles[WAVES_PER_WG][32 * XT_LD];
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int i = lane & 31, h = lane >> 5;
  float *tw = &tiles[wave][0];
  const int kc = i < D ? i : 0;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    long rows[16];
    if (idx) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long j = slab * SLAB + 2 * u + h;
        rows[u] = idx[j < M ? j : M - 1];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const long j = slab * SLAB + 2 * u + h;
        rows[u] = j < M ? j : M - 1;
      }
    }
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = X[rows[u] * ldx + kc];
    const float invD = 1.0f / (float)D;
    float my_mean = 0.f, my_rstd = 1.f;  // lane (h, i) keeps the statistics of sample 2u + h for u = i >> 1 ... see below
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float mean = 0.f, rstd = 1.f;
      if (use_ln0) {
        mean = half_reduce_sum(i < D ? v[u] : 0.f) * invD;
        const float d = v[u] - mean;
        rstd = 1.0f / sqrtf(half_reduce_sum(i < D ? d * d : 0.f) * invD + 1e-5f);
      }
      v[u] = i < D ? (v[u] - mean) * rstd : ((i == 31 && D < 32) ? 1.0f : 0.f);
      if (i == u) {  // lane (h, i = u) ends up with the statistics of sample 2u + h
        my_mean = mean;
        my_rstd = rstd;
      }
    }
    if (i < 16) {
      mu0_out[slab * SLAB + 2 * i + h] = my_mean;
      rstd0_out[slab * SLAB + 2 * i + h] = my_rstd;
    }
    // transposition: T[feature i][sample 2u + h], then lane (sample i, half h) reads its 16 features
#pragma unroll
    for (int u = 0; u < 16; ++u) tw[i * XT_LD + 2 * u + h] = v[u];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    f32x4 *op = reinterpret_cast<f32x4 *>(x0n + slab * (long)32 * SLAB) + lane;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = tw[(8 * qq + 4 * h + e) * XT_LD + i];
      op[qq * WAVE] = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 Wp[H][D] -> three bf16 images [term][tile][k-step][64 lanes] x 16 B in global memory, K zero-padded to KP
// ---------------------------------------------------------------------------------------------
// (WpB != NULL: the matrix is the column concatenation [Wp | WpB] with DA columns in Wp and D - DA in WpB -- the hidden-layer
// tangent's [W' | W'_dot])
__global__ __launch_bounds__(256) void k_split_image(const float *__restrict__ Wp, int H, int D, int KP,
                                                     u32x4 *__restrict__ img, const float *__restrict__ WpB = nullptr,
                                                     int DA = 0) {
  const int MT = H / 32, NJ = KP / 16, total = MT * NJ * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int ln = e & 63, j = (e >> 6) % NJ, t = (e >> 6) / NJ, m = 32 * t + (ln & 31), g = ln >> 5;
    unsigned p[3][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // B operand of k-step j, lane half g = x0n registers 8 j .. 8 j + 7; register R <-> feature feat_base(R) + 4 g
      const int f0 = feat_base(8 * j + 2 * c) + 4 * g, f1 = feat_base(8 * j + 2 * c + 1) + 4 * g;
      float w0, w1;
      if (WpB) {
        w0 = f0 < DA ? Wp[(long)m * DA + f0] : (f0 < D ? WpB[(long)m * (D - DA) + (f0 - DA)] : 0.f);
        w1 = f1 < DA ? Wp[(long)m * DA + f1] : (f1 < D ? WpB[(long)m * (D - DA) + (f1 - DA)] : 0.f);
      } else {
        w0 = f0 < D ? Wp[(long)m * D + f0] : 0.f;
        w1 = f1 < D ? Wp[(long)m * D + f1] : 0.f;
      }
      split3_rne(w0, w1, p[0][c], p[1][c], p[2][c]);
    }
#pragma unroll
    for (int term = 0; term < 3; ++term) img[(long)term * total + e] = u32x4{p[term][0], p[term][1], p[term][2], p[term][3]};
  }
}

// ---------------------------------------------------------------------------------------------
// x0n ATL(KP) -> H.  Each wave owns two slabs at a time (every A fragment read from L2 feeds 12 MFMAs); the fragments
// and activations of k-step j+1 are in flight while the MFMAs of k-step j run.
// MODE 0: ReLU + LayerNorm + mask epilogue; MODE 1 (tangent): the LayerNorm Jacobian (forward mode); MODE 2 (raw): the
// pre-activations z = W' x0n + b' as an ATL(HO) image (activation functions other than ReLU: csrc/elementwise.hip applies
// the activation and the LayerNorm in a separate element-wise launch, harl_amd/nets.py).
// ---------------------------------------------------------------------------------------------
// NJL > 0 (round 6): the fragments of the first NJL k-steps are staged into LDS once per workgroup (3 terms x HO/32 tiles x NJL
// KiB: 96 KiB at HO = 128, NJL = 8) and only the remaining k-steps stream from L2.  The streaming kernel asks the L2 for 16 KiB per
// wave and k-step -- 32 B per clock and CU with every CU of an XCD on the same 4 MB slice -- and stalls on it (133 us per launch
// at 204 800 rows where its issue model says 80); with half of the K = 2 H tangent GEMM's fragments (the W' half, identical
// for every Fisher-vector product) resident, that traffic halves.  Same instruction order per accumulator: bit-identical results.
// MEASURED: no gain (see wide_resident() below) -- kept opt-in for the record.
template <int HO, int MODE, int NJL = 0>
__global__ __launch_bounds__(WG_THREADS, 1) void k_fwd_wide(const float *__restrict__ x0n, const u32x4 *__restrict__ img,
                                                            const float *__restrict__ bp, float *__restrict__ xout,
                                                            uint32_t *__restrict__ mask_out, float *__restrict__ rstd_out,
                                                            const float *__restrict__ xprimal,
                                                            const uint32_t *__restrict__ mask_in,
                                                            const float *__restrict__ rstd_in, long n_slabs, int KP,
                                                            const float *__restrict__ x0n_b = nullptr, int KPA = 0) {
  // x0n_b != NULL: the input is the feature concatenation of TWO images, ATL(KPA) `x0n` and ATL(KP - KPA) `x0n_b` (slab g of
  // an ATL(KP) image is the concatenation of its 32-feature blocks, so k-steps below KPA / 16 read the first image and the
  // others the second): the hidden-layer tangent  z_dot = W' x_dot + W'_dot x_hat + b'_dot  as ONE K = 2 H GEMM
  constexpr int MT = HO / 32;
  const int NJ = KP / 16, TS = MT * NJ * 64;
  const int KA = x0n_b ? KPA : KP, NJA = KA / 16, KB = KP - KA;
  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5;
  const long n_pairs = (n_slabs + 1) / 2;
  const u32x4 *wl = img + lane;
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  const u32x4 *ll = reinterpret_cast<const u32x4 *>(lds_w) + lane;
  if constexpr (NJL > 0) {  // [term][tile][k-step < NJL][lane]
    u32x4 *li = reinterpret_cast<u32x4 *>(lds_w);
    for (int e = threadIdx.x; e < 3 * MT * NJL * 64; e += WG_THREADS) {
      const int ln = e & 63, j = (e >> 6) % NJL, t = ((e >> 6) / NJL) % MT, term = (e >> 6) / (NJL * MT);
      li[e] = img[(long)term * TS + (t * NJ + j) * 64 + ln];
    }
    __syncthreads();
  }
  for (long pair = (long)blockIdx.x * WAVES_PER_WG + wave; pair < n_pairs; pair += (long)gridDim.x * WAVES_PER_WG) {
    const long s0 = 2 * pair, s1 = s0 + 1 < n_slabs ? s0 + 1 : s0;
    const f32x4 *xp0 = reinterpret_cast<const f32x4 *>(x0n + s0 * (long)KA * SLAB) + lane;
    const f32x4 *xp1 = reinterpret_cast<const f32x4 *>(x0n + s1 * (long)KA * SLAB) + lane;
    const f32x4 *xq0 = x0n_b ? reinterpret_cast<const f32x4 *>(x0n_b + s0 * (long)KB * SLAB) + lane : xp0;
    const f32x4 *xq1 = x0n_b ? reinterpret_cast<const f32x4 *>(x0n_b + s1 * (long)KB * SLAB) + lane : xp1;
    f32x16 acc0[MT], acc1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[t][r] = acc1[t][r] = bp[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    u32x4 an[3][MT];
    f32x4 bn[2][2];
    auto fetch = [&](int j) {
      if (NJL > 0 && j < NJL) {  // (wave-uniform) resident k-steps
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int t = 0; t < MT; ++t) an[term][t] = ll[((term * MT + t) * NJL + j) * 64];
      } else {
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int t = 0; t < MT; ++t) an[term][t] = wl[(long)term * TS + (t * NJ + j) * 64];
      }
      const f32x4 *b0 = j < NJA ? xp0 + (2 * j) * WAVE : xq0 + (2 * (j - NJA)) * WAVE;  // (wave-uniform)
      const f32x4 *b1 = j < NJA ? xp1 + (2 * j) * WAVE : xq1 + (2 * (j - NJA)) * WAVE;
      bn[0][0] = b0[0];
      bn[0][1] = b0[WAVE];
      bn[1][0] = b1[0];
      bn[1][1] = b1[WAVE];
    };
    fetch(0);
    for (int j = 0; j < NJ; ++j) {
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
      if (j + 1 < NJ) fetch(j + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < MT; ++t) {
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
    wide_epilogue<HO, MODE>(acc0, s0, lane, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in);
    if (s1 != s0) wide_epilogue<HO, MODE>(acc1, s1, lane, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in);
  }
}

// ---------------------------------------------------------------------------------------------
// The same GEMM with the weight fragments SHARED by the four waves of the workgroup (round 5).  k_fwd_wide lets every wave
// stream its own A fragments from L2: 12 KiB of weights and 4 KiB of activations per k-step and wave for 24 MFMAs -- at 204 800
// rows (Humanoid-17x1: 561 tangent launches of this kind per update, 31 % of it) the launch moves 0.63 GB of weight fragments
// next to 0.42 GB of activations and is bound by what one CU can load (~15 B per cycle with this mix), 5x above the matrix
// pipe's time.  Here a PANEL of four k-steps (3 terms x HO/32 tiles x 4 x 1 KiB = 48 KiB at HO = 128) is copied into LDS once
// per workgroup-iteration (8 slabs) and read back as lane-linear ds_read_b128: a quarter of the weight traffic.  Two buffers,
// one barrier per panel: the next panel travels global -> registers during the current one's MFMAs and is written into the
// buffer the PREVIOUS panel used (every wave left that panel at the last barrier).  The panels of consecutive iterations form
// one cyclic sequence (the weights do not change), so the pipeline never drains.
// ---------------------------------------------------------------------------------------------
constexpr int WSH_PJ = 4;  // k-steps per panel
template <int HO, int MODE>
__global__ __launch_bounds__(WG_THREADS, 1) void k_fwd_wide_sh(const float *__restrict__ x0n, const u32x4 *__restrict__ img,
                                                               const float *__restrict__ bp, float *__restrict__ xout,
                                                               uint32_t *__restrict__ mask_out, float *__restrict__ rstd_out,
                                                               const float *__restrict__ xprimal,
                                                               const uint32_t *__restrict__ mask_in,
                                                               const float *__restrict__ rstd_in, long n_slabs, int KP,
                                                               const float *__restrict__ x0n_b = nullptr, int KPA = 0) {
  constexpr int MT = HO / 32, PAN = 3 * MT * WSH_PJ * 64, PER = PAN / WG_THREADS;  // u32x4 per panel buffer / per thread
  static_assert(PAN % WG_THREADS == 0, "panel fragments divide evenly over the threads");
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  u32x4 *wsh = reinterpret_cast<u32x4 *>(lds_w);  // [2][PAN]
  const int NJ = KP / 16, TS = MT * NJ * 64, NP = (NJ + WSH_PJ - 1) / WSH_PJ;
  const int KA = x0n_b ? KPA : KP, NJA = KA / 16, KB = KP - KA;
  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5;
  const long n_pairs = (n_slabs + 1) / 2;
  // panel pn -> registers: thread copies elements e = tid + 256 u of the panel, e = ((term * MT + t) * 4 + jj) * 64 + lane
  u32x4 wreg[PER];
  auto fetch_panel = [&](int pn) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = threadIdx.x + u * WG_THREADS, ln = e & 63, jj = (e >> 6) % WSH_PJ, tt = (e >> 6) / WSH_PJ;  // tt = term * MT + t
      const int j = pn * WSH_PJ + jj, term = tt / MT, t = tt - term * MT;
      wreg[u] = img[(long)term * TS + (t * NJ + (j < NJ ? j : NJ - 1)) * 64 + ln];
    }
  };
  auto store_panel = [&](int buf) {
#pragma unroll
    for (int u = 0; u < PER; ++u) wsh[buf * PAN + threadIdx.x + u * WG_THREADS] = wreg[u];
  };
  fetch_panel(0);
  store_panel(0);
  fetch_panel(NP > 1 ? 1 : 0);  // the panel after the first (panel 0 again when there is only one)
  __syncthreads();
  int buf = 0, pnext = NP > 1 ? 1 : 0;  // buffer of the current panel; index of the panel in flight
  for (long p0 = (long)blockIdx.x * WAVES_PER_WG; p0 < n_pairs; p0 += (long)gridDim.x * WAVES_PER_WG) {  // uniform trip count
    const bool live = p0 + wave < n_pairs;
    const long pair = live ? p0 + wave : n_pairs - 1;
    const long s0 = 2 * pair, s1 = s0 + 1 < n_slabs ? s0 + 1 : s0;
    const f32x4 *xp0 = reinterpret_cast<const f32x4 *>(x0n + s0 * (long)KA * SLAB) + lane;
    const f32x4 *xp1 = reinterpret_cast<const f32x4 *>(x0n + s1 * (long)KA * SLAB) + lane;
    const f32x4 *xq0 = x0n_b ? reinterpret_cast<const f32x4 *>(x0n_b + s0 * (long)KB * SLAB) + lane : xp0;
    const f32x4 *xq1 = x0n_b ? reinterpret_cast<const f32x4 *>(x0n_b + s1 * (long)KB * SLAB) + lane : xp1;
    f32x16 acc0[MT], acc1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[t][r] = acc1[t][r] = bp[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    f32x4 bn[2][2];
    auto fetch_act = [&](int j) {
      const f32x4 *b0 = j < NJA ? xp0 + (2 * j) * WAVE : xq0 + (2 * (j - NJA)) * WAVE;  // (wave-uniform)
      const f32x4 *b1 = j < NJA ? xp1 + (2 * j) * WAVE : xq1 + (2 * (j - NJA)) * WAVE;
      bn[0][0] = b0[0];
      bn[0][1] = b0[WAVE];
      bn[1][0] = b1[0];
      bn[1][1] = b1[WAVE];
    };
    fetch_act(0);
    for (int pn = 0; pn < NP; ++pn) {
      const u32x4 *wl = wsh + buf * PAN + lane;
#pragma unroll 1  // (unrolled, the compiler hoists all four k-steps' fragment reads and spills)
      for (int jj = 0; jj < WSH_PJ; ++jj) {
        const int j = pn * WSH_PJ + jj;
        if (j < NJ) {  // (uniform)
          u32x4 a[3][MT];
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int t = 0; t < MT; ++t) a[term][t] = wl[((term * MT + t) * WSH_PJ + jj) * 64];
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
          if (j + 1 < NJ) fetch_act(j + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < MT; ++t) {
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
      }
      // the panel in flight goes into the other buffer (last read one panel ago: every wave has passed that barrier), the one
      // after it is requested; one barrier per panel
      store_panel(buf ^ 1);
      pnext = pnext + 1 < NP ? pnext + 1 : 0;
      fetch_panel(pnext);
      __syncthreads();
      buf ^= 1;
    }
    if (live) {
      wide_epilogue<HO, MODE>(acc0, s0, lane, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in);
      if (s1 != s0) wide_epilogue<HO, MODE>(acc1, s1, lane, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused layers 1 + 2 for inputs up to 64 wide (equal widths H) FROM the cached x0n ATL(32 / 64) image: the variant of
// k_fwd_fused2 (mlp.hip) for identity row order.  No row gather, no LDS row staging, no input-LayerNorm statistics (all
// done once per buffer by the x0n kernels); both GEMMs are split_gemm() on the bf16 pipe (K = 32 / 64; K = H), x_hat_1
// stays in registers and is written only when a backward pass follows.  8 waves share one LDS copy of the six weight images.
// ---------------------------------------------------------------------------------------------
constexpr int F2X_WAVES = 8;

template <int H, int KP0>
__global__ __launch_bounds__(64 * F2X_WAVES, 2) void k_fwd_fused2x(
    const float *__restrict__ x0n, const float *__restrict__ W1p, int D, const float *__restrict__ b1p,
    const float *__restrict__ W2p, const float *__restrict__ b2p, int store1, float *__restrict__ x1out,
    uint32_t *__restrict__ mask1, float *__restrict__ rstd1, float *__restrict__ x2out, uint32_t *__restrict__ mask2,
    float *__restrict__ rstd2, long n_slabs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NTHR = 64 * F2X_WAVES, MT = H / 32, NJ1 = KP0 / 16, NJ2 = H / 16, NR = H / 2, NW = (NR + 31) / 32;
  PHASE_BEGIN();
  u32x4 *w1img = reinterpret_cast<u32x4 *>(lds);          // [3][MT][KP0/16][64]
  u32x4 *w2img = w1img + 3 * MT * NJ1 * 64;               // [3][MT][H/16][64]
  float *b1l = reinterpret_cast<float *>(w2img + 3 * MT * NJ2 * 64);
  float *b2l = b1l + H;
  for (int e = threadIdx.x; e < MT * NJ1 * 64; e += NTHR) {  // W1' [H][D] (row stride D), K zero-padded to KP0
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
  stage_split_matrix<H, H, false, NTHR>(w2img, W2p);
  for (int e = threadIdx.x; e < H; e += NTHR) {
    b1l[e] = b1p[e];
    b2l[e] = b2p[e];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = wave_id(), h = lane >> 5;
  const long slab0 = (long)blockIdx.x * F2X_WAVES + wave, slab_stride = (long)gridDim.x * F2X_WAVES;
  const u32x4 *wl1 = w1img + lane, *wl2 = w2img + lane;
  float xr[KP0 / 2];
  atl_load<KP0>(x0n, slab0 < n_slabs ? slab0 : 0, lane, xr);
  PHASE(10);
  for (long slab = slab0; slab < n_slabs; slab += slab_stride) {
    u32x4 a1[NJ1], a2[NJ1], a3[NJ1];
    split_acts<KP0 / 2>(xr, a1, a2, a3);
    atl_load<KP0>(x0n, slab + slab_stride < n_slabs ? slab + slab_stride : slab, lane, xr);  // one slab ahead
    float x1[NR];
    uint32_t bits1[NW];
    float r1;
    PHASE(0);
    {
      f32x16 acc[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = b1l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
      split_gemm<MT, NJ1>(wl1, a1, a2, a3, acc, [](int) {});
      PHASE(1);
      // ReLU + mask + LayerNorm of layer 1 (same arithmetic as relu_norm_regs in mlp.hip)
#pragma unroll
      for (int w = 0; w < NW; ++w) bits1[w] = 0u;
      float sum = 0.f;
#pragma unroll
      for (int R = 0; R < NR; ++R) {
        x1[R] = relu_push(acc[R >> 4][R & 15], bits1[R >> 5]);
        sum += x1[R];
      }
      sum = wave_sum32(sum);
      const float mean = sum * (1.0f / H);
      float vs = 0.f;
#pragma unroll
      for (int R = 0; R < NR; ++R) {
        x1[R] -= mean;
        vs += x1[R] * x1[R];
      }
      vs = wave_sum32(vs);
      r1 = 1.0f / sqrtf(vs * (1.0f / H) + 1e-5f);
#pragma unroll
      for (int R = 0; R < NR; ++R) x1[R] *= r1;
    }
    if (store1) {
      atl_store<H>(x1out, slab, lane, x1);
#pragma unroll
      for (int w = 0; w < NW; ++w) mask1[(slab * NW + w) * WAVE + lane] = bits1[w];
      if (lane < 32) rstd1[slab * SLAB + lane] = r1;
    }
    PHASE(2);
    u32x4 y1[NJ2], y2[NJ2], y3[NJ2];
    split_acts<NR>(x1, y1, y2, y3);
    PHASE(3);
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = b2l[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
    split_gemm<MT, NJ2>(wl2, y1, y2, y3, acc, [](int) {});
    PHASE(4);
    {
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
      const float mean = sum * (1.0f / H);
      float vs = 0.f;
#pragma unroll
      for (int R = 0; R < NR; ++R) {
        v[R] -= mean;
        vs += v[R] * v[R];
      }
      vs = wave_sum32(vs);
      const float rstd = 1.0f / sqrtf(vs * (1.0f / H) + 1e-5f);
#pragma unroll
      for (int R = 0; R < NR; ++R) v[R] *= rstd;
      atl_store<H>(x2out, slab, lane, v);
#pragma unroll
      for (int w = 0; w < NW; ++w) mask2[(slab * NW + w) * WAVE + lane] = bits[w];
      if (lane < 32) rstd2[slab * SLAB + lane] = rstd;
    }
    PHASE(5);
  }
  PHASE_END(0);
}

// The shared-panel kernel is OPT-IN (HARL_WIDE_SHARED=1, from 512 slabs on; read per call: A/B and the bit-for-bit comparison of the
// two kernels in the tests).  Measured on MI355X (round 5, gpurun call 7, Humanoid-17x1): 264.7 ms per update against 235.9 ms for
// the streaming kernel -- `fwd_wide` 0.220 vs 0.167 ms, `tangent_wide` 0.226 vs 0.175 ms: one barrier per 32-column panel with ONE
// workgroup per CU costs more than the per-wave fragment reads from L2 it saves (the same lesson as the one-launch backward,
// profiles/r05_bwd_fused_ab.md).  Kept for the record and for the bit-for-bit test.
constexpr size_t wide_sh_lds(int ho) { return (size_t)2 * 3 * (ho / 32) * WSH_PJ * 64 * 16; }
bool wide_shared(long n_slabs) {
  const char *e = getenv("HARL_WIDE_SHARED");
  return e && e[0] == '1' && n_slabs >= 512;
}

// Resident first-half fragments (k_fwd_wide NJL = 8) are OPT-IN (HARL_WIDE_RESIDENT=1, from 512 slabs on; A/B and the bit-for-bit
// test).  Measured on MI355X (round 6, 204 800 rows): the K = 2 H tangent GEMM 0.142 ms against 0.133 ms streaming, the 17-agent
// HATRPO update 244.3 against 241.0 ms -- the launch is not waiting for its weight fragments.  Nor for its operand splits: a
// software pipeline that issues the split of k-step j + 1 under the MFMAs of k-step j (built, bit-identical, 0.138 - 0.152 ms) and
// dealing the ragged last round as single slabs changed nothing either.  What the launch moves is 420 MB (x_dot, x_hat_in, the
// primal x_hat for the LayerNorm Jacobian, the output) = 3.2 TB/s at 0.133 ms against the 4.0 - 4.3 TB/s the one-image layer
// kernels reach: it is a streaming kernel at three quarters of the achievable rate with ONE wave per SIMD and ~32 KiB of loads in
// flight per CU, not the 2.5x-off-the-matrix-pipe kernel the issue model made of it (profiles/r06_wide_tangent_ab.md).
constexpr size_t wide_res_lds() { return (size_t)3 * 4 * 8 * 64 * 16; }
bool wide_resident(long n_slabs) {
  const char *e = getenv("HARL_WIDE_RESIDENT");
  return e && e[0] == '1' && n_slabs >= 512;
}

template <int MODE>
int launch_wide(const float *x0n, long M, int KP, const float *Wp, int D, const float *bp, int H, void *w_img, float *xout,
                uint32_t *mask_out, float *rstd_out, const float *xprimal, const uint32_t *mask_in, const float *rstd_in,
                hipStream_t s, const char *what) {
  if (M <= 0) return 0;
  if (H != 128 && H != 64) return bad("harl_mlp_*_wide: hidden width must be 64 or 128");
  if (KP % 32 != 0 || KP < D || KP > 512) return bad("harl_mlp_*_wide: KP must be a multiple of 32, >= D and <= 512");
  if (!w_img || !x0n) return bad("harl_mlp_*_wide: x0n and the weight-image scratch are required");
  const long n_slabs = n_slabs_of(M);
  const int total = (H / 32) * (KP / 16) * 64;
  hipLaunchKernelGGL(k_split_image, dim3((total + 255) / 256), dim3(256), 0, s, Wp, H, D, KP, reinterpret_cast<u32x4 *>(w_img));
  const long pairs = (n_slabs + 1) / 2, wgs = (pairs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  const int grid = (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
  if (wide_shared(n_slabs)) {  // weight fragments shared through LDS (k_fwd_wide_sh)
    if (H == 128) {
      allow_big_lds(k_fwd_wide_sh<128, MODE>, wide_sh_lds(128));
      hipLaunchKernelGGL((k_fwd_wide_sh<128, MODE>), dim3(grid), dim3(WG_THREADS), wide_sh_lds(128), s, x0n,
                         reinterpret_cast<const u32x4 *>(w_img), bp, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in, n_slabs, KP);
    } else {
      allow_big_lds(k_fwd_wide_sh<64, MODE>, wide_sh_lds(64));
      hipLaunchKernelGGL((k_fwd_wide_sh<64, MODE>), dim3(grid), dim3(WG_THREADS), wide_sh_lds(64), s, x0n,
                         reinterpret_cast<const u32x4 *>(w_img), bp, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in, n_slabs, KP);
    }
    return check_launch(what);
  }
  if (H == 128 && KP >= 256 && wide_resident(n_slabs)) {
    allow_big_lds(k_fwd_wide<128, MODE, 8>, wide_res_lds());
    hipLaunchKernelGGL((k_fwd_wide<128, MODE, 8>), dim3(grid), dim3(WG_THREADS), wide_res_lds(), s, x0n,
                       reinterpret_cast<const u32x4 *>(w_img), bp, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in, n_slabs, KP);
  } else if (H == 128)
    hipLaunchKernelGGL((k_fwd_wide<128, MODE>), dim3(grid), dim3(WG_THREADS), 0, s, x0n, reinterpret_cast<const u32x4 *>(w_img),
                       bp, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in, n_slabs, KP);
  else
    hipLaunchKernelGGL((k_fwd_wide<64, MODE>), dim3(grid), dim3(WG_THREADS), 0, s, x0n, reinterpret_cast<const u32x4 *>(w_img),
                       bp, xout, mask_out, rstd_out, xprimal, mask_in, rstd_in, n_slabs, KP);
  return check_launch(what);
}

}  // namespace

void harl::launch_split_image(const float *Wp, int H, int D, int KP, void *img, hipStream_t stream) {
  const int total = (H / 32) * (KP / 16) * 64;
  hipLaunchKernelGGL(k_split_image, dim3((total + 255) / 256), dim3(256), 0, stream, Wp, H, D, KP, reinterpret_cast<u32x4 *>(img));
}

HARL_PHASE_ACCESSOR(wide)

// Hidden-layer tangent in ONE launch:  x_out_dot = LNjac(mask . ([W' | W'_dot] [x_in_dot ; x_hat_in] + b'_dot))  -- a K = 2 HI
// GEMM over the two input images with the weight images streamed from L2 (k_fwd_wide MODE 1).  harl_mlp_tangent_hidden
// (mlp.hip) runs the two products as two launches through the output image (the two matrices do not fit the LDS together):
// 3.1 KB of traffic per sample against 2.1 KB here, and one launch less per hidden layer and Fisher-vector product.
extern "C" int harl_mlp_tangent_hidden2(const float *xin_dot, const float *xin, long M, int HI, int HO, const float *Wp,
                                        const float *Wdp, const float *bdp, void *w_img, const float *xprimal,
                                        const uint32_t *mask_in, const float *rstd_in, float *xout_dot, void *stream) {
  if (M <= 0) return 0;
  if ((HO != 128 && HO != 64) || (HI != 128 && HI != 64)) return bad("harl_mlp_tangent_hidden2: widths must be 64 or 128");
  if (!w_img || !xin_dot || !xin) return bad("harl_mlp_tangent_hidden2: inputs and the weight-image scratch are required");
  hipStream_t s = (hipStream_t)stream;
  const long n_slabs = n_slabs_of(M);
  const int KP = 2 * HI, total = (HO / 32) * (KP / 16) * 64;
  hipLaunchKernelGGL(k_split_image, dim3((total + 255) / 256), dim3(256), 0, s, Wp, HO, KP, KP, reinterpret_cast<u32x4 *>(w_img),
                     Wdp, HI);
  const long pairs = (n_slabs + 1) / 2, wgs = (pairs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  const int grid = (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
  if (wide_shared(n_slabs)) {
    if (HO == 128) {
      allow_big_lds(k_fwd_wide_sh<128, 1>, wide_sh_lds(128));
      hipLaunchKernelGGL((k_fwd_wide_sh<128, 1>), dim3(grid), dim3(WG_THREADS), wide_sh_lds(128), s, xin_dot,
                         reinterpret_cast<const u32x4 *>(w_img), bdp, xout_dot, nullptr, nullptr, xprimal, mask_in, rstd_in, n_slabs,
                         KP, xin, HI);
    } else {
      allow_big_lds(k_fwd_wide_sh<64, 1>, wide_sh_lds(64));
      hipLaunchKernelGGL((k_fwd_wide_sh<64, 1>), dim3(grid), dim3(WG_THREADS), wide_sh_lds(64), s, xin_dot,
                         reinterpret_cast<const u32x4 *>(w_img), bdp, xout_dot, nullptr, nullptr, xprimal, mask_in, rstd_in, n_slabs,
                         KP, xin, HI);
    }
    return check_launch("harl_mlp_tangent_hidden2");
  }
  if (HO == 128 && HI == 128 && wide_resident(n_slabs)) {
    allow_big_lds(k_fwd_wide<128, 1, 8>, wide_res_lds());
    hipLaunchKernelGGL((k_fwd_wide<128, 1, 8>), dim3(grid), dim3(WG_THREADS), wide_res_lds(), s, xin_dot,
                       reinterpret_cast<const u32x4 *>(w_img), bdp, xout_dot, nullptr, nullptr, xprimal, mask_in, rstd_in, n_slabs, KP,
                       xin, HI);
  } else if (HO == 128)
    hipLaunchKernelGGL((k_fwd_wide<128, 1>), dim3(grid), dim3(WG_THREADS), 0, s, xin_dot, reinterpret_cast<const u32x4 *>(w_img),
                       bdp, xout_dot, nullptr, nullptr, xprimal, mask_in, rstd_in, n_slabs, KP, xin, HI);
  else
    hipLaunchKernelGGL((k_fwd_wide<64, 1>), dim3(grid), dim3(WG_THREADS), 0, s, xin_dot, reinterpret_cast<const u32x4 *>(w_img),
                       bdp, xout_dot, nullptr, nullptr, xprimal, mask_in, rstd_in, n_slabs, KP, xin, HI);
  return check_launch("harl_mlp_tangent_hidden2");
}

extern "C" int harl_mlp_x0n_wide(const float *X, long ldx, const int64_t *idx, long M, int D, int use_ln0, float *x0n,
                                 float *mu0, float *rstd0, void *stream) {
  if (M <= 0) return 0;
  if (D < 1 || D > 512) return bad("harl_mlp_x0n_wide: 1 <= D <= 512");
  const long n_slabs = n_slabs_of(M);
  const int KP = ((D + 31) / 32) * 32, NC = (D + 63) / 64;
  const long wgs = (n_slabs + WAVES_PER_WG - 1) / WAVES_PER_WG;
  hipStream_t s = (hipStream_t)stream;
  if (D <= 32 && !idx && ldx == D) {  // (the 64-wide instantiation measured slower than k_x0n_wide<1>: 0.16 vs 0.13 ms at D = 54)
    const int grid = (int)(wgs < 1024 ? (wgs < 1 ? 1 : wgs) : 1024);
    hipLaunchKernelGGL(k_x0n_contig<32>, dim3(grid), dim3(WG_THREADS), 0, s, X, M, D, use_ln0, x0n, mu0, rstd0, n_slabs);
    return check_launch("harl_mlp_x0n_wide");
  }
  if (D <= 32) {
    const int grid = (int)(wgs < 512 ? (wgs < 1 ? 1 : wgs) : 512);
    hipLaunchKernelGGL(k_x0n_narrow, dim3(grid), dim3(WG_THREADS), 0, s, X, ldx, idx, M, D, use_ln0, x0n, mu0, rstd0, n_slabs);
    return check_launch("harl_mlp_x0n_wide");
  }
#define LX(NCv)                                                                                                         \
  {                                                                                                                     \
    const long cap = 256L * (NCv <= 3 ? 2 : 1);                                                                         \
    const int grid = (int)(wgs < cap ? (wgs < 1 ? 1 : wgs) : cap);                                                      \
    hipLaunchKernelGGL((k_x0n_wide<NCv>), dim3(grid), dim3(WG_THREADS), 0, s, X, ldx, idx, M, D, use_ln0, x0n, mu0, rstd0, \
                       n_slabs, KP);                                                                                    \
  }
  switch (NC) {
    case 1: LX(1) break;
    case 2: LX(2) break;
    case 3: LX(3) break;
    case 4: LX(4) break;
    case 5: LX(5) break;
    case 6: LX(6) break;
    case 7: LX(7) break;
    default: LX(8) break;
  }
#undef LX
  return check_launch("harl_mlp_x0n_wide");
}

extern "C" int harl_mlp_fwd_wide(const float *x0n, long M, int KP, const float *Wp, int D, const float *bp, int H,
                                 void *w_img, float *xout, uint32_t *relu_mask, float *rstd, void *stream) {
  return launch_wide<0>(x0n, M, KP, Wp, D, bp, H, w_img, xout, relu_mask, rstd, nullptr, nullptr, nullptr,
                        (hipStream_t)stream, "harl_mlp_fwd_wide");
}

extern "C" int harl_mlp_linear_wide(const float *x0n, long M, int KP, const float *Wp, int D, const float *bp, int H,
                                    void *w_img, float *zout, void *stream) {
  return launch_wide<2>(x0n, M, KP, Wp, D, bp, H, w_img, zout, nullptr, nullptr, nullptr, nullptr, nullptr,
                        (hipStream_t)stream, "harl_mlp_linear_wide");
}

extern "C" int harl_mlp_tangent_wide(const float *x0n, long M, int KP, const float *Wdp, int D, const float *bdp, int H,
                                     void *w_img, const float *x1, const uint32_t *mask1, const float *rstd1, float *x1dot,
                                     void *stream) {
  return launch_wide<1>(x0n, M, KP, Wdp, D, bdp, H, w_img, x1dot, nullptr, nullptr, x1, mask1, rstd1,
                        (hipStream_t)stream, "harl_mlp_tangent_wide");
}

extern "C" int harl_mlp_fwd_fused2x(const float *x0n, long M, const float *W1p, int D, const float *b1p, const float *W2p,
                                    const float *b2p, int H, int store1, float *x1out, uint32_t *mask1, float *rstd1,
                                    float *x2out, uint32_t *mask2, float *rstd2, void *stream) {
  if (M <= 0) return 0;
  if (D < 1 || D > 64) return bad("harl_mlp_fwd_fused2x: input width must be <= 64");
  if (H != 128 && H != 64) return bad("harl_mlp_fwd_fused2x: hidden width must be 64 or 128");
  const long n_slabs = n_slabs_of(M);
  const int kp0 = D <= 32 ? 32 : 64;
  const size_t shm = split_image_bytes(H, kp0) + split_image_bytes(H, H) + 2 * (size_t)H * sizeof(float);
  const long wgs = (n_slabs + F2X_WAVES - 1) / F2X_WAVES;
  const int grid = (int)(wgs < 256 ? (wgs < 1 ? 1 : wgs) : 256);
  hipStream_t s = (hipStream_t)stream;
#define LF(Hv, Kv)                                                                                                       \
  {                                                                                                                      \
    allow_big_lds(k_fwd_fused2x<Hv, Kv>, shm);                                                                           \
    hipLaunchKernelGGL((k_fwd_fused2x<Hv, Kv>), dim3(grid), dim3(64 * F2X_WAVES), shm, s, x0n, W1p, D, b1p, W2p, b2p, store1, \
                       x1out, mask1, rstd1, x2out, mask2, rstd2, n_slabs);                                                \
  }
  if (H == 128) {
    if (kp0 == 32) LF(128, 32) else LF(128, 64)
  } else {
    if (kp0 == 32) LF(64, 32) else LF(64, 64)
  }
#undef LF
  return check_launch("harl_mlp_fwd_fused2x");
}
