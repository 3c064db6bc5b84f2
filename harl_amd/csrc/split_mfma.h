// split_mfma.h -- fp32 GEMMs on the bf16 matrix pipe of gfx950 with an EXACT three-way operand split.
//
// Every fp32 operand is decomposed into three bf16 values, x = x1 + x2 + x3 (8 + 8 + 8 significand bits: the sum is the
// fp32 value bit for bit), and the product  w·x  is evaluated as the six cross terms of weight <= 2^-16:
//     w1 x1 + (w1 x2 + w2 x1) + (w2 x2 + w1 x3 + w3 x1)
// on v_mfma_f32_32x32x16_bf16 (products exact, fp32 accumulation), smallest terms first.  The dropped terms
// (w2 x3, w3 x2, w3 x3) are <= 2^-24 relative to |w x|.  Measured against an fp64 reference on H = 128 dot products
// (tools/mfma_bf16x3.hip, profiles/r01_mfma_bf16x3.txt): max error 2.3e-7 of sum|w x| vs 3.5e-7 for the fp32 MFMA's
// k-ordered fmaf chain, rms 2.2e-8 vs 2.8e-8 -- i.e. fp32-class results at 6/16 of the fp32 MFMA's pipe time
// (bf16 MFMA = 16x the fp32 rate).  With the fp32 pipe the hidden-layer kernels sit on BOTH roofs at once (H = 128:
// 31 FLOP per HBM byte = the chip's fp32-MFMA / HBM balance); with the split they are HBM-bound.
//
// Layouts.  C/D is dtype independent (lane = sample, register R <-> feature f(R, h), common.h), so activations still chain
// from layer to layer in registers.  For the 16-wide k-step j the B operand of lane (sample, half h) is its registers
// R = 8j .. 8j+7 (features f(8j+i, h)), packed in pairs: any k order is allowed as long as A uses the same one.
// A operand (weights, LDS): lane (m, g) of tile t, k-step j holds the 8 bf16  W[row(t, m)][f(8j+i, g)], i = 0..7 -- one
// ds_read_b128 from a lane-linear image [term][t][j][64 lanes] x 16 B (conflict-free).
#pragma once
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace harl {

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4 &a, const u32x4 &b, const f32x16 &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// exact split of two floats into three packed bf16 pairs (low half = first value).  Truncation at every level: the three
// pieces are the three bytes-and-a-bit of the significand, all of the sign of x (3 VALU ops per value, 1.5 per pack).
template <bool PACKED = PK_DEFAULT>
__device__ __forceinline__ void split3(float f0, float f1, unsigned &p1, unsigned &p2, unsigned &p3) {
  // v_perm takes the high halves directly (no masking needed for the packed terms)
  const unsigned b0 = __float_as_uint(f0), b1 = __float_as_uint(f1);
  p1 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  if constexpr (PACKED) {
    // the two remainders of a pair are ONE packed subtraction (v_pk_add_f32 with neg modifiers): 9 VALU ops per pair, at
    // the price of 64-bit-aligned register pairs (more pressure: the kernels that live at the register limit opt out)
    const f32x2 r = f32x2{f0, f1} - f32x2{__uint_as_float(b0 & 0xffff0000u), __uint_as_float(b1 & 0xffff0000u)};
    const unsigned c0 = __float_as_uint(r[0]), c1 = __float_as_uint(r[1]);
    p2 = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    // the last remainder has <= 8 significant bits: exact in bf16
    const f32x2 q = r - f32x2{__uint_as_float(c0 & 0xffff0000u), __uint_as_float(c1 & 0xffff0000u)};
    p3 = __builtin_amdgcn_perm(__float_as_uint(q[1]), __float_as_uint(q[0]), 0x07060302u);
  } else {
    const float r0 = f0 - __uint_as_float(b0 & 0xffff0000u), r1 = f1 - __uint_as_float(b1 & 0xffff0000u);
    const unsigned c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
    p2 = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(c0 & 0xffff0000u), q1 = r1 - __uint_as_float(c1 & 0xffff0000u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
  }
}

// round-to-nearest variant for the weights (split once per workgroup; v_cvt_pk_bf16_f32 is slow but unbiased)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void split3_rne(float f0, float f1, unsigned &p1, unsigned &p2, unsigned &p3) {
  p1 = cvt_pk_bf16(f0, f1);
  const float r0 = f0 - __uint_as_float(p1 << 16), r1 = f1 - __uint_as_float(p1 & 0xffff0000u);
  p2 = cvt_pk_bf16(r0, r1);
  const float q0 = r0 - __uint_as_float(p2 << 16), q1 = r1 - __uint_as_float(p2 & 0xffff0000u);
  p3 = cvt_pk_bf16(q0, q1);
}

// activations: NR = H/2 accumulator-layout registers of one lane -> NR/8 k-step operands per term
template <int NR, bool PACKED = PK_DEFAULT>
__device__ __forceinline__ void split_acts(const float (&v)[NR], u32x4 (&x1)[NR / 8], u32x4 (&x2)[NR / 8], u32x4 (&x3)[NR / 8]) {
#pragma unroll
  for (int j = 0; j < NR / 8; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned a, b, d;
      split3<PACKED>(v[8 * j + 2 * c], v[8 * j + 2 * c + 1], a, b, d);
      x1[j][c] = a;
      x2[j][c] = b;
      x3[j][c] = d;
    }
}

constexpr size_t split_image_bytes(int rows, int k) { return (size_t)3 * rows * k * 2; }

// fp32 Wp[H][D] -> the three bf16 images [term][tile][k-step][64 lanes] x 16 B in GLOBAL memory, K zero-padded to KP (one small
// launch on `stream`; wide.hip).  The operand of the kernels that stream their weight fragments from L2 (k_fwd_wide, k_fwd_trunk).
void launch_split_image(const float *Wp, int H, int D, int KP, void *img, hipStream_t stream);

// Workgroup-cooperative staging of the three weight images from the fp32 row-major matrix Wp[HO][HI].
//   TRANSPOSED = false: GEMM rows = Wp rows (forward, M = HO, K = HI):  A[row][k] = Wp[row][k]
//   TRANSPOSED = true : GEMM rows = Wp columns (backward dX, M = HI, K = HO):  A[row][k] = Wp[k][row]
// image index ((term * MT + t) * NJ + j) * 64 + lane,  MT = M/32 tiles, NJ = K/16 k-steps.
template <int HO, int HI, bool TRANSPOSED, int NTHR>
__device__ __forceinline__ void stage_split_matrix(u32x4 *__restrict__ img, const float *__restrict__ Wp) {
  constexpr int M = TRANSPOSED ? HI : HO, K = TRANSPOSED ? HO : HI, MT = M / 32, NJ = K / 16;
  constexpr int TOTAL = MT * NJ * 64, PER = (TOTAL + NTHR - 1) / NTHR;
  // all of a thread's weights are fetched before the first split: one L2 latency for the whole prologue instead of one
  // per fragment (the prologue is paid by every workgroup of every launch)
  float w[PER][8];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = threadIdx.x + u * NTHR, ec = e < TOTAL ? e : 0;
    const int ln = ec & 63, j = (ec >> 6) % NJ, t = (ec >> 6) / NJ, m = 32 * t + (ln & 31), g = ln >> 5;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int k = feat_base(8 * j + c) + 4 * g;
      w[u][c] = TRANSPOSED ? Wp[(long)k * HI + m] : Wp[(long)m * HI + k];
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = threadIdx.x + u * NTHR;
    if (e < TOTAL) {
      unsigned p[3][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) split3_rne(w[u][2 * c], w[u][2 * c + 1], p[0][c], p[1][c], p[2][c]);
#pragma unroll
      for (int term = 0; term < 3; ++term) img[term * TOTAL + e] = u32x4{p[term][0], p[term][1], p[term][2], p[term][3]};
    }
  }
}

// acc[t] += A_t · B for all MT tiles and NJ k-steps.  `wl` = image + lane.  The three fragments of step s+1 are read from
// LDS while the six MFMAs of step s run; sched_barrier pins that order (hipcc otherwise hoists the reads and spills).
// PRE(s) is called once per step before the MFMAs (used by the callers to interleave their own loads).
template <int MT, int NJ, typename PRE>
__device__ __forceinline__ void split_gemm(const u32x4 *__restrict__ wl, const u32x4 (&x1)[NJ], const u32x4 (&x2)[NJ],
                                           const u32x4 (&x3)[NJ], f32x16 (&acc)[MT], PRE &&pre) {
  constexpr int TS = MT * NJ * 64;
  u32x4 wb[2][3];
#pragma unroll
  for (int term = 0; term < 3; ++term) wb[0][term] = wl[term * TS];
#pragma unroll
  for (int s = 0; s < MT * NJ; ++s) {
    const int j = s / MT, t = s % MT, cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < MT * NJ) {
      const int j1 = (s + 1) / MT, t1 = (s + 1) % MT;
#pragma unroll
      for (int term = 0; term < 3; ++term) wb[nxt][term] = wl[term * TS + (t1 * NJ + j1) * 64];
    }
    pre(s);
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = mfma_bf16(wb[cur][2], x1[j], acc[t]);
    acc[t] = mfma_bf16(wb[cur][0], x3[j], acc[t]);
    acc[t] = mfma_bf16(wb[cur][1], x2[j], acc[t]);
    acc[t] = mfma_bf16(wb[cur][1], x1[j], acc[t]);
    acc[t] = mfma_bf16(wb[cur][0], x2[j], acc[t]);
    acc[t] = mfma_bf16(wb[cur][0], x1[j], acc[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// split_gemm with a filler hook behind EVERY product MFMA (fill(s, k6), s = step, k6 = 0..5): independent VALU work of the
// caller issued in the matrix instructions' shadows (<= 5 plain VALU per MFMA hide for free: profiles/r03_mfma_valu_overlap.md).
// Same fragments, same six products in the same order as split_gemm: bit-identical accumulators.
template <int MT, int NJ, typename FILLF>
__device__ __forceinline__ void split_gemm_fill(const u32x4 *__restrict__ wl, const u32x4 (&x1)[NJ], const u32x4 (&x2)[NJ],
                                                const u32x4 (&x3)[NJ], f32x16 (&acc)[MT], FILLF &&fill) {
  constexpr int TS = MT * NJ * 64;
  u32x4 wb[2][3];
#pragma unroll
  for (int term = 0; term < 3; ++term) wb[0][term] = wl[term * TS];
#pragma unroll
  for (int s = 0; s < MT * NJ; ++s) {
    const int j = s / MT, t = s % MT, cur = s & 1, nxt = cur ^ 1;
    if (s + 1 < MT * NJ) {
      const int j1 = (s + 1) / MT, t1 = (s + 1) % MT;
#pragma unroll
      for (int term = 0; term < 3; ++term) wb[nxt][term] = wl[term * TS + (t1 * NJ + j1) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = mfma_bf16(wb[cur][2], x1[j], acc[t]);
    fill(s, 0);
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = mfma_bf16(wb[cur][0], x3[j], acc[t]);
    fill(s, 1);
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = mfma_bf16(wb[cur][1], x2[j], acc[t]);
    fill(s, 2);
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = mfma_bf16(wb[cur][1], x1[j], acc[t]);
    fill(s, 3);
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = mfma_bf16(wb[cur][0], x2[j], acc[t]);
    fill(s, 4);
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = mfma_bf16(wb[cur][0], x1[j], acc[t]);
    fill(s, 5);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace harl
