// mfma_transpose.h -- transposes on the matrix pipe for the weight-gradient GEMMs (gfx950).
//
// dW[o][k] = sum_s dz[s][o] x[s][k] reduces over SAMPLES, so both operands are needed with lane = feature and the samples
// along the MFMA k index, while every producer holds them with lane = sample (accumulator layout, common.h).  Instead of a
// round trip through LDS, a split operand (three bf16 terms, split_mfma.h) is multiplied by a PERMUTED IDENTITY: as the A
// operand (M = sample) times B = identity columns selecting the features of one 32-feature block, the product lands in the
// C layout with lane = feature and 16 samples sigma(r, h) = (r&3) + 8 (r>>2) + 4 h per lane -- exact (each output is one
// bf16 value times 1.0), 6 MFMAs per 32 x 32 block of a three-term operand, no LDS, no barriers.
#pragma once
#include "split_mfma.h"

namespace harl {

constexpr unsigned BF16_ONE_LO = 0x00003F80u, BF16_ONE_HI = 0x3F800000u;

// Permuted identities for a 32-feature block of an accumulator-layout operand: k-step 2t supplies features
// 32t + (i&3) + 8(i>>2) + 4g (element i of lane half g), k-step 2t+1 the same + 16 (common.h, feat_base).
struct Ident {
  u32x4 j0, j1;
};
__device__ __forceinline__ Ident make_ident(int lane) {
  const int n = lane & 31, g = lane >> 5;
  Ident I;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int f0 = ((2 * c) & 3) + 8 * ((2 * c) >> 2) + 4 * g, f1 = f0 + 1;
    I.j0[c] = (n == f0 ? BF16_ONE_LO : 0u) | (n == f1 ? BF16_ONE_HI : 0u);
    I.j1[c] = (n == 16 + f0 ? BF16_ONE_LO : 0u) | (n == 16 + f1 ? BF16_ONE_HI : 0u);
  }
  return I;
}
// identity for an operand whose lane half g holds entries 8g .. 8g+7 of a <= 16-entry vector (head gradients)
__device__ __forceinline__ u32x4 make_ident16(int lane) {
  const int n = lane & 31, g = lane >> 5;
  u32x4 I;
#pragma unroll
  for (int c = 0; c < 4; ++c) I[c] = (n == 8 * g + 2 * c ? BF16_ONE_LO : 0u) | (n == 8 * g + 2 * c + 1 ? BF16_ONE_HI : 0u);
  return I;
}

__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {  // two exactly-bf16 floats -> packed pair (low = first)
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}

// acc (lane = feature, 16 samples) -> the two k-step operands of a weight-gradient MFMA
__device__ __forceinline__ void pack_transposed(const f32x16 &acc, u32x4 &o0, u32x4 &o1) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    o0[c] = pack_hi16(acc[2 * c], acc[2 * c + 1]);
    o1[c] = pack_hi16(acc[8 + 2 * c], acc[8 + 2 * c + 1]);
  }
}

// One 32-feature block (k-steps 2t, 2t+1 of the three split terms) -> transposed operands T[term][k-step].
// SUM: also return sum_r (t1 + t2 + t3)[r] = the lane's feature summed over its 16 samples (bias gradients).
template <bool SUM>
__device__ __forceinline__ float transpose_block(const u32x4 &x1a, const u32x4 &x1b, const u32x4 &x2a, const u32x4 &x2b,
                                                 const u32x4 &x3a, const u32x4 &x3b, const Ident &I, u32x4 (&T)[3][2]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 c1 = mfma_bf16(x1a, I.j0, zero), c2 = mfma_bf16(x2a, I.j0, zero), c3 = mfma_bf16(x3a, I.j0, zero);
  c1 = mfma_bf16(x1b, I.j1, c1);
  c2 = mfma_bf16(x2b, I.j1, c2);
  c3 = mfma_bf16(x3b, I.j1, c3);
  pack_transposed(c1, T[0][0], T[0][1]);
  pack_transposed(c2, T[1][0], T[1][1]);
  pack_transposed(c3, T[2][0], T[2][1]);
  float s = 0.f;
  if constexpr (SUM) {
    f32x2 a = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      a += f32x2{c3[2 * r], c3[2 * r + 1]};
      a += f32x2{c2[2 * r], c2[2 * r + 1]};
      a += f32x2{c1[2 * r], c1[2 * r + 1]};
    }
    s = a[0] + a[1];
  }
  return s;
}

// 16 accumulator-layout registers (one 32-feature block) -> split + transposed
template <bool SUM, bool PACKED = PK_DEFAULT>
__device__ __forceinline__ float split_transpose_block(const float *v16, const Ident &I, u32x4 (&T)[3][2]) {
  u32x4 y1[2], y2[2], y3[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned a, b, d;
      split3<PACKED>(v16[8 * j + 2 * c], v16[8 * j + 2 * c + 1], a, b, d);
      y1[j][c] = a;
      y2[j][c] = b;
      y3[j][c] = d;
    }
  return transpose_block<SUM>(y1[0], y1[1], y2[0], y2[1], y3[0], y3[1], I, T);
}

// acc += A^T-block x B^T-block over the slab's 32 samples: 2 k-steps x the six cross products (smallest first)
__device__ __forceinline__ void dw_tile(f32x16 &acc, const u32x4 (&A)[3][2], const u32x4 (&B)[3][2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    acc = mfma_bf16(A[2][ks], B[0][ks], acc);
    acc = mfma_bf16(A[0][ks], B[2][ks], acc);
    acc = mfma_bf16(A[1][ks], B[1][ks], acc);
    acc = mfma_bf16(A[1][ks], B[0][ks], acc);
    acc = mfma_bf16(A[0][ks], B[1][ks], acc);
    acc = mfma_bf16(A[0][ks], B[0][ks], acc);
  }
}

// Combine the four waves' weight-gradient accumulators acc[MT_][NT_] (+ per-lane bias sums db[MT_]) through LDS in fixed
// order and write ONE partial row  dWp[32 MT_][KP] | dbp[32 MT_]  (the layout harl_reduce_partials_multi expects);
// rows gridDim.x .. n_part_rows-1 of the arena are cleared (this launch runs at most one workgroup per CU).
template <int MT_, int NT_>
__device__ __forceinline__ void finish_partials(f32x16 (&acc)[MT_][NT_], float (&db)[MT_], float *buf,
                                                float *__restrict__ part, int n_part_rows) {
  constexpr int HO = 32 * MT_, KP = 32 * NT_, ROW = HO * KP + HO;
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31, h = lane >> 5;
  float dbt[MT_];
#pragma unroll
  for (int a = 0; a < MT_; ++a) dbt[a] = wave_sum32(db[a]);
  __syncthreads();  // buf may alias LDS that other waves were still reading
  for (int w = 0; w < WAVES_PER_WG; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < MT_; ++a) {
#pragma unroll
        for (int b = 0; b < NT_; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
            float *p = buf + o * KP + 32 * b + i;
            *p = (w == 0 ? 0.f : *p) + acc[a][b][r];
          }
        if (h == 0) {
          float *p = buf + HO * KP + 32 * a + i;
          *p = (w == 0 ? 0.f : *p) + dbt[a];
        }
      }
    }
    __syncthreads();
  }
  float *out = part + (long)blockIdx.x * ROW;
  for (int e = threadIdx.x; e < ROW; e += WG_THREADS) out[e] = buf[e];
  for (int row = blockIdx.x + gridDim.x; row < n_part_rows; row += gridDim.x) {
    float *z = part + (long)row * ROW;
    for (int e = threadIdx.x; e < ROW; e += WG_THREADS) z[e] = 0.f;
  }
}

}  // namespace harl
