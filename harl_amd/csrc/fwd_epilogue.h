// fwd_epilogue.h -- the per-slab epilogues of the forward kernels (mlp.hip, wide.hip, trunk.hip): ReLU + mask + LayerNorm on the
// accumulator registers.  One definition each, so that a layer computed inside a fused launch (trunk.hip) is the same arithmetic,
// operation for operation, as the layer kernel it replaces.
#pragma once
#include "common.h"

namespace harl {

// ---------------------------------------------------------------------------------------------
// epilogue shared by both forward kernels: relu, relu bit-mask, LayerNorm statistics over the
// H features of each sample (in-lane + partner half), normalise, store ATL / mask / rstd.
// ---------------------------------------------------------------------------------------------
template <int HO>
__device__ __forceinline__ void relu_norm_regs(f32x16 (&acc)[HO / 32], float (&v)[HO / 2], uint32_t (&bits)[(HO / 2 + 31) / 32],
                                               float &rstd_out) {
  constexpr int NR = HO / 2;
#pragma unroll
  for (int w = 0; w < (NR + 31) / 32; ++w) bits[w] = 0u;
#pragma unroll
  for (int R = 0; R < NR; ++R) v[R] = relu_push(acc[R >> 4][R & 15], bits[R >> 5]);
  // statistics and normalisation on register pairs (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32)
  f32x2 s2v = {0.f, 0.f};
#pragma unroll
  for (int P = 0; P < NR / 2; ++P) s2v += f32x2{v[2 * P], v[2 * P + 1]};
  float sum = s2v[0] + s2v[1];
  sum = wave_sum32(sum);
  const float mean = sum * (1.0f / HO);
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
  const float rstd = 1.0f / sqrtf(__builtin_fmaf(vs, 1.0f / HO, 1e-5f));
  const f32x2 rv = {rstd, rstd};
#pragma unroll
  for (int P = 0; P < NR / 2; ++P) {
    const f32x2 o = f32x2{v[2 * P], v[2 * P + 1]} * rv;
    v[2 * P] = o[0];
    v[2 * P + 1] = o[1];
  }
  rstd_out = rstd;
}

template <int HO>
__device__ __forceinline__ void act_store(const float (&v)[HO / 2], const uint32_t (&bits)[(HO / 2 + 31) / 32], float rstd,
                                          int lane, long slab, float *__restrict__ xout,
                                          uint32_t *__restrict__ mask_out, float *__restrict__ rstd_out) {
  constexpr int NW = (HO / 2 + 31) / 32;
  atl_store<HO>(xout, slab, lane, v);
#pragma unroll
  for (int w = 0; w < NW; ++w) mask_out[(slab * NW + w) * WAVE + lane] = bits[w];
  if (lane < 32) rstd_out[slab * SLAB + lane] = rstd;
}

template <int HO>
__device__ __forceinline__ void relu_norm_store(f32x16 (&acc)[HO / 32], int lane, long slab, float *__restrict__ xout,
                                                uint32_t *__restrict__ mask_out, float *__restrict__ rstd_out) {
  float v[HO / 2];
  uint32_t bits[(HO / 2 + 31) / 32];
  float rstd;
  relu_norm_regs<HO>(acc, v, bits, rstd);
  act_store<HO>(v, bits, rstd, lane, slab, xout, mask_out, rstd_out);
}

// The wide first layer's variant of the same epilogue (wide.hip): scalar sums in register order (kept apart from
// relu_norm_regs, whose packed sums round differently -- every kernel keeps the arithmetic its goldens were recorded with).
template <int HO>
__device__ __forceinline__ void wide_relu_norm_regs(f32x16 (&acc)[HO / 32], float (&v)[HO / 2], uint32_t (&bits)[(HO / 2 + 31) / 32],
                                                    float &rstd_out) {
  constexpr int NR = HO / 2, NW = (NR + 31) / 32;
#pragma unroll
  for (int w = 0; w < NW; ++w) bits[w] = 0u;
  float sum = 0.f;
#pragma unroll
  for (int R = 0; R < NR; ++R) {
    v[R] = relu_push(acc[R >> 4][R & 15], bits[R >> 5]);
    sum += v[R];
  }
  sum = wave_sum32(sum);
  const float mean = sum * (1.0f / HO);
  // explicit fused multiply-adds, one dependent chain in register order: left as `vs += v * v` the compiler picks per call site
  // which products it contracts (a packed fma for the even elements, multiply + add for the odd ones, differently in two kernels
  // that inline this function) and the statistic moves by an ulp in ~1 % of the rows
  float vs = 0.f;
#pragma unroll
  for (int R = 0; R < NR; ++R) {
    v[R] -= mean;
    vs = __builtin_fmaf(v[R], v[R], vs);
  }
  vs = wave_sum32(vs);
  const float rstd = 1.0f / sqrtf(__builtin_fmaf(vs, 1.0f / HO, 1e-5f));
#pragma unroll
  for (int R = 0; R < NR; ++R) v[R] *= rstd;
  rstd_out = rstd;
}

// epilogue of one slab of the wide GEMMs.  MODE 0: ReLU + LayerNorm + mask; MODE 1 (tangent): the LayerNorm Jacobian (forward mode,
// the bias slot carries b'_dot):  x_dot = LNjac(mask * z_dot) with the PRIMAL x_hat / mask / rstd;  MODE 2 (raw): z as an ATL image
template <int HO, int MODE>
__device__ __forceinline__ void wide_epilogue(f32x16 (&acc)[HO / 32], long slab, int lane, float *__restrict__ xout,
                                              uint32_t *__restrict__ mask_out, float *__restrict__ rstd_out,
                                              const float *__restrict__ xprimal, const uint32_t *__restrict__ mask_in,
                                              const float *__restrict__ rstd_in) {
  constexpr int NR = HO / 2, NW = (NR + 31) / 32;
  if constexpr (MODE == 2) {
    float z[NR];
#pragma unroll
    for (int R = 0; R < NR; ++R) z[R] = acc[R >> 4][R & 15];
    atl_store<HO>(xout, slab, lane, z);
  } else if constexpr (MODE == 1) {
    float xh[NR];
    atl_load<HO>(xprimal, slab, lane, xh);
    uint32_t bits[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) bits[w] = mask_in[(slab * NW + w) * WAVE + lane];
    const float rstd = rstd_in[slab * SLAB + (lane & 31)];
    float ad[NR];
    float q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int R = 0; R < NR; ++R) {
      ad[R] = mask_pop(acc[R >> 4][R & 15], bits[R >> 5]);
      q1 += ad[R];
      q2 += ad[R] * xh[R];
    }
    q1 = wave_sum32(q1);
    q2 = wave_sum32(q2);
    q1 *= (1.0f / HO);
    q2 *= (1.0f / HO);
#pragma unroll
    for (int R = 0; R < NR; ++R) ad[R] = rstd * (ad[R] - q1 - xh[R] * q2);
    atl_store<HO>(xout, slab, lane, ad);
  } else {
    uint32_t bits[NW];
    float v[NR];
    float rstd;
    wide_relu_norm_regs<HO>(acc, v, bits, rstd);
    act_store<HO>(v, bits, rstd, lane, slab, xout, mask_out, rstd_out);
  }
}

}  // namespace harl
