// Does VALU work of one wave issue under the bf16 MFMAs of ANOTHER wave on the same SIMD (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
// Modes (256 workgroups, one per CU; ITER outer iterations; per iteration a "matrix block" = 48 dependent-chain bf16 MFMAs
// (8 chains of 6 on 4 accumulators, as split_gemm issues them) and a "vector block" = 384 independent-ish v_fma_f32):
//   M1  4 waves (1/SIMD): matrix blocks only            V1  4 waves: vector blocks only
//   A1  4 waves: matrix block then vector block (serial in one wave -- what the update kernels do)
//   A2  8 waves (2/SIMD): each wave runs HALF the iterations of A1 (same total work)
//   S2  8 waves: waves 0-3 matrix only, waves 4-7 vector only (same total work as A1)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void matrix_block(f32x16 (&acc)[4], const u32x4 &a, const u32x4 &b) {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
      acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[s & 3], 0, 0, 0);
  }
}
__device__ __forceinline__ void vector_block(float (&v)[16], float c) {
#pragma unroll
  for (int r = 0; r < 24; ++r)
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = __builtin_fmaf(v[k], c, 0.5f);
}

// I1: ONE wave interleaves the two blocks instruction by instruction: after every MFMA (dependent chain of 6 on one
// accumulator, as in split_gemm) 8 independent v_fma_f32 -- does VALU work issue in the shadow of the wave's OWN MFMAs?
__device__ __forceinline__ void interleaved_block(f32x16 (&acc)[4], const u32x4 &a, const u32x4 &b, float (&v)[16], float c) {
#pragma unroll
  for (int s = 0; s < 8; ++s) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      acc[s & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[s & 3], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[(8 * (k & 1) + q) & 15] = __builtin_fmaf(v[(8 * (k & 1) + q) & 15], c, 0.5f);
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters, float c) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float v[16];
  for (int kx = 0; kx < 16; ++kx) v[kx] = threadIdx.x * 1e-3f + kx;
  const u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) matrix_block(acc, a, b);
    if (MODE == 1) vector_block(v, c);
    if (MODE == 2) { matrix_block(acc, a, b); __builtin_amdgcn_sched_barrier(0); vector_block(v, c); __builtin_amdgcn_sched_barrier(0); }
    if (MODE == 3) { if (wave < 4) matrix_block(acc, a, b); else vector_block(v, c); }
    if (MODE == 4) interleaved_block(acc, a, b, v, c);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int kx = 0; kx < 16; ++kx) s += v[kx];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(int threads, int iters, float *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, 0.999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 512 * 4);
  const int IT = 4000;
  const float m1 = run<0>(256, IT, out), v1 = run<1>(256, IT, out), a1 = run<2>(256, IT, out), a2 = run<2>(512, IT / 2, out),
              s2 = run<3>(512, IT, out), m2 = run<0>(512, IT / 2, out), v2 = run<1>(512, IT / 2, out), i1 = run<4>(256, IT, out),
              i2 = run<4>(512, IT / 2, out);
  printf("M1 matrix only, 1 wave/SIMD            %.3f ms  (%.1f cycles/MFMA at 2.4 GHz)\n", m1, m1 * 1e-3 * 2.4e9 / (IT * 48.0));
  printf("V1 vector only, 1 wave/SIMD            %.3f ms  (%.2f cycles/VALU at 2.4 GHz)\n", v1, v1 * 1e-3 * 2.4e9 / (IT * 384.0));
  printf("M2 matrix only, 2 waves/SIMD           %.3f ms\n", m2);
  printf("V2 vector only, 2 waves/SIMD           %.3f ms  (%.2f cycles/VALU)\n", v2, v2 * 1e-3 * 2.4e9 / (IT * 384.0));
  printf("A1 matrix then vector, 1 wave/SIMD     %.3f ms  (M1 + V1 = %.3f)\n", a1, m1 + v1);
  printf("A2 same work on 2 waves/SIMD           %.3f ms\n", a2);
  printf("S2 matrix waves + vector waves         %.3f ms  (max(M1, V1) = %.3f)\n", s2, m1 > v1 ? m1 : v1);
  printf("I1 MFMA + 8 VALU interleaved, 1 wave/SIMD %.3f ms  (same work as A1: 48 MFMA + 384 VALU per iteration)\n", i1);
  printf("I2 same on 2 waves/SIMD                 %.3f ms\n", i2);
  return 0;
}
