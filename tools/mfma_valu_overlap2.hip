// MFMA || VALU overlap on gfx950, second take (round 3; VERDICT r02 "next round" item 3).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/mfma_valu_overlap2.hip -o /tmp/ov2 && /tmp/ov2   (SLP would pack the
//   filler v_fma_f32 into v_pk_fma_f32, which is a different instruction class beside MFMAs)
// Round 2's test (tools/mfma_valu_overlap.hip) chained six MFMAs on ONE accumulator -- the order split_gemm issued them in
// -- and concluded that no VALU instruction issues in the shadow of an MFMA.  This one separates the variables:
//   ORDER   chain   : tile-major outer loop, the 6 products of a tile back to back on one accumulator (round-2 split_gemm)
//           rotate  : product-major outer loop, consecutive MFMAs go to 4 DIFFERENT accumulators (same 24 MFMAs per trip)
//   NV      VALU instructions (independent v_fma_f32 / v_perm_b32 mix) placed behind every MFMA: 0, 2, 4, 5, 6, 8
//   SHAPE   v_mfma_f32_32x32x16_bf16 (32 cycles) / v_mfma_f32_16x16x32_bf16 (16 cycles, NV halved to keep VALU per pipe-cycle)
//   GRID    1 workgroup (no power / clock effect) and 256 workgroups; 1 wave per SIMD, and 2 waves per SIMD (matrix-only wave +
//           VALU-only wave sharing a SIMD, and two mixed waves)
// Reported: shader cycles per MFMA from s_memtime around the loop (wave 0 of workgroup 0; clock independent) and the wall time
// of the launch (HIP events).  The instruction order is pinned with sched_group_barrier and checked in the ISA dump
// (profiles/r03_mfma_valu_overlap.md quotes the loop bodies).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__device__ __forceinline__ void valu(float (&v)[16], unsigned (&u)[8], float c, int &slot) {
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int s = (slot + q) & 15;
    if (((slot + q) & 3) == 3)
      u[s & 7] = __builtin_amdgcn_perm(u[s & 7], u[(s + 1) & 7], 0x07060302u);
    else
      v[s] = __builtin_fmaf(v[s], c, 0.5f);
  }
  slot += NV;
}

// MODE bit0: 1 = rotate accumulators, 0 = chain on one accumulator.  ROLE (2 waves/SIMD runs): 0 both, 1 matrix only, 2 VALU only
template <int ROT, int NV, int SMALL>
__device__ __forceinline__ void body(f32x16 (&acc)[4], f32x4 (&acs)[4], const u32x4 &a, const u32x4 &b, float (&v)[16],
                                     unsigned (&u)[8], float c, bool do_m, bool do_v) {
  int slot = 0;
#pragma unroll
  for (int o = 0; o < (ROT ? 6 : 4); ++o) {
#pragma unroll
    for (int i = 0; i < (ROT ? 4 : 6); ++i) {
      const int t = ROT ? i : o;
      if (do_m) {
        if (SMALL)
          acs[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acs[t], 0, 0, 0);
        else
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0);
      }
      if (do_v) valu<NV>(v, u, c, slot);
      if (do_m) __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      if (do_v && NV) __builtin_amdgcn_sched_group_barrier(0x2, NV, 0);
    }
  }
}

// ROLE 0: every wave runs MFMA + VALU.  ROLE 1: waves 0-3 MFMA only, waves 4-7 VALU only (same SIMDs).
template <int ROT, int NV, int SMALL, int ROLE>
__global__ __launch_bounds__(512, 1) void k(float *out, long long *cyc, int iters, float c) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  f32x4 acs[4];
  for (int t = 0; t < 4; ++t) {
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int r = 0; r < 4; ++r) acs[t][r] = 0.f;
  }
  float v[16];
  unsigned u[8];
  for (int kx = 0; kx < 16; ++kx) v[kx] = threadIdx.x * 1e-3f + kx;
  for (int kx = 0; kx < 8; ++kx) u[kx] = threadIdx.x * 77u + kx;
  const u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
  const bool do_m = ROLE == 0 || (ROLE == 1 && wave < 4), do_v = ROLE == 0 || ROLE == 2 || wave >= 4;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (ROLE == 0) body<ROT, NV, SMALL>(acc, acs, a, b, v, u, c, true, true);
    else if (do_m) body<ROT, NV, SMALL>(acc, acs, a, b, v, u, c, true, false);
    else body<ROT, NV, SMALL>(acc, acs, a, b, v, u, c, false, true);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) {
    for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int r = 0; r < 4; ++r) s += acs[t][r];
  }
  for (int kx = 0; kx < 16; ++kx) s += v[kx];
  for (int kx = 0; kx < 8; ++kx) s += (float)u[kx];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

static float *g_out;
static long long *g_cyc;

template <int ROT, int NV, int SMALL, int ROLE>
void run(const char *name, int grid, int threads, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ROT, NV, SMALL, ROLE>), dim3(grid), dim3(threads), 0, 0, g_out, g_cyc, iters, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ROT, NV, SMALL, ROLE>), dim3(grid), dim3(threads), 0, 0, g_out, g_cyc, iters, 0.999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c[8];
  hipMemcpy(c, g_cyc, sizeof(c), hipMemcpyDeviceToHost);
  const double n_mfma = 24.0 * iters;
  const int last = threads / 64 - 1;
  printf("%-44s grid %3d x %d waves | %7.3f ms | s_memtime cycles per MFMA slot: wave0 %6.2f  wave%d %6.2f | %5.2f GHz-equivalent\n", name,
         grid, threads / 64, ms, c[0] / n_mfma, last, c[last] / n_mfma, c[0] / (ms * 1e6));
}

#define BOTH(ROT, NV, SMALL, ROLE, name, threads)                 \
  run<ROT, NV, SMALL, ROLE>(name, 1, threads, IT);                \
  run<ROT, NV, SMALL, ROLE>(name, 256, threads, IT);

int main() {
  hipMalloc(&g_out, 256 * 512 * 4);
  hipMalloc(&g_cyc, 64);
  const int IT = 20000;
  printf("# 24 MFMAs per trip; 'cycles per MFMA slot' = loop cycles / (24 x trips); VALU-only rows use the same divisor\n");
  printf("## 32x32x16 bf16, 1 wave per SIMD\n");
  BOTH(0, 0, 0, 0, "chain  NV=0", 256)
  BOTH(1, 0, 0, 0, "rotate NV=0", 256)
  BOTH(0, 2, 0, 0, "chain  NV=2", 256)
  BOTH(1, 2, 0, 0, "rotate NV=2", 256)
  BOTH(0, 4, 0, 0, "chain  NV=4", 256)
  BOTH(1, 4, 0, 0, "rotate NV=4", 256)
  BOTH(0, 5, 0, 0, "chain  NV=5", 256)
  BOTH(1, 5, 0, 0, "rotate NV=5", 256)
  BOTH(0, 6, 0, 0, "chain  NV=6", 256)
  BOTH(1, 6, 0, 0, "rotate NV=6", 256)
  BOTH(0, 8, 0, 0, "chain  NV=8", 256)
  BOTH(1, 8, 0, 0, "rotate NV=8", 256)
  BOTH(1, 12, 0, 0, "rotate NV=12", 256)
  BOTH(1, 16, 0, 0, "rotate NV=16", 256)
  printf("## VALU only (NV per slot, no MFMA), 1 wave per SIMD\n");
  BOTH(1, 4, 0, 2, "VALU only NV=4", 256)
  BOTH(1, 8, 0, 2, "VALU only NV=8", 256)
  printf("## 32x32x16 bf16, 2 waves per SIMD: waves 0-3 MFMA only, waves 4-7 VALU only (NV per MFMA slot)\n");
  BOTH(0, 4, 0, 1, "chain  M-wave + V-wave NV=4", 512)
  BOTH(1, 4, 0, 1, "rotate M-wave + V-wave NV=4", 512)
  BOTH(0, 8, 0, 1, "chain  M-wave + V-wave NV=8", 512)
  BOTH(1, 8, 0, 1, "rotate M-wave + V-wave NV=8", 512)
  BOTH(1, 12, 0, 1, "rotate M-wave + V-wave NV=12", 512)
  printf("## 32x32x16 bf16, 2 mixed waves per SIMD (each wave: MFMA + NV VALU; twice the work of the 1-wave rows per SIMD)\n");
  BOTH(0, 4, 0, 0, "chain  2 mixed waves NV=4", 512)
  BOTH(1, 4, 0, 0, "rotate 2 mixed waves NV=4", 512)
  BOTH(0, 8, 0, 0, "chain  2 mixed waves NV=8", 512)
  BOTH(1, 8, 0, 0, "rotate 2 mixed waves NV=8", 512)
  printf("## 16x16x32 bf16, 1 wave per SIMD\n");
  BOTH(0, 0, 1, 0, "chain  NV=0 (16x16x32)", 256)
  BOTH(1, 0, 1, 0, "rotate NV=0 (16x16x32)", 256)
  BOTH(0, 2, 1, 0, "chain  NV=2 (16x16x32)", 256)
  BOTH(1, 2, 1, 0, "rotate NV=2 (16x16x32)", 256)
  BOTH(1, 3, 1, 0, "rotate NV=3 (16x16x32)", 256)
  BOTH(0, 4, 1, 0, "chain  NV=4 (16x16x32)", 256)
  BOTH(1, 4, 1, 0, "rotate NV=4 (16x16x32)", 256)
  return 0;
}
