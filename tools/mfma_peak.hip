// Sustained fp32 MFMA rate of the chip (v_mfma_f32_32x32x2_f32, register operands only, no memory traffic):
// the practical ceiling the MLP kernels are compared with in DESIGN.md.  Build: hipcc --offload-arch=gfx950 -O3
// tools/mfma_peak.hip -o gpurun_out/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 12345.678f) out[threadIdx.x] = s;  // never true; keeps the chain alive
}

template <int NACC>
static void run(int wgs_per_cu, int iters) {
  float *out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1e-7f);
  hipDeviceSynchronize();
  std::vector<float> ms;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1e-7f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float t;
    hipEventElapsedTime(&t, e0, e1);
    ms.push_back(t);
  }
  const double flops = (double)grid * 4 * iters * 16 * 4096.0;
  float best = ms[0], last = ms.back();
  for (float t : ms) best = t < best ? t : best;
  printf("independent accumulators %d, %d waves/SIMD, %d MFMAs/wave: best %.3f ms = %.1f TFLOP/s, 5th run %.3f ms = %.1f TFLOP/s\n",
         NACC, wgs_per_cu, iters * 16, best, flops / best / 1e9, last, flops / last / 1e9);
  hipFree(out);
}

int main() {
  // short (~0.25 ms, like one MLP kernel) and long (~50 ms, sustained clocks under power limit)
  for (int iters : {256, 50000}) {
    run<4>(1, iters);
    run<4>(2, iters / 2);
    run<1>(2, iters / 2);
    run<2>(2, iters / 2);
  }
  return 0;
}
