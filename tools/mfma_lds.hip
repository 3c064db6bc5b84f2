// Where does the MFMA pipe time go in the hidden-layer kernels?  Synthetic variants of the k_fwd_hidden inner loop:
//   V0 registers only | V1 + A fragments from LDS (same double-buffered pattern) | V2 + B stream from HBM (ring prefetch)
//   V3 + epilogue-like VALU work and 16 float4 stores per slab
// Build/run: hipcc --offload-arch=gfx950 -O3 tools/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int V>
__global__ __launch_bounds__(512, 2) void k_var(const float *__restrict__ W, const float *__restrict__ xin,
                                               float *__restrict__ xout, long n_slabs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int H = 128, LDW = H + 1, NQ = 16, NT = 4;
  for (int e = threadIdx.x; e < H * H; e += blockDim.x) lds[(e / H) * LDW + (e % H)] = W[e];
  // V10/V11: MFMA-pipe mutex between the two waves of a SIMD (8-wave workgroup): token[simd] says whose turn it is
  volatile int *tok = reinterpret_cast<volatile int *>(lds + H * LDW);  // [4] token, [4] arrivals, [8] done
  if (threadIdx.x < 16) const_cast<int *>(tok)[threadIdx.x] = 0;
  __syncthreads();
  int simd = 0, rank = 0;
  if (V >= 10) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(hwid));  // SIMD_ID bits [5:4]
    simd = (int)hwid;
    if ((threadIdx.x & 63) == 0) rank = atomicAdd(const_cast<int *>(&tok[4 + simd]), 1);
    rank = __builtin_amdgcn_readfirstlane(rank);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const float *wl_lane = lds + i * LDW + 4 * h;
  float aX[16], aY[16];
  auto lds_frag = [&](int q, float (&a)[16]) {
    const float *wq = wl_lane + 32 * (q >> 2) + 8 * (q & 3);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < NT; ++t) a[c * NT + t] = wq[32 * t * LDW + c];
  };
  if (V >= 1) lds_frag(0, aX);
  else
    for (int k = 0; k < 16; ++k) aX[k] = aY[k] = 1e-3f * (k + lane);
  const int nwv = blockDim.x >> 6;
  const long slab0 = (long)blockIdx.x * nwv + wave, stride = (long)gridDim.x * nwv;
  f32x4 ring[2][4];
  if (V >= 2) {
    const f32x4 *p0 = reinterpret_cast<const f32x4 *>(xin + slab0 * 4096L) + lane;
    for (int u = 0; u < 4; ++u) ring[0][u] = p0[u * 64];
  } else {
    for (int u = 0; u < 4; ++u) ring[0][u] = ring[1][u] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
  }
  float keep = 0.f;
  if (V == 4 && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(127);  // stagger the two waves of each SIMD
  if (V == 5 && (blockIdx.x & 1)) __builtin_amdgcn_s_sleep(127);
  for (long slab = slab0; slab < n_slabs; slab += stride) {
    const f32x4 *xp = reinterpret_cast<const f32x4 *>(xin + slab * 4096L) + lane;
    const long ns = slab + stride < n_slabs ? slab + stride : slab;
    const f32x4 *xn = reinterpret_cast<const f32x4 *>(xin + ns * 4096L) + lane;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#define SUB(u, q, CONS, PROD, ACUR, ANXT)                                              \
    if (V >= 2) ring[PROD][u] = (q) + 4 < NQ ? xp[((q) + 4) * 64] : xn[((q) + 4 - NQ) * 64]; \
    if (V >= 1) lds_frag(((q) + 1) % NQ, ANXT);                                         \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) _Pragma("unroll") for (int t = 0; t < NT; ++t)     \
        acc[t] = MFMA(ACUR[c * NT + t], ring[CONS][u][c], acc[t]);
    if (V >= 10) {  // acquire the SIMD's matrix pipe
      while (tok[simd] != rank && tok[8 + simd * 2 + (1 - rank)] == 0) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll 1
    for (int qo = 0; qo < NQ; qo += 8) {
      SUB(0, qo + 0, 0, 1, aX, aY)
      SUB(1, qo + 1, 0, 1, aY, aX)
      SUB(2, qo + 2, 0, 1, aX, aY)
      SUB(3, qo + 3, 0, 1, aY, aX)
      SUB(0, qo + 4, 1, 0, aX, aY)
      SUB(1, qo + 5, 1, 0, aY, aX)
      SUB(2, qo + 6, 1, 0, aX, aY)
      SUB(3, qo + 7, 1, 0, aY, aX)
    }
    if (V >= 10) {  // release: partner's turn
      if ((threadIdx.x & 63) == 0) tok[simd] = 1 - rank;
    }
    if (V == 7) {  // stores only
      f32x4 *op = reinterpret_cast<f32x4 *>(xout + slab * 4096L) + lane;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = acc[t][4 * g + c];
          op[(4 * t + g) * 64] = o;
        }
    } else if (V == 8) {  // ~1/3 of the VALU epilogue: relu + sum
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += fmaxf(acc[t][r], 0.f);
      keep += s;
    } else if (V == 9) {  // the V6 epilogue twice (2x VALU work)
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = fmaxf(acc[t][r], 0.f + rep);
            acc[t][r] = v;
            s += v;
          }
        s += __shfl_xor(s, 32);
        const float mean = s * (1.f / 128);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[t][r] - mean;
            s2 += d * d;
          }
        s2 += __shfl_xor(s2, 32);
        const float rstd = 1.0f / sqrtf(s2 * (1.f / 128) + 1e-5f);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            acc[t][r] = (acc[t][r] - mean) * rstd;
            keep += acc[t][r];
          }
      }
    } else if (V >= 3) {  // relu + layernorm-like epilogue and the ATL store
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = fmaxf(acc[t][r], 0.f);
          acc[t][r] = v;
          s += v;
        }
      s += __shfl_xor(s, 32);
      const float mean = s * (1.f / 128);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[t][r] - mean;
          s2 += d * d;
        }
      s2 += __shfl_xor(s2, 32);
      const float rstd = 1.0f / sqrtf(s2 * (1.f / 128) + 1e-5f);
      if (V == 6) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) keep += (acc[t][r] - mean) * rstd;
      } else {
        f32x4 *op = reinterpret_cast<f32x4 *>(xout + slab * 4096L) + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (acc[t][4 * g + c] - mean) * rstd;
            op[(4 * t + g) * 64] = o;
          }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) keep += acc[t][0] + acc[t][7];
    }
  }
  if (V >= 10 && (threadIdx.x & 63) == 0) tok[8 + simd * 2 + rank] = 1;  // never block the partner again
  if (keep == 12345.678f) xout[threadIdx.x] = keep;
}

template <int V>
static void run(const float *W, const float *xin, float *xout, long n_slabs, const char *what, int grid = 512, int thr = 256) {
  const size_t shm = 128 * 129 * 4 + 64;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_var<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 8; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_var<V>, dim3(grid), dim3(thr), shm, 0, W, xin, xout, n_slabs);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float t;
    hipEventElapsedTime(&t, e0, e1);
    if (rep >= 2 && t < best) best = t;
  }
  printf("V%d %-58s %.4f ms  %.1f TFLOP/s\n", V, what, best, (double)n_slabs * 256 * 4096.0 / best / 1e9);
}

int main() {
  const long B = 819200, n_slabs = B / 32;
  float *W, *xin, *xout;
  hipMalloc(&W, 128 * 128 * 4);
  hipMalloc(&xin, B * 128 * 4);
  hipMalloc(&xout, B * 128 * 4);
  hipMemset(W, 0, 128 * 128 * 4);
  hipMemset(xin, 0, B * 128 * 4);
  run<0>(W, xin, xout, n_slabs, "registers only");
  run<1>(W, xin, xout, n_slabs, "+ A fragments from LDS (double-buffered, pinned)");
  run<2>(W, xin, xout, n_slabs, "+ B stream from HBM (ring prefetch 4 q-steps)");
  run<3>(W, xin, xout, n_slabs, "+ relu/LayerNorm epilogue and 16 float4 stores per slab");
  run<4>(W, xin, xout, n_slabs, "V3 + upper half of the workgroups delayed by 8k cycles");
  run<5>(W, xin, xout, n_slabs, "V3 + odd workgroups delayed by 8k cycles");
  run<6>(W, xin, xout, n_slabs, "V2 + relu/LayerNorm VALU work only (no stores)");
  run<7>(W, xin, xout, n_slabs, "V2 + 16 float4 stores per slab only (no VALU epilogue)");
  run<3>(W, xin, xout, n_slabs, "V3 as ONE 8-wave workgroup per CU (no mutex)", 256, 512);
  run<10>(W, xin, xout, n_slabs, "V3, 8-wave workgroup, MFMA-pipe mutex per SIMD pair", 256, 512);
  run<8>(W, xin, xout, n_slabs, "V2 + 1/3 VALU epilogue (relu + sum)");
  run<9>(W, xin, xout, n_slabs, "V2 + 2x VALU epilogue");
  run<2>(W, xin, xout, n_slabs, "V2 with ONE wave per SIMD (256 workgroups)", 256);
  run<6>(W, xin, xout, n_slabs, "V6 (VALU epilogue) with ONE wave per SIMD", 256);
  run<3>(W, xin, xout, n_slabs, "V3 (epilogue + stores) with ONE wave per SIMD", 256);
  return 0;
}
