// Pilot for the next optimisation step of the hidden-layer kernels: ONE wave per SIMD (whole 512-register file), the
// ReLU/LayerNorm epilogue and the stores of slab k sliced into the MFMA stream of slab k+1 (ping-pong accumulators).
// Compare with tools/mfma_lds.hip V2 (no epilogue) / V3 (epilogue after the MFMAs, two waves per SIMD).
// Build/run: hipcc --offload-arch=gfx950 -O3 tools/mfma_pipe.hip -o /tmp/mfma_pipe && /tmp/mfma_pipe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct Epi {
  float s, s2, mean, rstd;
};

template <int Q>
__device__ __forceinline__ void epi_slice(f32x16 (&P)[4], Epi &e, float *__restrict__ xout, long slab, int lane) {
  if constexpr (Q < 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = fmaxf(P[Q][r], 0.f);
      P[Q][r] = v;
      e.s += v;
    }
  } else if constexpr (Q == 4) {
    e.s += __shfl_xor(e.s, 32);
    e.mean = e.s * (1.f / 128);
  } else if constexpr (Q < 9) {
    constexpr int t = Q - 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = P[t][r] - e.mean;
      P[t][r] = d;
      e.s2 += d * d;
    }
  } else if constexpr (Q == 9) {
    e.s2 += __shfl_xor(e.s2, 32);
    e.rstd = 1.0f / sqrtf(e.s2 * (1.f / 128) + 1e-5f);
  } else if constexpr (Q < 14) {
    constexpr int t = Q - 10;
    f32x4 *op = reinterpret_cast<f32x4 *>(xout + slab * 4096L) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = P[t][4 * g + c] * e.rstd;
      op[(4 * t + g) * 64] = o;
    }
  } else if constexpr (Q == 15) {
    e.s = 0.f;
    e.s2 = 0.f;
  }
}

template <bool PIPE, int WPS>  // WPS: waves per SIMD the launch is sized for (1: 256 WGs, 2: 512 WGs)
__global__ __launch_bounds__(256, WPS) void k_pipe(const float *__restrict__ W, const float *__restrict__ xin,
                                                   float *__restrict__ xout, long n_slabs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int H = 128, LDW = H + 1, NQ = 16, NT = 4;
  for (int e = threadIdx.x; e < H * H; e += 256) lds[(e / H) * LDW + (e % H)] = W[e];
  __syncthreads();
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
  lds_frag(0, aX);
  const long slab0 = (long)blockIdx.x * 4 + wave, stride = (long)gridDim.x * 4;
  f32x4 ringA[8], ringB[8];
  {
    const f32x4 *p0 = reinterpret_cast<const f32x4 *>(xin + slab0 * 4096L) + lane;
#pragma unroll
    for (int u = 0; u < 8; ++u) ringA[u] = p0[u * 64];
  }
  f32x16 acc0[NT], acc1[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
  Epi e{0.f, 0.f, 0.f, 1.f};
  long prev = n_slabs;  // dummy slab behind the real ones: the first phase's "previous slab" epilogue lands there

#define SUB(u, q, CONS, PROD, ACUR, ANXT, ACC, PREV)                                               \
    PROD[u] = (q) + 8 < NQ ? xp[((q) + 8) * 64] : xn[((q) + 8 - NQ) * 64];                          \
    lds_frag(((q) + 1) % NQ, ANXT);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) _Pragma("unroll") for (int t = 0; t < NT; ++t)   \
        ACC[t] = MFMA(ACUR[c * NT + t], CONS[u][c], ACC[t]);                                       \
    if (PIPE) epi_slice<q>(PREV, e, xout, prev, lane);
#define PHASE(ACC, PREV)                                                                           \
    {                                                                                              \
      const f32x4 *xp = reinterpret_cast<const f32x4 *>(xin + slab * 4096L) + lane;                \
      const long ns = slab + stride < n_slabs ? slab + stride : slab;                              \
      const f32x4 *xn = reinterpret_cast<const f32x4 *>(xin + ns * 4096L) + lane;                  \
      _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int r = 0; r < 16; ++r) ACC[t][r] = 0.f; \
      SUB(0, 0, ringA, ringB, aX, aY, ACC, PREV) SUB(1, 1, ringA, ringB, aY, aX, ACC, PREV)        \
      SUB(2, 2, ringA, ringB, aX, aY, ACC, PREV) SUB(3, 3, ringA, ringB, aY, aX, ACC, PREV)        \
      SUB(4, 4, ringA, ringB, aX, aY, ACC, PREV) SUB(5, 5, ringA, ringB, aY, aX, ACC, PREV)        \
      SUB(6, 6, ringA, ringB, aX, aY, ACC, PREV) SUB(7, 7, ringA, ringB, aY, aX, ACC, PREV)        \
      SUB(0, 8, ringB, ringA, aX, aY, ACC, PREV) SUB(1, 9, ringB, ringA, aY, aX, ACC, PREV)        \
      SUB(2, 10, ringB, ringA, aX, aY, ACC, PREV) SUB(3, 11, ringB, ringA, aY, aX, ACC, PREV)      \
      SUB(4, 12, ringB, ringA, aX, aY, ACC, PREV) SUB(5, 13, ringB, ringA, aY, aX, ACC, PREV)      \
      SUB(6, 14, ringB, ringA, aX, aY, ACC, PREV) SUB(7, 15, ringB, ringA, aY, aX, ACC, PREV)      \
      if (!PIPE) {                                                                                 \
        epi_slice<0>(ACC, e, xout, slab, lane); epi_slice<1>(ACC, e, xout, slab, lane);            \
        epi_slice<2>(ACC, e, xout, slab, lane); epi_slice<3>(ACC, e, xout, slab, lane);            \
        epi_slice<4>(ACC, e, xout, slab, lane); epi_slice<5>(ACC, e, xout, slab, lane);            \
        epi_slice<6>(ACC, e, xout, slab, lane); epi_slice<7>(ACC, e, xout, slab, lane);            \
        epi_slice<8>(ACC, e, xout, slab, lane); epi_slice<9>(ACC, e, xout, slab, lane);            \
        epi_slice<10>(ACC, e, xout, slab, lane); epi_slice<11>(ACC, e, xout, slab, lane);          \
        epi_slice<12>(ACC, e, xout, slab, lane); epi_slice<13>(ACC, e, xout, slab, lane);          \
        epi_slice<15>(ACC, e, xout, slab, lane);                                                   \
      }                                                                                            \
      prev = slab;                                                                                 \
    }
  long slab = slab0;
  for (; slab + stride < n_slabs; slab += 2 * stride) {
    PHASE(acc0, acc1)
    slab += stride;
    PHASE(acc1, acc0)
    slab -= stride;
  }
  if (slab < n_slabs) {  // odd tail
    PHASE(acc0, acc1)
    if (PIPE) {
      epi_slice<0>(acc0, e, xout, prev, lane); epi_slice<1>(acc0, e, xout, prev, lane); epi_slice<2>(acc0, e, xout, prev, lane);
      epi_slice<3>(acc0, e, xout, prev, lane); epi_slice<4>(acc0, e, xout, prev, lane); epi_slice<5>(acc0, e, xout, prev, lane);
      epi_slice<6>(acc0, e, xout, prev, lane); epi_slice<7>(acc0, e, xout, prev, lane); epi_slice<8>(acc0, e, xout, prev, lane);
      epi_slice<9>(acc0, e, xout, prev, lane); epi_slice<10>(acc0, e, xout, prev, lane); epi_slice<11>(acc0, e, xout, prev, lane);
      epi_slice<12>(acc0, e, xout, prev, lane); epi_slice<13>(acc0, e, xout, prev, lane);
    }
  } else if (PIPE) {
    epi_slice<0>(acc1, e, xout, prev, lane); epi_slice<1>(acc1, e, xout, prev, lane); epi_slice<2>(acc1, e, xout, prev, lane);
    epi_slice<3>(acc1, e, xout, prev, lane); epi_slice<4>(acc1, e, xout, prev, lane); epi_slice<5>(acc1, e, xout, prev, lane);
    epi_slice<6>(acc1, e, xout, prev, lane); epi_slice<7>(acc1, e, xout, prev, lane); epi_slice<8>(acc1, e, xout, prev, lane);
    epi_slice<9>(acc1, e, xout, prev, lane); epi_slice<10>(acc1, e, xout, prev, lane); epi_slice<11>(acc1, e, xout, prev, lane);
    epi_slice<12>(acc1, e, xout, prev, lane); epi_slice<13>(acc1, e, xout, prev, lane);
  }
}

template <bool PIPE, int WPS>
static void run(const float *W, const float *xin, float *xout, long n_slabs, const char *what) {
  const size_t shm = 128 * 129 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k_pipe<PIPE, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 8; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_pipe<PIPE, WPS>), dim3(256 * WPS), dim3(256), shm, 0, W, xin, xout, n_slabs);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float t;
    hipEventElapsedTime(&t, e0, e1);
    if (rep >= 2 && t < best) best = t;
  }
  printf("%-72s %.4f ms  %.1f TFLOP/s\n", what, best, (double)n_slabs * 256 * 4096.0 / best / 1e9);
}

int main() {
  const long B = 819200, n_slabs = B / 32;
  float *W, *xin, *xout;
  hipMalloc(&W, 128 * 128 * 4);
  hipMalloc(&xin, B * 128 * 4);
  hipMalloc(&xout, (B + 32) * 128 * 4);
  hipMemset(W, 0, 128 * 128 * 4);
  hipMemset(xin, 0, B * 128 * 4);
  run<false, 2>(W, xin, xout, n_slabs, "epilogue after the MFMAs, 2 waves/SIMD (today's structure, RD=8)");
  run<false, 1>(W, xin, xout, n_slabs, "epilogue after the MFMAs, 1 wave/SIMD");
  run<true, 1>(W, xin, xout, n_slabs, "epilogue of slab k sliced into the MFMA stream of slab k+1, 1 wave/SIMD");
  run<true, 2>(W, xin, xout, n_slabs, "same, launched as 2 workgroups per CU (register file permitting)");
  return 0;
}
