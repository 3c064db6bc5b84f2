// Issue cost of the VALU / LDS instructions the epilogues are made of, in shader cycles per wave-instruction (s_memtime),
// one wave per SIMD, independent dependency chains (16 registers round-robin).  gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_cost.hip -o tools/_bin/valu_cost && tools/_bin/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256, 1) void k(float *out, long long *cyc, int iters, float c, unsigned m) {
  __shared__ float lds[4096];
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 0.001f + i;
  unsigned *u = reinterpret_cast<unsigned *>(v);
  lds[threadIdx.x] = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define X(i)                                                                                                           \
  if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(v[i]) : "v"(c));                                        \
  if (OP == 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(m));                                             \
  if (OP == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));                                             \
  if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(m));                  \
  if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&v[2 * i])) : "v"(*reinterpret_cast<double *>(&v[2 * ((i + 1) & 15)])));  \
  if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double *>(&v[2 * i])) : "v"(*reinterpret_cast<double *>(&v[2 * ((i + 1) & 15)])));  \
  if (OP == 6) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(v[i])::"vcc");            \
  if (OP == 7) asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_cndmask_b32 %0, 0, %1, vcc\n\tv_addc_co_u32 %2, vcc, %2, %2, vcc" : "=&v"(v[16 + i]), "+v"(v[i]), "+v"(u[(i + 8) & 15])::"vcc");  \
  if (OP == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));                                              \
  if (OP == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));                                              \
  if (OP == 10) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));                                             \
  if (OP == 11) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));                                     \
  if (OP == 12) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));                                                  \
  if (OP == 13) asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\tv_cndmask_b32 %1, 0, %1, vcc" : "+v"(u[i]), "+v"(v[16 + i])::"vcc"); \
  if (OP == 14) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));                                                          \
  if (OP == 15) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 8) & 15]));                       \
  if (OP == 16) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(u[i]) : "v"(u[31]));           \
  if (OP == 17) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]));                 \
  if (OP == 18) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(v[16 + i]));                            \
  if (OP == 19) asm volatile("v_bfe_u32 %0, %0, 16, 16" : "+v"(u[i]));
    REP16(X)
    REP16(X)
#undef X
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float *g_out;
static long long *g_cyc;
template <int OP>
void run(const char *name, int per) {
  const int IT = 4000;
  hipLaunchKernelGGL((k<OP>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, IT, 0.999f, 0xffff0000u);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
  printf("%-46s %6.2f cycles per group of %d instruction(s)\n", name, (double)c / (32.0 * IT), per);
}
int main() {
  hipMalloc(&g_out, 256 * 256 * 4);
  hipMalloc(&g_cyc, 64);
  run<0>("v_fma_f32", 1);
  run<1>("v_and_b32", 1);
  run<2>("v_sub_f32", 1);
  run<10>("v_add_f32", 1);
  run<9>("v_mul_f32", 1);
  run<8>("v_max_f32", 1);
  run<18>("v_fmac_f32", 1);
  run<3>("v_perm_b32", 1);
  run<12>("v_lshlrev_b32", 1);
  run<19>("v_bfe_u32", 1);
  run<4>("v_pk_add_f32", 1);
  run<5>("v_pk_fma_f32", 1);
  run<11>("v_cvt_pk_bf16_f32", 1);
  run<6>("v_cmp + v_cndmask (relu)", 2);
  run<7>("v_cmp + v_cndmask + v_addc (relu_push)", 3);
  run<13>("v_add_co + v_cndmask (mask_pop)", 2);
  run<14>("v_exp_f32", 1);
  run<15>("v_permlane32_swap_b32", 1);
  run<16>("ds_bpermute_b32 + wait", 1);
  run<17>("v_mov_b32_dpp row_shr:1", 1);
  return 0;
}
