// Issue cost and exactness of candidate instruction sequences for the exact three-way bf16 split of fp32 operands
// (csrc/split_mfma.h) on gfx950.  Round 3's valu_cost.hip measured v_and_b32 and v_perm_b32 -- the two instructions the split
// is made of -- at 8.2 / 8.5 cycles against 4.9 for a plain VALU op; this looks for cheaper exact sequences.
//   hipcc --offload-arch=gfx950 -O3 tools/split_cost.hip -o tools/_bin/split_cost && tools/_bin/split_cost
// Part 1: single instructions, 16 independent chains, one wave per SIMD, shader cycles per instruction.
// Part 2: whole split sequences per PAIR of values (8 independent pairs in flight), cycles per pair + exactness
//         (t1 + t2 + t3 == x in double, every term exactly a bf16, over 2^22 values with wide exponents).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256, 1) void k1(float *out, long long *cyc, int iters, float c, unsigned m, unsigned sel) {
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 0.001f + i;
  unsigned *u = reinterpret_cast<unsigned *>(v);
  const unsigned sm = __builtin_amdgcn_readfirstlane(m), ssel = __builtin_amdgcn_readfirstlane(sel);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define X(i)                                                                                                             \
  if (OP == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(m));                                                \
  if (OP == 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "s"(sm));                                               \
  if (OP == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[i]));                                                 \
  if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "v"(sel));                   \
  if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 15]), "s"(ssel));                  \
  if (OP == 6) asm volatile("v_pack_b32_f16 %0, %0, %1 op_sel:[1,1,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));              \
  if (OP == 7) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(u[16 + i]), "s"(sm));                     \
  if (OP == 8) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(v[i]) : "v"(u[16 + i]), "v"(m));                      \
  if (OP == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));                          \
  if (OP == 10) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[i]));                                                    \
  if (OP == 11) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[16 + i]));                              \
  if (OP == 12) asm volatile("v_bfi_b32 %0, %1, %0, 0" : "+v"(u[i]) : "s"(sm));                                           \
  if (OP == 13) asm volatile("v_cvt_f32_bf16 %0, %0" : "+v"(u[i]));                                                       \
  if (OP == 14) asm volatile("v_cvt_f32_bf16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(u[i])); \
  if (OP == 15) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));                                               \
  if (OP == 16) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&v[2 * (i & 7)])) : "v"(*reinterpret_cast<double *>(&v[16 + 2 * (i & 7)]))); \
  if (OP == 17) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[i]));                                                         \
  if (OP == 18) asm volatile("v_cmp_lt_f32 vcc, 0, %0\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "+v"(v[i]), "+v"(u[16 + (i & 7)])::"vcc"); \
  if (OP == 19) asm volatile("v_cmp_lt_f32 s[20:21], 0, %0\n\tv_addc_co_u32 %1, s[22:23], %1, %1, s[20:21]" : "+v"(v[i]), "+v"(u[16 + (i & 7)])::"s20", "s21", "s22", "s23"); \
  if (OP == 20) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "s"(sm), "v"(u[16 + i]));                       \
  if (OP == 21) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(u[i]));                                                    \
  if (OP == 22) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(m));                                               \
  if (OP == 23) asm volatile("v_or_b32 %0, %1, %0" : "+v"(u[i]) : "v"(m));                                                \
  if (OP == 24) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(v[16 + i]));                           \
  if (OP == 25) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(v[16 + i]));                                           \
  if (OP == 26) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "s"(sm), "v"(u[16 + i]));                              \
  if (OP == 27) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[16 + i]), "v"(u[16 + ((i + 1) & 15)]), "s"(ssel)); \
  if (OP == 28) asm volatile("v_cmp_lt_f32 vcc, 0, %2\n\tv_cndmask_b32 %0, 0, %2, vcc\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "=&v"(v[i]), "+v"(u[16 + (i & 7)]) : "v"(v[24 + (i & 7)]) : "vcc"); \
  if (OP == 29) asm volatile("v_max_f32 %0, 0, %2\n\tv_cmp_lt_f32 vcc, 0, %2\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc" : "=&v"(v[i]), "+v"(u[16 + (i & 7)]) : "v"(v[24 + (i & 7)]) : "vcc");
    REP16(X)
    REP16(X)
#undef X
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- whole split sequences: (f0, f1) -> p1, p2, p3 (packed bf16 pairs, low half = first value) ---------------------------
// V0: the library's split3<true>: perm / and / pk_sub (truncation)
// V1: same with the mask and the selector in SGPRs
// V2: v_pack_b32_f16 op_sel instead of v_perm
// V3: cvt_pk (RNE) + v_dot2_f32_bf16 remainders
// V4: cvt_pk (RNE) + shift / and + pk_sub
// V5: truncation with dot2 remainders: p = pack_hi ; r = dot2(p, -1 sel, f)
template <int V>
__device__ __forceinline__ void split_pair(float f0, float f1, unsigned &p1, unsigned &p2, unsigned &p3, unsigned sm, unsigned ssel,
                                           unsigned neg_lo, unsigned neg_hi) {
  if (V == 0 || V == 1) {
    const unsigned M = V == 0 ? 0xffff0000u : sm, S = V == 0 ? 0x07060302u : ssel;
    const unsigned b0 = __float_as_uint(f0), b1 = __float_as_uint(f1);
    p1 = __builtin_amdgcn_perm(b1, b0, S);
    const f32x2 r = f32x2{f0, f1} - f32x2{__uint_as_float(b0 & M), __uint_as_float(b1 & M)};
    const unsigned c0 = __float_as_uint(r[0]), c1 = __float_as_uint(r[1]);
    p2 = __builtin_amdgcn_perm(c1, c0, S);
    const f32x2 q = r - f32x2{__uint_as_float(c0 & M), __uint_as_float(c1 & M)};
    p3 = __builtin_amdgcn_perm(__float_as_uint(q[1]), __float_as_uint(q[0]), S);
  } else if (V == 2) {
    const unsigned b0 = __float_as_uint(f0), b1 = __float_as_uint(f1);
    asm("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(p1) : "v"(b0), "v"(b1));
    const f32x2 r = f32x2{f0, f1} - f32x2{__uint_as_float(b0 & sm), __uint_as_float(b1 & sm)};
    const unsigned c0 = __float_as_uint(r[0]), c1 = __float_as_uint(r[1]);
    asm("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(p2) : "v"(c0), "v"(c1));
    const f32x2 q = r - f32x2{__uint_as_float(c0 & sm), __uint_as_float(c1 & sm)};
    asm("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(p3) : "v"(q[0]), "v"(q[1]));
  } else if (V == 3) {
    float r0, r1, q0, q1;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(f0), "v"(f1));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r0) : "v"(p1), "s"(neg_lo), "v"(f0));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1) : "v"(p1), "s"(neg_hi), "v"(f1));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(r0), "v"(r1));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(q0) : "v"(p2), "s"(neg_lo), "v"(r0));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(q1) : "v"(p2), "s"(neg_hi), "v"(r1));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p3) : "v"(q0), "v"(q1));
  } else if (V == 4) {
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p1) : "v"(f0), "v"(f1));
    const f32x2 r = f32x2{f0, f1} - f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & sm)};
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p2) : "v"(r[0]), "v"(r[1]));
    const f32x2 q = r - f32x2{__uint_as_float(p2 << 16), __uint_as_float(p2 & sm)};
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p3) : "v"(q[0]), "v"(q[1]));
  } else if (V == 5) {
    float r0, r1, q0, q1;
    asm("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(p1) : "v"(f0), "v"(f1));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r0) : "v"(p1), "s"(neg_lo), "v"(f0));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1) : "v"(p1), "s"(neg_hi), "v"(f1));
    asm("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(p2) : "v"(r0), "v"(r1));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(q0) : "v"(p2), "s"(neg_lo), "v"(r0));
    asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(q1) : "v"(p2), "s"(neg_hi), "v"(r1));
    asm("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(p3) : "v"(q0), "v"(q1));
  } else if (V == 6) {  // V1 with the perm written as an SGPR-selector asm and the ands as SGPR-mask asm (no literals)
    const unsigned b0 = __float_as_uint(f0), b1 = __float_as_uint(f1);
    unsigned t0, t1, c0, c1, w0, w1;
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(p1) : "v"(b1), "v"(b0), "s"(ssel));
    asm("v_and_b32 %0, %1, %2" : "=v"(t0) : "s"(sm), "v"(b0));
    asm("v_and_b32 %0, %1, %2" : "=v"(t1) : "s"(sm), "v"(b1));
    const f32x2 r = f32x2{f0, f1} - f32x2{__uint_as_float(t0), __uint_as_float(t1)};
    c0 = __float_as_uint(r[0]);
    c1 = __float_as_uint(r[1]);
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(p2) : "v"(c1), "v"(c0), "s"(ssel));
    asm("v_and_b32 %0, %1, %2" : "=v"(w0) : "s"(sm), "v"(c0));
    asm("v_and_b32 %0, %1, %2" : "=v"(w1) : "s"(sm), "v"(c1));
    const f32x2 q = r - f32x2{__uint_as_float(w0), __uint_as_float(w1)};
    asm("v_perm_b32 %0, %1, %2, %3" : "=v"(p3) : "v"(q[1]), "v"(q[0]), "s"(ssel));
  }
}

template <int V>
__global__ __launch_bounds__(256, 1) void k2(const float *__restrict__ in, unsigned *__restrict__ out, long long *cyc, int iters,
                                             unsigned m, unsigned sel) {
  // timing: 16 values (8 pairs) per lane, re-split `iters` times (the inputs are perturbed by the previous terms so that
  // nothing is loop invariant); results are written for the exactness check on the first pass (iters == 1)
  const unsigned sm = __builtin_amdgcn_readfirstlane(m), ssel = __builtin_amdgcn_readfirstlane(sel);
  const unsigned neg_lo = __builtin_amdgcn_readfirstlane(0x0000BF80u), neg_hi = __builtin_amdgcn_readfirstlane(0xBF800000u);
  float f[16];
  const long base = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  for (int i = 0; i < 16; ++i) f[i] = in[base + i];
  unsigned p[8][3];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) split_pair<V>(f[2 * i], f[2 * i + 1], p[i][0], p[i][1], p[i][2], sm, ssel, neg_lo, neg_hi);
    if (it + 1 < iters) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // keep the loop honest: fold one result bit back into the inputs
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f[2 * i]) : "v"(p[i][2] & 1u));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  for (int i = 0; i < 8; ++i)
    for (int t = 0; t < 3; ++t) out[(base / 2 + i) * 3 + t] = p[i][t];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static float *g_out;
static long long *g_cyc;
template <int OP>
void run1(const char *name, int per) {
  const int IT = 4000;
  hipLaunchKernelGGL((k1<OP>), dim3(256), dim3(256), 0, 0, g_out, g_cyc, IT, 0.999f, 0xffff0000u, 0x07060302u);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
  printf("%-58s %6.2f cycles per group of %d instruction(s)\n", name, (double)c / (32.0 * IT), per);
}

static std::vector<float> g_in;
static float *d_in;
static unsigned *d_p;
template <int V>
void run2(const char *name) {
  const long NV = (long)g_in.size();
  const int blocks = (int)(NV / 16 / 256);
  hipLaunchKernelGGL((k2<V>), dim3(blocks), dim3(256), 0, 0, d_in, d_p, g_cyc, 1, 0xffff0000u, 0x07060302u);
  hipDeviceSynchronize();
  std::vector<unsigned> p(NV / 2 * 3);
  hipMemcpy(p.data(), d_p, p.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0, first = -1;
  double worst = 0;
  for (long i = 0; i < NV / 2; ++i) {
    for (int hsel = 0; hsel < 2; ++hsel) {
      const float x = g_in[2 * i + hsel];
      double s = 0;
      for (int t = 0; t < 3; ++t) {
        const unsigned w = p[i * 3 + t];
        const unsigned bits = hsel ? (w & 0xffff0000u) : (w << 16);
        float tf;
        memcpy(&tf, &bits, 4);
        s += (double)tf;
      }
      if (s != (double)x && !(std::isnan(x))) {
        if (first < 0) first = 2 * i + hsel;
        ++bad;
        const double e = std::fabs(s - (double)x) / (std::fabs((double)x) + 1e-300);
        if (e > worst) worst = e;
      }
    }
  }
  const int IT = 2000;
  hipLaunchKernelGGL((k2<V>), dim3(256), dim3(256), 0, 0, d_in, d_p, g_cyc, IT, 0xffff0000u, 0x07060302u);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
  printf("%-58s %7.2f cycles per pair   inexact %ld of %ld (worst rel %.2e%s)\n", name, (double)c / (8.0 * IT), bad, NV, worst,
         first >= 0 ? ", first at " : "");
  if (first >= 0) printf("      first inexact input: %a\n", g_in[first]);
}

int main() {
  hipMalloc(&g_out, 256 * 256 * 4);
  hipMalloc(&g_cyc, 64);
  printf("# part 1: issue cost per instruction (shader cycles, one wave per SIMD, 16 independent chains)\n");
  run1<15>("v_sub_f32 (reference: plain VALU)", 1);
  run1<24>("v_fma_f32 3 VGPR", 1);
  run1<25>("v_mov_b32", 1);
  run1<0>("v_and_b32 v, VGPR mask, v (in place)", 1);
  run1<1>("v_and_b32 v, SGPR mask, v (in place)", 1);
  run1<2>("v_and_b32 v, literal 0xffff0000, v (in place)", 1);
  run1<26>("v_and_b32 d, SGPR mask, s (out of place)", 1);
  run1<22>("v_xor_b32 VGPR", 1);
  run1<23>("v_or_b32 VGPR", 1);
  run1<3>("v_perm_b32 VGPR selector", 1);
  run1<4>("v_perm_b32 SGPR selector", 1);
  run1<27>("v_perm_b32 SGPR selector, out of place", 1);
  run1<6>("v_pack_b32_f16 op_sel:[1,1,0]", 1);
  run1<7>("v_dot2_f32_bf16 (SGPR operand)", 1);
  run1<8>("v_dot2_f32_bf16 (VGPR operand)", 1);
  run1<9>("v_cvt_pk_bf16_f32", 1);
  run1<10>("v_lshlrev_b32 16", 1);
  run1<21>("v_lshrrev_b32 16", 1);
  run1<11>("v_alignbit_b32 .., 31", 1);
  run1<12>("v_bfi_b32 SGPR mask", 1);
  run1<20>("v_and_or_b32", 1);
  run1<13>("v_cvt_f32_bf16", 1);
  run1<14>("v_cvt_f32_bf16_sdwa WORD_1", 1);
  run1<16>("v_pk_add_f32", 1);
  run1<17>("v_max_f32", 1);
  run1<18>("v_cmp_lt_f32 vcc + v_addc_co_u32 (mask bit push)", 2);
  run1<19>("v_cmp_lt_f32 s[20:21] + v_addc_co_u32 (SGPR pair)", 2);
  run1<28>("relu_push: v_cmp + v_cndmask + v_addc", 3);
  run1<29>("relu_push': v_max + v_cmp + v_addc", 3);

  printf("# part 2: whole split of a pair of fp32 values into three packed bf16 pairs\n");
  const long NV = 1L << 22;
  g_in.resize(NV);
  srand(7);
  for (long i = 0; i < NV; ++i) {
    // sign, 24 random significand bits, exponent spread over [-40, 40]; a few special values
    unsigned bits = ((unsigned)rand() << 16) ^ (unsigned)rand();
    const int e = 127 - 40 + rand() % 81;
    bits = (bits & 0x807fffffu) | ((unsigned)e << 23);
    if (i % 1024 == 0) bits = 0u;
    if (i % 1024 == 1) bits = 0x80000000u;
    if (i % 1024 == 2) bits = 0x3f800000u;
    if (i % 1024 == 3) bits = (bits & 0xff800000u) | 0x007fffffu;  // all-ones significand
    if (i % 1024 == 4) bits = (bits & 0xff800000u) | 0x00008000u;  // tie case of the first rounding
    if (i % 1024 == 5) bits = (bits & 0xff800000u) | 0x00018000u;
    if (i % 1024 >= 6 && i % 1024 < 16) bits = (bits & 0x807fffffu) | ((unsigned)(i % 1024 - 6) << 23);  // exponent fields 0..9: fp32 denormals and
                                                                                                  // values whose high half reads as an f16 denormal
    memcpy(&g_in[i], &bits, 4);
  }
  hipMalloc(&d_in, NV * 4);
  hipMalloc(&d_p, NV / 2 * 3 * 4);
  hipMemcpy(d_in, g_in.data(), NV * 4, hipMemcpyHostToDevice);
  run2<0>("V0 perm/and/pk_sub, literals (library today)");
  run2<1>("V1 same, mask and selector in SGPRs (compiler's choice)");
  run2<6>("V6 same, SGPR operands forced by asm");
  run2<2>("V2 v_pack_b32_f16 op_sel instead of v_perm");
  run2<3>("V3 cvt_pk_bf16 (RNE) + v_dot2_f32_bf16 remainders");
  run2<4>("V4 cvt_pk_bf16 (RNE) + shift/and + pk_sub");
  run2<5>("V5 v_pack hi halves + v_dot2_f32_bf16 remainders");
  return 0;
}
