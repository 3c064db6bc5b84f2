// host_rng.hip -- host-side (no device code): a bit-exact, faster restatement of torch.randperm on the CPU generator.
//
// The reference draws every minibatch permutation with torch.randperm(B) on the global CPU generator
// (on_policy_actor_buffer.py:131, on_policy_critic_buffer_ep.py:223); for B = 819200 ATen's serial Fisher-Yates costs
// 35-95 ms of host time per draw, 20 draws per train() -- far more than the whole update on the GPU.  This file replays
// exactly the same algorithm (ATen randperm_cpu, "small n" branch: for i in [0, n-1): z = random() % (n - i);
// swap(r[i], r[i+z]), with random() = one tempered mt19937 output) from a COPY of the generator state, on 32-bit indices
// with the random numbers generated in bulk.  torch's own generator is then advanced by the same n-1 draws
// (harl_amd/buffers.py), so the RNG stream of the run is unchanged.  tests/test_cabi.py checks the permutations against
// torch.randperm for many (seed, n).
#include <cstdint>
#include <cstdlib>
#include <immintrin.h>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/harl_hip.h"

namespace {
constexpr int MT_N = 624, MT_M = 397;

struct Mt {
  uint32_t s[MT_N];
  int left, next;
  static inline uint32_t tw(uint32_t u, uint32_t v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
  }
  void next_state() {  // at::mt19937::next_state
    uint32_t *p = s;
    left = MT_N;
    next = 0;
    for (int j = MT_N - MT_M + 1; --j; p++) *p = p[MT_M] ^ tw(p[0], p[1]);
    for (int j = MT_M; --j; p++) *p = p[MT_M - MT_N] ^ tw(p[0], p[1]);
    *p = p[MT_M - MT_N] ^ tw(p[0], s[0]);
  }
  inline uint32_t draw() {  // at::mt19937::operator()
    if (--left == 0) next_state();
    uint32_t y = s[next++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
};
// at::mt19937::next_state with 8 words per step (unaligned loads; the recurrence reaches back 227 / forward 397 words, so
// 8-wide blocks never read a word written in the same block).  hipcc's host compiler does not vectorise the scalar loops.
__attribute__((target("avx2"))) inline __m256i twist8(const uint32_t *pu, const uint32_t *pv, const uint32_t *pm) {
  const __m256i upper = _mm256_set1_epi32((int)0x80000000u), lower = _mm256_set1_epi32(0x7fffffff),
                matrix = _mm256_set1_epi32((int)0x9908b0dfu), one = _mm256_set1_epi32(1);
  const __m256i u = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(pu));
  const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(pv));
  const __m256i m = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(pm));
  const __m256i y = _mm256_or_si256(_mm256_and_si256(u, upper), _mm256_and_si256(v, lower));
  const __m256i odd = _mm256_cmpeq_epi32(_mm256_and_si256(v, one), one);  // (v & 1) ? matrix : 0
  return _mm256_xor_si256(_mm256_xor_si256(m, _mm256_srli_epi32(y, 1)), _mm256_and_si256(odd, matrix));
}

__attribute__((target("avx2"))) void next_state_avx2(Mt &g) {
  uint32_t *s = g.s;
  int i = 0;
  for (; i + 8 <= MT_N - MT_M; i += 8)  // words 0 .. 226: partner s[i + 397]
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(s + i), twist8(s + i, s + i + 1, s + i + MT_M));
  for (; i < MT_N - MT_M; ++i) s[i] = s[i + MT_M] ^ Mt::tw(s[i], s[i + 1]);
  for (; i + 8 <= MT_N - 1; i += 8)     // words 227 .. 622: partner s[i - 227] (already refreshed)
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(s + i), twist8(s + i, s + i + 1, s + i + MT_M - MT_N));
  for (; i < MT_N - 1; ++i) s[i] = s[i + MT_M - MT_N] ^ Mt::tw(s[i], s[i + 1]);
  s[MT_N - 1] = s[MT_M - 1] ^ Mt::tw(s[MT_N - 1], s[0]);
  g.left = MT_N;
  g.next = 0;
}

__attribute__((target("avx512f"))) inline __m512i twist16(const uint32_t *pu, const uint32_t *pv, const uint32_t *pm) {
  const __m512i upper = _mm512_set1_epi32((int)0x80000000u), lower = _mm512_set1_epi32(0x7fffffff),
                matrix = _mm512_set1_epi32((int)0x9908b0dfu), one = _mm512_set1_epi32(1);
  const __m512i u = _mm512_loadu_si512(pu), v = _mm512_loadu_si512(pv), m = _mm512_loadu_si512(pm);
  const __m512i y = _mm512_or_si512(_mm512_and_si512(u, upper), _mm512_and_si512(v, lower));
  const __mmask16 odd = _mm512_test_epi32_mask(v, one);
  const __m512i r = _mm512_xor_si512(m, _mm512_srli_epi32(y, 1));
  return _mm512_mask_xor_epi32(r, odd, r, matrix);
}

__attribute__((target("avx512f"))) void next_state_avx512(Mt &g) {
  uint32_t *s = g.s;
  int i = 0;
  for (; i + 16 <= MT_N - MT_M; i += 16) _mm512_storeu_si512(s + i, twist16(s + i, s + i + 1, s + i + MT_M));
  for (; i < MT_N - MT_M; ++i) s[i] = s[i + MT_M] ^ Mt::tw(s[i], s[i + 1]);
  for (; i + 16 <= MT_N - 1; i += 16) _mm512_storeu_si512(s + i, twist16(s + i, s + i + 1, s + i + MT_M - MT_N));
  for (; i < MT_N - 1; ++i) s[i] = s[i + MT_M - MT_N] ^ Mt::tw(s[i], s[i + 1]);
  s[MT_N - 1] = s[MT_M - 1] ^ Mt::tw(s[MT_N - 1], s[0]);
  g.left = MT_N;
  g.next = 0;
}

// instruction set of the host loops: 2 = AVX-512, 1 = AVX2, 0 = baseline; HARL_RNG_ISA=avx2|base caps it (tests)
int rng_isa() {
  int isa = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
  if (const char *e = std::getenv("HARL_RNG_ISA")) {
    if (!std::strcmp(e, "base")) isa = 0;
    else if (!std::strcmp(e, "avx2") && isa > 1) isa = 1;
  }
  return isa;
}

// k[i] = i + random() % (n - i) for i < nd, in two vectorisable passes:
//  1. the tempered mt19937 outputs in bulk, one 624-word state block at a time (the per-draw form -- refresh test, load,
//     temper, store -- runs at ~3 ns per draw; the block form at ~0.5 ns, and the refresh loops vectorise too);
//  2. the remainders: the quotient comes from a double division (exact to within one unit for 32-bit operands, fixed up
//     below), which pipelines / vectorises; an integer `%` per element is 3x slower.
// Same draw order and the same generator bookkeeping (left / next) as at::mt19937::operator().
#define HARL_DRAWS_BODY                                                                                        \
  long i = 0;                                                                                                  \
  while (i < nd) {                                                                                             \
    if (g.left <= 1) { /* operator(): --left == 0 -> next_state() */                                           \
      HARL_REFRESH(g);                                                                                         \
      g.left = MT_N + 1;                                                                                       \
    }                                                                                                          \
    const long avail = g.left - 1, take = nd - i < avail ? nd - i : avail;                                     \
    const uint32_t *src = g.s + g.next;                                                                        \
    for (long j = 0; j < take; ++j) {                                                                          \
      uint32_t y = src[j];                                                                                     \
      y ^= (y >> 11);                                                                                          \
      y ^= (y << 7) & 0x9d2c5680u;                                                                             \
      y ^= (y << 15) & 0xefc60000u;                                                                            \
      y ^= (y >> 18);                                                                                          \
      k[i + j] = y;                                                                                            \
    }                                                                                                          \
    g.left -= (int)take;                                                                                       \
    g.next += (int)take;                                                                                       \
    i += take;                                                                                                 \
  }                                                                                                            \
  for (long e = 0; e < nd; ++e) {                                                                              \
    const uint32_t z = k[e], m = (uint32_t)(n - e);                                                            \
    uint32_t q = (uint32_t)((double)z / (double)m);                                                            \
    uint32_t rem = z - q * m;                                                                                  \
    if ((int32_t)rem < 0) rem += m; /* q one too large */                                                      \
    if (rem >= m) rem -= m;         /* q one too small */                                                      \
    k[e] = (uint32_t)e + rem;                                                                                  \
  }

#define HARL_REFRESH(g) next_state_avx512(g)
__attribute__((target("avx512f,avx2"))) void draws_and_targets_avx512(Mt &g, uint32_t *k, long n, long nd) { HARL_DRAWS_BODY }
#undef HARL_REFRESH
#define HARL_REFRESH(g) next_state_avx2(g)
__attribute__((target("avx2"))) void draws_and_targets_avx2(Mt &g, uint32_t *k, long n, long nd) { HARL_DRAWS_BODY }
#undef HARL_REFRESH
#define HARL_REFRESH(g) (g).next_state()
void draws_and_targets_base(Mt &g, uint32_t *k, long n, long nd) { HARL_DRAWS_BODY }
#undef HARL_REFRESH
#undef HARL_DRAWS_BODY

// discard nd draws: only the state refreshes remain (the outputs are never tempered)
#define HARL_SKIP_BODY                 \
  while (nd > 0) {                     \
    if (g.left <= 1) {                 \
      HARL_REFRESH(g);                 \
      g.left = MT_N + 1;               \
    }                                  \
    const long avail = g.left - 1, take = nd < avail ? nd : avail; \
    g.left -= (int)take;               \
    g.next += (int)take;               \
    nd -= take;                        \
  }
#define HARL_REFRESH(g) next_state_avx512(g)
__attribute__((target("avx512f,avx2"))) void skip_avx512(Mt &g, long nd) { HARL_SKIP_BODY }
#undef HARL_REFRESH
#define HARL_REFRESH(g) next_state_avx2(g)
__attribute__((target("avx2"))) void skip_avx2(Mt &g, long nd) { HARL_SKIP_BODY }
#undef HARL_REFRESH
#define HARL_REFRESH(g) (g).next_state()
void skip_base(Mt &g, long nd) { HARL_SKIP_BODY }
#undef HARL_REFRESH
#undef HARL_SKIP_BODY

// ---------------------------------------------------------------------------------------------
// Jump-ahead.  Advancing the generator by n draws only refreshes state blocks, floor(n / 624) of them; for the draw counts of
// a data-parallel run (every rank replays the GLOBAL batch: 6.5 M draws per sampler call at 8 x 4096 threads) that is
// 10 000 block refreshes = 0.8 ms per call, 20 calls per update.  The block refresh is a linear map over GF(2): with
// T = one word step of the recurrence x[i+624] = x[i+397] ^ tw(x[i], x[i+1]) on the 19937-bit state, a jump by J steps is
// g(T) with g(x) = x^J mod phi(x), phi = the characteristic polynomial of T (degree 19937), and
//     (g(T) s)[j] = XOR over the set bits i of g of x[i + j],   x[0 ..] = the word sequence that starts at the current block,
// i.e. 20 561 generated words and ~10 000 XORs of 624-word windows: ~0.1 ms independent of J (Haramoto, Matsumoto, Nishimura,
// Panneton, L'Ecuyer: "Efficient jump ahead for F2-linear random number generators", 2008).  phi comes from Berlekamp-Massey
// on 2 x 19937 output bits (once per process, ~15 ms); g from square-and-multiply (once per distinct block count, cached).
// The low 31 bits of a block's word 0 are not part of the linear state, so the jump stops ONE block short and the last
// refresh is the ordinary next_state(): every word of the final block is then exact, as the bit-identical-state test demands.
// ---------------------------------------------------------------------------------------------
constexpr int MT_DEG = 19937, PW = (MT_DEG + 1 + 63) / 64;  // polynomial words (bits 0 .. 19937)

struct Poly {
  uint64_t w[PW];
};
inline bool pbit(const uint64_t *w, int i) { return (w[i >> 6] >> (i & 63)) & 1u; }

const Poly &mt_charpoly() {
  static Poly phi;
  static bool done = false;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (done) return phi;
  // output bit sequence b_t = bit 0 of the t-th generated word, from an arbitrary non-zero state
  constexpr int NB = 2 * MT_DEG + 64;
  std::vector<uint8_t> b(NB);
  Mt g;
  for (int i = 0; i < MT_N; ++i) g.s[i] = 0x9e3779b9u * (uint32_t)(i + 1) + 12345u;
  for (int t = 0; t < NB;) {
    g.next_state();
    for (int i = 0; i < MT_N && t < NB; ++i, ++t) b[t] = g.s[i] & 1u;
  }
  // Berlekamp-Massey over GF(2) on bit sets; `win` holds b[n-1], b[n-2], ... at bits 1, 2, ... (bit 0 = b[n] pairs with c_0)
  constexpr int W = (MT_DEG + 2 + 63) / 64 + 1;
  std::vector<uint64_t> C(W, 0), B(W, 0), T(W, 0), win(W, 0);
  C[0] = B[0] = 1;
  int L = 0, m = 1;
  for (int n = 0; n < NB; ++n) {
    for (int k = W - 1; k > 0; --k) win[k] = (win[k] << 1) | (win[k - 1] >> 63);  // window <<= 1, insert b[n] at bit 0
    win[0] = (win[0] << 1) | b[n];
    uint64_t acc = 0;
    const int lw = (L >> 6) + 1;
    for (int k = 0; k < lw && k < W; ++k) acc ^= C[k] & win[k];
    const bool d = __builtin_parityll(acc);
    if (!d) {
      ++m;
      continue;
    }
    const bool grow = 2 * L <= n;
    if (grow) T = C;
    const int ws = m >> 6, bs = m & 63;  // C ^= B << m
    for (int k = W - 1; k >= ws; --k) {
      uint64_t v = B[k - ws] << bs;
      if (bs && k - ws - 1 >= 0) v |= B[k - ws - 1] >> (64 - bs);
      C[k] ^= v;
    }
    if (grow) {
      L = n + 1 - L;
      B = T;
      m = 1;
    } else {
      ++m;
    }
  }
  // connection polynomial C (c_0 = 1, degree L = 19937) -> characteristic polynomial phi(x) = x^L C(1/x)
  for (int k = 0; k < PW; ++k) phi.w[k] = 0;
  if (L == MT_DEG)
    for (int i = 0; i <= L; ++i)
      if (pbit(C.data(), i)) phi.w[(L - i) >> 6] |= 1ull << ((L - i) & 63);
  done = true;
  return phi;
}

// r = a * a mod phi   (a, r: degree < 19937)
void poly_sqr_mod(const Poly &a, const Poly &phi, Poly &r) {
  static const auto spread = [] {  // byte -> 16 bits with zeros interleaved
    std::vector<uint16_t> t(256);
    for (int v = 0; v < 256; ++v) {
      uint16_t o = 0;
      for (int k = 0; k < 8; ++k) o |= (uint16_t)((v >> k) & 1) << (2 * k);
      t[v] = o;
    }
    return t;
  }();
  uint64_t sq[2 * PW + 1];
  for (int k = 0; k < PW; ++k) {
    uint64_t lo = 0, hi = 0;
    for (int bt = 0; bt < 4; ++bt) {
      lo |= (uint64_t)spread[(a.w[k] >> (8 * bt)) & 0xff] << (16 * bt);
      hi |= (uint64_t)spread[(a.w[k] >> (32 + 8 * bt)) & 0xff] << (16 * bt);
    }
    sq[2 * k] = lo;
    sq[2 * k + 1] = hi;
  }
  sq[2 * PW] = 0;
  for (int dgr = 2 * (MT_DEG - 1); dgr >= MT_DEG; --dgr) {  // clear the high bits with shifted copies of phi
    if (!pbit(sq, dgr)) continue;
    const int sh = dgr - MT_DEG, ws = sh >> 6, bs = sh & 63;
    for (int k = 0; k < PW; ++k) {
      sq[k + ws] ^= phi.w[k] << bs;
      if (bs) sq[k + ws + 1] ^= phi.w[k] >> (64 - bs);
    }
  }
  for (int k = 0; k < PW; ++k) r.w[k] = sq[k];
}

// g = x^steps mod phi, cached per step count.  Returned BY VALUE (2.5 KB): the cache is cleared when it grows past 64 entries,
// so a reference into it would dangle as soon as a second thread advanced the generator by a new step count.
Poly jump_poly(long steps) {
  static std::map<long, Poly> cache;
  static std::mutex mu;
  const Poly &phi = mt_charpoly();
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(steps);
  if (it != cache.end()) return it->second;
  Poly g, t;
  for (int k = 0; k < PW; ++k) g.w[k] = 0;
  g.w[0] = 1;  // x^0
  int top = 63;
  while (top > 0 && !((steps >> top) & 1)) --top;
  for (int bit = top; bit >= 0; --bit) {
    poly_sqr_mod(g, phi, t);
    g = t;
    if ((steps >> bit) & 1) {  // g *= x
      for (int k = PW - 1; k > 0; --k) g.w[k] = (g.w[k] << 1) | (g.w[k - 1] >> 63);
      g.w[0] <<= 1;
      if (pbit(g.w, MT_DEG))
        for (int k = 0; k < PW; ++k) g.w[k] ^= phi.w[k];
    }
  }
  if (cache.size() > 64) cache.clear();
  return cache.emplace(steps, g).first->second;
}

#define HARL_XOR_WINDOW_BODY                                              \
  for (int i = 0; i < MT_DEG; ++i) {                                      \
    if (!pbit(gp.w, i)) continue;                                         \
    const uint32_t *src = x + i;                                          \
    HARL_XOR624(y, src)                                                   \
  }
#define HARL_XOR624(y, src) \
  for (int j = 0; j < MT_N; j += 16) _mm512_storeu_si512(y + j, _mm512_xor_si512(_mm512_loadu_si512(y + j), _mm512_loadu_si512(src + j)));
__attribute__((target("avx512f"))) void xor_windows_avx512(const Poly &gp, const uint32_t *x, uint32_t *y) { HARL_XOR_WINDOW_BODY }
#undef HARL_XOR624
#define HARL_XOR624(y, src)                                                                                             \
  for (int j = 0; j < MT_N; j += 8)                                                                                     \
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(y + j), _mm256_xor_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(y + j)), \
                                                                             _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + j))));
__attribute__((target("avx2"))) void xor_windows_avx2(const Poly &gp, const uint32_t *x, uint32_t *y) { HARL_XOR_WINDOW_BODY }
#undef HARL_XOR624
#define HARL_XOR624(y, src) \
  for (int j = 0; j < MT_N; ++j) y[j] ^= src[j];
void xor_windows_base(const Poly &gp, const uint32_t *x, uint32_t *y) { HARL_XOR_WINDOW_BODY }
#undef HARL_XOR624
#undef HARL_XOR_WINDOW_BODY

void refresh(Mt &g, int isa) {
  if (isa == 2) next_state_avx512(g);
  else if (isa == 1) next_state_avx2(g);
  else g.next_state();
}

// the state block `blocks` refreshes ahead of g's current one (g.left / g.next untouched), blocks >= 2
void jump_blocks(Mt &g, long blocks, int isa) {
  const Poly gp = jump_poly((blocks - 1) * (long)MT_N);
  constexpr int NBLK = (MT_DEG + MT_N - 1) / MT_N + 1;  // word sequence x[0 .. 19937 + 623]
  static thread_local std::vector<uint32_t> xs((NBLK + 1) * MT_N + 16);
  alignas(64) uint32_t y[MT_N + 16];
  uint32_t *x = xs.data();
  Mt t = g;
  std::memcpy(x, t.s, sizeof(t.s));
  for (int bk = 1; bk <= NBLK; ++bk) {
    refresh(t, isa);
    std::memcpy(x + bk * MT_N, t.s, sizeof(t.s));
  }
  std::memset(y, 0, sizeof(y));
  if (isa == 2) xor_windows_avx512(gp, x, y);
  else if (isa == 1) xor_windows_avx2(gp, x, y);
  else xor_windows_base(gp, x, y);
  std::memcpy(g.s, y, sizeof(g.s));
  refresh(g, isa);  // the last block refresh is the ordinary one: word 0's low bits are outside the linear state
}

thread_local long g_force_jump_min = -1;  // harl_rng_jump: threshold override for the calling thread
long jump_min_blocks() {
  if (g_force_jump_min >= 0) return g_force_jump_min;
  static const long v = [] {
    const char *e = std::getenv("HARL_RNG_JUMP_MIN_BLOCKS");
    return e ? std::atol(e) : 2500L;  // ~0.2 ms of direct block refreshes (AVX-512); below that the direct skip is cheaper
  }();
  return v;
}

// discard nd draws with the jump for the bulk: same bookkeeping as HARL_SKIP_BODY
void skip_with_jump(Mt &g, long nd, int isa) {
  // refreshes the direct skip would perform: the first happens when g.left <= 1, i.e. after g.left - 1 further draws
  const long avail0 = g.left >= 1 ? g.left - 1 : 0;
  if (nd <= avail0) {
    g.left -= (int)nd;
    g.next += (int)nd;
    return;
  }
  const long rest = nd - avail0;                 // draws after the first refresh point
  const long k = (rest + MT_N - 1) / MT_N;       // number of refreshes (>= 1)
  if (k >= jump_min_blocks()) {
    jump_blocks(g, k, isa);
  } else {
    for (long r = 0; r < k; ++r) refresh(g, isa);
  }
  const long used = rest - (k - 1) * MT_N;       // draws taken from the last block (1 .. 624)
  g.next = (int)used;
  g.left = MT_N + 1 - (int)used;
}

bool load_state(Mt &g, const uint8_t *state_in, long state_bytes) {
  if (state_bytes < 24 + 8 * MT_N) return false;
  int32_t left, seeded;
  uint64_t next;
  std::memcpy(&left, state_in + 8, 4);
  std::memcpy(&seeded, state_in + 12, 4);
  std::memcpy(&next, state_in + 16, 8);
  if (!seeded || left < 0 || left > MT_N || next > (uint64_t)MT_N) return false;
  for (int i = 0; i < MT_N; ++i) {
    uint64_t v;
    std::memcpy(&v, state_in + 24 + 8 * i, 8);
    g.s[i] = (uint32_t)v;
  }
  g.left = left;
  g.next = (int)next;
  return true;
}

void store_state(const Mt &g, const uint8_t *state_in, long state_bytes, uint8_t *state_out) {
  if (state_out != state_in) std::memcpy(state_out, state_in, (size_t)state_bytes);
  const int32_t left = g.left;
  const uint64_t next = (uint64_t)g.next;
  std::memcpy(state_out + 8, &left, 4);
  std::memcpy(state_out + 16, &next, 8);
  for (int i = 0; i < MT_N; ++i) {
    const uint64_t v = g.s[i];
    std::memcpy(state_out + 24 + 8 * i, &v, 8);
  }
}
}  // namespace

// state_in/state_out: the bytes of torch.get_rng_state() (CPUGeneratorImplState: uint64 seed; int left; int seeded;
// uint64 next; uint64 state[624]; ...); state_out = state_in advanced by the n-1 draws (pass it to torch.set_rng_state).
// out: int32[n] (the permutation), scratch: uint32[n] -- both caller-owned and reusable (fresh allocations cost more in
// page faults than the algorithm itself).  Returns 0, or -2 on an unexpected layout / n outside the 32-bit-draw branch.
extern "C" int harl_randperm_replay(const uint8_t *state_in, long state_bytes, long n, int32_t *out, uint32_t *scratch,
                                    uint8_t *state_out) {
  if (state_bytes < 24 + 8 * MT_N || n < 0 || n >= (long)(0xffffffffu / 20)) return -2;
  Mt g;
  int32_t left, seeded;
  uint64_t next;
  std::memcpy(&left, state_in + 8, 4);
  std::memcpy(&seeded, state_in + 12, 4);
  std::memcpy(&next, state_in + 16, 8);
  if (!seeded || left < 0 || left > MT_N || next > (uint64_t)MT_N) return -2;
  for (int i = 0; i < MT_N; ++i) {
    uint64_t v;
    std::memcpy(&v, state_in + 24 + 8 * i, 8);
    g.s[i] = (uint32_t)v;
  }
  g.left = left;
  g.next = (int)next;
  uint32_t *r = reinterpret_cast<uint32_t *>(out), *k = scratch;
  for (long i = 0; i < n; ++i) r[i] = (uint32_t)i;
  const long nd = n > 0 ? n - 1 : 0;
  const int isa = rng_isa();
  if (isa == 2) draws_and_targets_avx512(g, k, n, nd);
  else if (isa == 1) draws_and_targets_avx2(g, k, n, nd);
  else draws_and_targets_base(g, k, n, nd);
  constexpr long PF = 24;  // the swap partner is a random element of a 3 MB array: prefetch it a few iterations ahead
  for (long i = 0; i < nd; ++i) {
    if (i + PF < nd) __builtin_prefetch(r + k[i + PF], 1, 1);
    const uint32_t kk = k[i];
    const uint32_t sav = r[i];
    r[i] = r[kk];
    r[kk] = sav;
  }
  if (state_out) {
    if (state_out != state_in) std::memcpy(state_out, state_in, (size_t)state_bytes);
    left = g.left;
    next = (uint64_t)g.next;
    std::memcpy(state_out + 8, &left, 4);
    std::memcpy(state_out + 16, &next, 8);
    for (int i = 0; i < MT_N; ++i) {
      const uint64_t v = g.s[i];
      std::memcpy(state_out + 24 + 8 * i, &v, 8);
    }
  }
  return 0;
}

// state_out = state_in advanced by n_draws 32-bit draws (what torch.randperm(n_draws + 1) consumes when the permutation
// itself is not needed): ~0.3 ms per 819200 draws instead of ~1 ms for Tensor.random_ on the same generator.
extern "C" int harl_rng_advance(const uint8_t *state_in, long state_bytes, long n_draws, uint8_t *state_out) {
  Mt g;
  if (n_draws < 0 || !state_out || !load_state(g, state_in, state_bytes)) return -2;
  const int isa = rng_isa();
  if (n_draws / MT_N >= jump_min_blocks()) skip_with_jump(g, n_draws, isa);
  else if (isa == 2) skip_avx512(g, n_draws);
  else if (isa == 1) skip_avx2(g, n_draws);
  else skip_base(g, n_draws);
  store_state(g, state_in, state_bytes, state_out);
  return 0;
}

extern "C" int harl_rng_jump(const uint8_t *state_in, long state_bytes, long n_draws, uint8_t *state_out) {
  Mt g;
  if (n_draws < 0 || !state_out || !load_state(g, state_in, state_bytes)) return -2;
  g_force_jump_min = 2;
  skip_with_jump(g, n_draws, rng_isa());
  g_force_jump_min = -1;
  store_state(g, state_in, state_bytes, state_out);
  return 0;
}
