// elementwise.hip -- HBM-bound scan / reduction / optimiser kernels of the HAPPO update path (gfx950).
//
//   harl_gae_returns        reverse scan over t, one lane per rollout column (coalesced across n)
//   harl_masked_moments     advantage statistics (masked sum / sumsq / count, fp64)
//   harl_adv_normalize      (adv - mean) / (std + 1e-5)
//   harl_factor_update      factor *= agg_d exp(new - old)
//   harl_sum_sumsq / harl_valuenorm_apply   PopArt-style running statistics
//   harl_gradnorm_clip_adam fused ||g|| + clip + Adam over one flat arena
//   harl_fold_linear / harl_unfold_linear_grads / harl_reduce_partials / harl_reduce_scalars
#include <mutex>
#include <map>
#include <unordered_map>
#include <utility>

#include "common.h"
#include "../../include/harl_hip.h"

#include <cstdio>
#include <cstring>

namespace harl {
bool lds_opt_in_needed(const void *kernel, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<const void *, size_t> done;
  std::lock_guard<std::mutex> lk(mu);
  size_t &have = done[kernel];
  if (have >= bytes) return false;
  have = bytes;
  return true;
}

static thread_local char g_err[512] = "";
void set_error(const char *msg) {
  std::strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    set_error(buf);
    return -1;
  }
  return 0;
}
}  // namespace harl

using namespace harl;

extern "C" int harl_version(void) { return 100; }
extern "C" const char *harl_last_error(void) { return g_err; }

// =============================================================================================
// GAE / returns.  20 algorithmic bytes per (t, column): 4 loads + 1 store (+4 for advantages).
// One lane per column: every load/store of a wave is 256 contiguous bytes.  The carry chain is
// sequential in t but the *inputs* do not depend on it, so the loop is unrolled by 8 and the
// compiler hoists the 8 steps' loads ahead of the dependent arithmetic (loads in flight per lane:
// 32) -- that, plus >=2 waves per SIMD at N >= 64K columns... at N = 4096 there are only 64 waves,
// so latency, not bandwidth, bounds this kernel; it is 0.1 % of the update either way.
// Arithmetic is the reference's, operation for operation, with FMA contraction disabled, so
// `returns` is bit-identical to NumPy's.
// =============================================================================================
struct DenormStats {
  float mean, sd;
  int on;
};

__device__ __forceinline__ DenormStats load_denorm(const float *vn) {
#pragma clang fp contract(off)
  DenormStats s;
  s.on = vn != nullptr;
  s.mean = 0.f;
  s.sd = 1.f;
  if (s.on) {  // valuenorm.py:38-45
    float d = fmaxf(vn[2], 1e-5f);
    float mean = vn[0] / d;
    float mean_sq = vn[1] / d;
    float var = fmaxf(mean_sq - mean * mean, 1e-2f);
    s.mean = mean;
    s.sd = sqrtf(var);
  }
  return s;
}
__device__ __forceinline__ float denorm(const DenormStats &s, float v) {
#pragma clang fp contract(off)
  if (!s.on) return v;
  float t = v * s.sd;
  return t + s.mean;
}

// One column (rollout thread, or thread x agent for FP) per lane, 64-lane workgroups (N = 4096 columns -> 64 workgroups on 64
// CUs instead of 16).  The scan over t is sequential per column and kept in the reference's operation order (bit-exact,
// contraction off); what is parallel is the MEMORY side: the inputs of GAE_TC time steps are fetched into registers before
// the dependent arithmetic of those steps starts (4 * GAE_TC independent coalesced loads in flight per lane instead of one
// load latency per step: the loop was latency-bound at ~550 ns / step).
constexpr int GAE_TC = 20;  // (8 until round 6 session 3; 2 x 20 steps of inputs in flight per lane: 160 registers)
template <bool GAE, bool PTL, bool FP_ORDER>
__global__ __launch_bounds__(64) void k_gae(const float *__restrict__ rewards, float *__restrict__ value_preds,
                                            const float *__restrict__ masks, const float *__restrict__ bad_masks,
                                            const float *__restrict__ next_value, const float *__restrict__ vn,
                                            float *__restrict__ returns, float *__restrict__ adv, int T, int ncols,
                                            float gamma, float gl) {
#pragma clang fp contract(off)
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  const DenormStats ds = load_denorm(vn);
  const long N = ncols;
  const float nv = next_value[c];
  float gae = 0.f, ret1 = nv;
  float dv1 = denorm(ds, nv);  // value_preds[t+1], denormalised
  if (GAE) value_preds[(long)T * N + c] = nv;  // critic_buffer_ep.py:107
  else returns[(long)T * N + c] = nv;
  // the inputs of chunk k + 1 are requested BEFORE the dependent arithmetic of chunk k (two register sets, round 6 session 3):
  // one memory round trip per chunk used to be exposed (25 of them per 200-step scan, ~1.5 us each)
  auto load = [&](int t0, float (&r)[GAE_TC], float (&v0)[GAE_TC], float (&m1)[GAE_TC], float (&b1)[GAE_TC]) {
#pragma unroll
    for (int k = 0; k < GAE_TC; ++k) {
      const int t = t0 - k;
      const bool ok = t >= 0;
      const long o = (long)(ok ? t : 0) * N + c;
      r[k] = rewards[o];
      v0[k] = value_preds[o];
      m1[k] = masks[o + N];
      b1[k] = PTL ? bad_masks[o + N] : 1.f;
    }
  };
  auto scan = [&](int t0, const float (&r)[GAE_TC], const float (&v0)[GAE_TC], const float (&m1)[GAE_TC], const float (&b1)[GAE_TC]) {
#pragma clang fp contract(off)
#pragma unroll
    for (int k = 0; k < GAE_TC; ++k) {
      const int t = t0 - k;
      if (t < 0) break;
      const float dv0 = denorm(ds, v0[k]);
      float ret;
      if (GAE) {
        float delta = r[k] + (gamma * dv1) * m1[k];
        delta = delta - dv0;
        const float carry = FP_ORDER ? (gl * gae) * m1[k] : (gl * m1[k]) * gae;
        gae = delta + carry;
        if (PTL) gae = b1[k] * gae;
        ret = gae + dv0;
        dv1 = dv0;
      } else {
        ret = (ret1 * gamma) * m1[k] + r[k];
        if (PTL) ret = ret * b1[k] + (1.f - b1[k]) * dv0;
        ret1 = ret;
      }
      returns[(long)t * N + c] = ret;
      if (adv) adv[(long)t * N + c] = ret - dv0;
    }
  };
  float rA[GAE_TC], vA[GAE_TC], mA[GAE_TC], bA[GAE_TC], rB[GAE_TC], vB[GAE_TC], mB[GAE_TC], bB[GAE_TC];
  int t0 = T - 1;
  load(t0, rA, vA, mA, bA);
  while (t0 >= 0) {
    if (t0 - GAE_TC >= 0) load(t0 - GAE_TC, rB, vB, mB, bB);
    scan(t0, rA, vA, mA, bA);
    t0 -= GAE_TC;
    if (t0 < 0) break;
    if (t0 - GAE_TC >= 0) load(t0 - GAE_TC, rA, vA, mA, bA);
    scan(t0, rB, vB, mB, bB);
    t0 -= GAE_TC;
  }
}

extern "C" int harl_gae_returns(const float *rewards, float *value_preds, const float *masks,
                                const float *bad_masks, const float *next_value, const float *vn_stats,
                                float *returns, float *advantages, int T, int ncols, float gamma,
                                float gamma_lambda, int use_gae, int use_proper_time_limits, int fp_order,
                                void *stream) {
  if (T <= 0 || ncols <= 0) return 0;
  dim3 block(64), grid((ncols + 63) / 64);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(G, P, F)                                                                                     \
  hipLaunchKernelGGL((k_gae<G, P, F>), grid, block, 0, s, rewards, value_preds, masks, bad_masks, next_value, \
                     vn_stats, returns, advantages, T, ncols, gamma, gamma_lambda)
  if (use_gae) {
    if (use_proper_time_limits) {
      if (fp_order) LAUNCH(true, true, true); else LAUNCH(true, true, false);
    } else {
      if (fp_order) LAUNCH(true, false, true); else LAUNCH(true, false, false);
    }
  } else {
    if (use_proper_time_limits) LAUNCH(false, true, false); else LAUNCH(false, false, false);
  }
#undef LAUNCH
  return check_launch("harl_gae_returns");
}

// =============================================================================================
// masked moments (fp64 accumulation, FIXED summation order: the same bits on every run)
// =============================================================================================
// Every block leaves its partial {sum, sumsq, count} in a scratch row; the block that takes the last ticket adds the rows in
// index order onto out3.  The scratch belongs to the launch STREAM (one allocation per stream, made on first use): launches
// on one stream are serialised, launches on different streams (the runner's side streams) never share rows or the ticket,
// and the ticket is cleared by a memset ahead of every launch, so an aborted launch cannot poison the next one.
constexpr int MM_MAX_BLOCKS = 1024;
struct MmScratch {
  double part[MM_MAX_BLOCKS][3];
  unsigned ticket;
};

__global__ __launch_bounds__(256) void k_masked_moments(const float *__restrict__ x, const float *__restrict__ active,
                                                        long n, double *__restrict__ out3, MmScratch *__restrict__ ws) {
  double(*g_mm_part)[3] = ws->part;
  unsigned &g_mm_ticket = ws->ticket;
  double s1 = 0, s2 = 0, cnt = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float a = active ? active[i] : 1.f;
    if (a != 0.f) {
      double v = x[i];
      s1 += v;
      s2 += v * v;
      cnt += 1.0;
    }
  }
  s1 = wave_reduce_sum_d(s1);
  s2 = wave_reduce_sum_d(s2);
  cnt = wave_reduce_sum_d(cnt);
  __shared__ double sh[3][4];
  __shared__ unsigned s_last;
  int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) {
    sh[0][w] = s1;
    sh[1][w] = s2;
    sh[2][w] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    __hip_atomic_store(&g_mm_part[blockIdx.x][threadIdx.x],
                       sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();  // release this block's row before taking the ticket
    s_last = __hip_atomic_fetch_add(&g_mm_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();  // acquire the other blocks' rows
  __shared__ double red[3][256];
  for (int k = 0; k < 3; ++k) {  // rows tid, tid + 256, ... in that order; then the 256 lane sums in lane order
    double t = 0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256)
      t += __hip_atomic_load(&g_mm_part[b][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[k][threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0;
    for (int i = 0; i < 256; ++i) t += red[threadIdx.x][i];
    out3[threadIdx.x] += t;
  }
}

static_assert(sizeof(MmScratch) <= HARL_MM_SCRATCH_BYTES, "HARL_MM_SCRATCH_BYTES too small");

// `scratch`: HARL_MM_SCRATCH_BYTES of device memory owned by the caller (8-byte aligned; one block per stream that may have
// a launch in flight) -- the library allocates nothing (include/harl_hip.h conventions).
extern "C" int harl_masked_moments(const float *x, const float *active, long n, double *out3, void *scratch, void *stream) {
  if (n <= 0) return 0;
  long nb = (n + 2047) / 2048;
  if (nb > MM_MAX_BLOCKS) nb = MM_MAX_BLOCKS;
  hipStream_t s = (hipStream_t)stream;
  MmScratch *ws = static_cast<MmScratch *>(scratch);
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 7) != 0) {
    set_error("harl_masked_moments: scratch must be HARL_MM_SCRATCH_BYTES of 8-byte aligned device memory");
    return -2;
  }
  (void)hipMemsetAsync(&ws->ticket, 0, sizeof(unsigned), s);
  hipLaunchKernelGGL(k_masked_moments, dim3((unsigned)nb), dim3(256), 0, s, x, active, n, out3, ws);
  return check_launch("harl_masked_moments");
}

__global__ __launch_bounds__(256) void k_adv_normalize(const float *__restrict__ adv, const double *__restrict__ mom,
                                                       float *__restrict__ out, long n) {
  double cnt = mom[2];
  double meand = mom[0] / cnt;
  double vard = mom[1] / cnt - meand * meand;
  float mean = (float)meand;
  float denom = (float)sqrt(vard > 0 ? vard : 0.0) + 1e-5f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = (adv[i] - mean) / denom;
}

extern "C" int harl_adv_normalize(const float *adv, const double *moments3, float *adv_out, long n, void *stream) {
  if (n <= 0) return 0;
  long nb = (n + 1023) / 1024;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_adv_normalize, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, adv, moments3, adv_out, n);
  return check_launch("harl_adv_normalize");
}

// =============================================================================================
// Rollout-side row arithmetic on the head outputs (models/base/act.py:45-157, distributions.py:31-103): what
// StochasticPolicy.forward / evaluate_actions do with the distribution once the head has produced its parameters.  The random
// draws themselves stay with torch's device generator (the reference's Normal.sample() / Categorical.sample() use it), this
// kernel does everything around them.  One thread per row.
//   kind 0  DiagGaussian: head = mean [M, D]; sigma_d = sigmoid(log_std_d / x_coef) * y_coef
//           actions = mean + sigma * noise (noise == NULL: the mode), logp_d = log N(a_d; mean_d, sigma_d),
//           ent_rows = sum_d (0.5 + 0.5 log 2 pi + log sigma_d), sigma_out [D]
//   kind 1  Categorical, n_heads heads whose normalised logits lie side by side in head [M, D] (head_off: n_heads + 1
//           offsets, NULL = one head): probs = exp(head) (the multinomial's input), argmax_out [M, n_heads] = first largest
//           logit per head, logp [M, n_heads] (or [M, 1] with sum_heads) = logit of `actions` (float indices [M, n_heads];
//           NULL: of the argmax), ent_rows = - sum_j clamp(logit_j) p_j over all heads (act.py:117-141 adds the heads' entropies)
// =============================================================================================
__global__ __launch_bounds__(256) void k_dist_rows(const float *__restrict__ head, long M, int D, int kind,
                                                   const float *__restrict__ log_std, float std_x_coef, float std_y_coef,
                                                   const float *__restrict__ noise, float *__restrict__ actions,
                                                   const int *__restrict__ head_off, int n_heads, int sum_heads,
                                                   float *__restrict__ logp, float *__restrict__ probs,
                                                   float *__restrict__ argmax_out, float *__restrict__ ent_rows,
                                                   float *__restrict__ sigma_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (kind == 0) {
    if (sigma_out && i < D) sigma_out[i] = (1.0f / (1.0f + expf(-log_std[i] / std_x_coef))) * std_y_coef;  // as the loss kernels
    if (i >= M) return;
    float ent = 0.f;
    for (int d = 0; d < D; ++d) {
      const float sig = (1.0f / (1.0f + expf(-log_std[d] / std_x_coef))) * std_y_coef;
      const float lsig = logf(sig);
      const float mu = head[i * D + d];
      const float a = noise ? mu + sig * noise[i * D + d] : mu;
      if (actions) actions[i * D + d] = a;
      if (logp) {
        const float diff = a - mu;
        logp[i * D + d] = -(diff * diff) / (2.0f * sig * sig) - lsig - 0.9189385332046727f;  // torch Normal.log_prob
      }
      ent += 0.5f + 0.9189385332046727f + lsig;
    }
    if (ent_rows) ent_rows[i] = ent;
    return;
  }
  if (i >= M) return;
  float ent = 0.f, lsum = 0.f;
  for (int hd = 0; hd < n_heads; ++hd) {
    const int lo = head_off ? head_off[hd] : 0, hi = head_off ? head_off[hd + 1] : D;
    int best = lo;
    float bv = head[i * D + lo];
    for (int j = lo; j < hi; ++j) {
      const float lg = head[i * D + j];
      const float p = expf(lg);
      if (probs) probs[i * D + j] = p;
      ent -= fmaxf(lg, -3.4028234663852886e38f) * p;
      if (lg > bv) {
        bv = lg;
        best = j;
      }
    }
    if (argmax_out) argmax_out[i * n_heads + hd] = (float)(best - lo);
    if (logp) {
      // a stored action outside [0, hi - lo) (padding rows, a corrupted buffer, a wrong nvec) must not become an out-of-bounds
      // read: the torch gather this replaces would have raised; here the log-prob of such a row is NaN
      const int a = actions ? (int)actions[i * n_heads + hd] : best - lo;
      const bool a_ok = a >= 0 && a < hi - lo;
      const float lp = a_ok ? head[i * D + lo + a] : __builtin_nanf("");
      if (sum_heads) lsum += lp;
      else logp[i * n_heads + hd] = lp;
    }
  }
  if (logp && sum_heads) logp[i] = lsum;
  if (ent_rows) ent_rows[i] = ent;
}

extern "C" int harl_dist_rows(const float *head, long M, int act_dim, int kind, const float *log_std, float std_x_coef,
                              float std_y_coef, const float *noise, float *actions, const int *head_off, int n_heads,
                              int sum_heads, float *logp, float *probs, float *argmax_out, float *ent_rows, float *sigma_out,
                              void *stream) {
  if (M <= 0) return 0;
  if (act_dim < 1 || (kind != 0 && kind != 1) || (kind == 0 && !log_std) || (kind == 1 && n_heads < 1)) {
    set_error("harl_dist_rows: bad arguments");
    return -1;
  }
  const long rows = M > act_dim ? M : act_dim;
  hipLaunchKernelGGL(k_dist_rows, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, head, M, act_dim, kind,
                     log_std, std_x_coef, std_y_coef, noise, actions, head_off, n_heads, sum_heads, logp, probs, argmax_out,
                     ent_rows, sigma_out);
  return check_launch("harl_dist_rows");
}

// mean over the counted entries of a harl_masked_moments triple {sum x, sum x^2, count}: the (active-mask-weighted) mean of
// the entropy rows (act.py:104-157), as one fp32 scalar on the device
__global__ void k_moments_mean(const double *__restrict__ mom, float *__restrict__ out) { out[0] = (float)(mom[0] / mom[2]); }

extern "C" int harl_moments_mean(const double *moments3, float *mean_out, void *stream) {
  hipLaunchKernelGGL(k_moments_mean, dim3(1), dim3(1), 0, (hipStream_t)stream, moments3, mean_out);
  return check_launch("harl_moments_mean");
}

// Row tables of a recurrent minibatch (nets.build_seq, buffer mode): sequence j of the batch covers source rows first[j] + l * stride
// of the t-major flattened buffers (the reference's chunk slicing, on_policy_actor_buffer.py:255-322, and naive whole-column
// sampling, :180-221); padding sequences (j >= m) replay sequence 0.  One launch instead of the eight small torch kernels
// (arange, broadcast add, two gathers, cat ...) every recurrent minibatch cost before -- 45 of them per 8-agent SMAC update, with
// the GPU idle in between.  idx / valid_idx: int64 row indices [L * m_pad] / [L * m]; mask_rows = masks_src[idx]; h0 = h0_src[first].
__global__ __launch_bounds__(256) void k_build_seq(const int64_t *__restrict__ first, int m, int m_pad, int L, long stride,
                                                   const float *__restrict__ masks_src, const float *__restrict__ h0_src, int H,
                                                   int64_t *__restrict__ idx, int64_t *__restrict__ valid_idx,
                                                   float *__restrict__ mask_rows, float *__restrict__ h0) {
  const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x, gn = (long)gridDim.x * blockDim.x;
  const long n_rows = (long)L * m_pad;
  for (long e = gt; e < n_rows; e += gn) {
    const int l = (int)(e / m_pad), j = (int)(e - (long)l * m_pad);
    const long row = first[j < m ? j : 0] + (long)l * stride;
    idx[e] = row;
    mask_rows[e] = masks_src[row];
    if (valid_idx && j < m) valid_idx[(long)l * m + j] = row;
  }
  const long n_h = (long)m_pad * H;
  for (long e = gt; e < n_h; e += gn) {
    const int j = (int)(e / H), f = (int)(e - (long)j * H);
    h0[e] = h0_src[first[j < m ? j : 0] * H + f];
  }
}

extern "C" int harl_build_seq(const int64_t *first, int m, int m_pad, int L, long stride, const float *masks_src,
                              const float *h0_src, int H, int64_t *idx, int64_t *valid_idx, float *mask_rows, float *h0,
                              void *stream) {
  if (m <= 0 || m_pad < m || L <= 0 || H <= 0) {
    set_error("harl_build_seq: bad arguments");
    return -1;
  }
  const long work = (long)L * m_pad > (long)m_pad * H ? (long)L * m_pad : (long)m_pad * H;
  long blocks = (work + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_build_seq, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, first, m, m_pad, L, stride,
                     masks_src, h0_src, H, idx, valid_idx, mask_rows, h0);
  return check_launch("harl_build_seq");
}

// Shader-clock probe (measurement aid of bench.py, no counterpart in the reference): ONE lane waits for `ticks` periods of the
// constant 100 MHz counter (s_memrealtime) and reports how many shader cycles (s_memtime) went by in the meantime.  Launched
// on a side stream next to a training step it gives the clock the chip actually sustains under THAT load (the matrix-pipe
// fraction of the bench line was computed against the nominal 2.4 GHz before).  Bounded: the realtime counter always advances.
__global__ void k_clock_probe(long long *__restrict__ out, long long ticks) {
  if (threadIdx.x != 0) return;
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long c0 = __builtin_readcyclecounter();
  long long r;
  do {
    __builtin_amdgcn_s_sleep(64);
    r = __builtin_amdgcn_s_memrealtime();
  } while (r - r0 < ticks);
  out[0] = __builtin_readcyclecounter() - c0;
  out[1] = r - r0;
}

extern "C" int harl_clock_probe(long long *cycles_ticks, long ticks, void *stream) {
  if (ticks <= 0 || ticks > 100000000L) {  // at most one second
    set_error("harl_clock_probe: ticks must be in (0, 1e8]");
    return -1;
  }
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, cycles_ticks, (long long)ticks);
  return check_launch("harl_clock_probe");
}

// =============================================================================================
// factor *= agg_d exp(new - old)
// =============================================================================================
__global__ __launch_bounds__(256) void k_factor_update(float *__restrict__ factor, const float *__restrict__ nl,
                                                       const float *__restrict__ ol, long n, int D, int agg_mean) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float acc = agg_mean ? 0.f : 1.f;
    for (int d = 0; d < D; ++d) {
      float r = expf(nl[i * D + d] - ol[i * D + d]);
      acc = agg_mean ? acc + r : acc * r;
    }
    if (agg_mean) acc = acc / (float)D;
    factor[i] = factor[i] * acc;
  }
}

extern "C" int harl_factor_update(float *factor, const float *new_logp, const float *old_logp, long n, int act_dim,
                                  int agg_mean, void *stream) {
  if (n <= 0) return 0;
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_factor_update, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, factor, new_logp, old_logp,
                     n, act_dim, agg_mean);
  return check_launch("harl_factor_update");
}

// =============================================================================================
// ValueNorm
// =============================================================================================
__global__ __launch_bounds__(256) void k_sum_sumsq(const float *__restrict__ x, const int64_t *__restrict__ idx, long m,
                                                   double *__restrict__ out2) {
  double s1 = 0, s2 = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long)gridDim.x * blockDim.x) {
    double v = x[idx ? idx[i] : i];
    s1 += v;
    s2 += v * v;
  }
  s1 = wave_reduce_sum_d(s1);
  s2 = wave_reduce_sum_d(s2);
  __shared__ double sh[2][4];
  int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) {
    sh[0][w] = s1;
    sh[1][w] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(&out2[threadIdx.x], sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

extern "C" int harl_sum_sumsq(const float *x, const int64_t *idx, long m, double *sums2, void *stream) {
  if (m <= 0) return 0;
  long nb = (m + 2047) / 2048;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_sum_sumsq, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, idx, m, sums2);
  return check_launch("harl_sum_sumsq");
}

__global__ void k_valuenorm_apply(float *__restrict__ vn, const double *__restrict__ sums2, double count, float w,
                                  float omw) {
#pragma clang fp contract(off)
  if (threadIdx.x == 0 && blockIdx.x == 0) {  // valuenorm.py:47-64
    float bm = (float)(sums2[0] / count);
    float bsq = (float)(sums2[1] / count);
    float rm = vn[0] * w;
    vn[0] = rm + bm * omw;
    float rs = vn[1] * w;
    vn[1] = rs + bsq * omw;
    float db = vn[2] * w;
    vn[2] = db + omw;
  }
}

extern "C" int harl_valuenorm_apply(float *vn_stats, const double *sums2, double count, double beta, void *stream) {
  float w = (float)beta;
  float omw = (float)(1.0 - beta);
  hipLaunchKernelGGL(k_valuenorm_apply, dim3(1), dim3(64), 0, (hipStream_t)stream, vn_stats, sums2, count, w, omw);
  return check_launch("harl_valuenorm_apply");
}

// =============================================================================================
// grad-norm + clip + Adam, one workgroup (P <= a few 100 k elements; launch-latency class)
// =============================================================================================
__global__ __launch_bounds__(1024) void k_gradnorm_clip_adam(float *__restrict__ p, const float *__restrict__ g,
                                                             float *__restrict__ m, float *__restrict__ v, long n,
                                                             const float *__restrict__ grad_scale, int use_clip,
                                                             float max_norm, float lr_over_bc1, float beta1,
                                                             float beta2, float omb1, float omb2, float eps, float wd,
                                                             float bc2_sqrt, double *__restrict__ info_out) {
  const float scale = grad_scale ? *grad_scale : 1.f;
  double ss = 0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    float gi = g[i] * scale;
    ss += (double)gi * gi;
  }
  ss = wave_reduce_sum_d(ss);
  __shared__ double sh[16];
  __shared__ float s_coef;
  int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) sh[w] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    float norm = (float)sqrt(t);
    float coef = 1.f;
    if (use_clip) {  // torch clip_grad_norm_: coef = clamp(max_norm / (total + 1e-6), max=1); grads always scaled
      coef = max_norm / (norm + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;
    }
    s_coef = coef;
    if (info_out) info_out[0] += norm;
  }
  __syncthreads();
  const float coef = s_coef * scale;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    float gi = g[i] * coef;
    float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    float mi = m[i];
    mi = mi + omb1 * (gi - mi);                 // exp_avg.lerp_(grad, 1 - beta1)
    float vi = v[i] * beta2 + (omb2 * gi) * gi;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    m[i] = mi;
    v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - lr_over_bc1 * (mi / denom);
  }
}

extern "C" int harl_gradnorm_clip_adam(float *param, float *grad, float *exp_avg, float *exp_avg_sq, long n,
                                       const float *grad_scale, int use_clip, float max_norm, double lr, double beta1,
                                       double beta2, float eps, float weight_decay, double bias_correction1,
                                       double bias_correction2, double *info_out, void *stream) {
  if (n <= 0) return 0;
  float step_size = (float)(lr / bias_correction1);
  float bc2_sqrt = (float)sqrt(bias_correction2);
  hipLaunchKernelGGL(k_gradnorm_clip_adam, dim3(1), dim3(1024), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, grad_scale, use_clip, max_norm, step_size, (float)beta1, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, bc2_sqrt, info_out);
  return check_launch("harl_gradnorm_clip_adam");
}

// =============================================================================================
// LayerNorm-affine folding and its adjoint
// =============================================================================================
__global__ __launch_bounds__(64) void k_fold_linear(const float *__restrict__ W, const float *__restrict__ b,
                                                    const float *__restrict__ gamma, const float *__restrict__ beta,
                                                    float *__restrict__ Wp, float *__restrict__ bp, int in_dim) {
  int o = blockIdx.x;
  float acc = 0.f;
  for (int k = threadIdx.x; k < in_dim; k += 64) {
    float w = W[(long)o * in_dim + k];
    Wp[(long)o * in_dim + k] = gamma ? w * gamma[k] : w;
    if (beta) acc += w * beta[k];
  }
  acc = wave_reduce_sum(acc);
  if (threadIdx.x == 0) bp[o] = b[o] + acc;
}

extern "C" int harl_fold_linear(const float *W, const float *b, const float *gamma, const float *beta, float *Wp,
                                float *bp, int out_dim, int in_dim, void *stream) {
  hipLaunchKernelGGL(k_fold_linear, dim3(out_dim), dim3(64), 0, (hipStream_t)stream, W, b, gamma, beta, Wp, bp, in_dim);
  return check_launch("harl_fold_linear");
}

// one block per input column k (dgamma/dbeta need a reduction over o), plus row work spread over threads
__global__ __launch_bounds__(64) void k_unfold(const float *__restrict__ dWp, const float *__restrict__ dbp, int ldp,
                                               const float *__restrict__ W, const float *__restrict__ gamma,
                                               const float *__restrict__ beta, float *__restrict__ dW,
                                               float *__restrict__ db, float *__restrict__ dgamma,
                                               float *__restrict__ dbeta, int out_dim, int in_dim, int accumulate) {
  int k = blockIdx.x;
  float g = gamma ? gamma[k] : 1.f;
  float be = beta ? beta[k] : 0.f;
  double sg = 0.0, sb = 0.0;  // double: heavy cancellation over o (see k_reduce_partials_multi)
  for (int o = threadIdx.x; o < out_dim; o += 64) {
    float dwp = dWp[(long)o * ldp + k];
    float dbo = dbp[o];
    float w = W[(long)o * in_dim + k];
    dW[(long)o * in_dim + k] = dwp * g + dbo * be;
    sg += (double)w * (double)dwp;
    sb += (double)w * (double)dbo;
    if (k == 0) db[o] = dbo;
  }
  if (dgamma) {
    sg = wave_reduce_sum_d(sg);
    sb = wave_reduce_sum_d(sb);
    if (threadIdx.x == 0) {  // several Linears may share one LayerNorm (the three GRU gate blocks): accumulate
      dgamma[k] = accumulate ? (float)((double)dgamma[k] + sg) : (float)sg;
      dbeta[k] = accumulate ? (float)((double)dbeta[k] + sb) : (float)sb;
    }
  }
}

extern "C" int harl_unfold_linear_grads(const float *dWp, const float *dbp, int ldp, const float *W,
                                        const float *gamma, const float *beta, float *dW, float *db, float *dgamma,
                                        float *dbeta, int out_dim, int in_dim, int accumulate, void *stream) {
  hipLaunchKernelGGL(k_unfold, dim3(in_dim), dim3(64), 0, (hipStream_t)stream, dWp, dbp, ldp, W, gamma, beta, dW, db,
                     dgamma, dbeta, out_dim, in_dim, accumulate);
  return check_launch("harl_unfold_linear_grads");
}

// deterministic fixed-order reduction of per-workgroup partials
__global__ __launch_bounds__(256) void k_reduce_partials(const float *__restrict__ part, int n_wg, long elems,
                                                         float *__restrict__ out) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int w = 0;
  for (; w + 3 < n_wg; w += 4) {
    s0 += part[(long)(w + 0) * elems + e];
    s1 += part[(long)(w + 1) * elems + e];
    s2 += part[(long)(w + 2) * elems + e];
    s3 += part[(long)(w + 3) * elems + e];
  }
  for (; w < n_wg; ++w) s0 += part[(long)w * elems + e];
  out[e] = (s0 + s1) + (s2 + s3);
}

// =============================================================================================
// Activation functions other than ReLU (harl/utils/models_tools.py:28-50: sigmoid, tanh, leaky_relu, selu -- the ones
// nn.init.calculate_gain accepts, i.e. the ones the reference's MLPLayer can be built with, mlp.py:19-23).  A coverage path
// composed from the verified GEMM kernels in raw mode (harl_mlp_linear / harl_mlp_linear_wide) and two element-wise launches
// over ATL images, one wave per 32-sample slab:
//   harl_act_ln_fwd :  a = act(z),  x_hat = LayerNorm(a)  -> x_hat, mean(a), rstd        (mlp.py:25-38: Linear, act, LayerNorm)
//   harl_act_bwd    :  dz = da * act'(z), in place, with act' taken from the activation VALUE a = x_hat / rstd + mean
//                      (tanh: 1 - a^2, sigmoid: a (1 - a), leaky_relu: a > 0 ? 1 : 0.01, selu: a > 0 ? scale : a + scale alpha)
// where da is what the backward kernels produce when they are handed an all-ones ReLU mask (harl_mlp_bwd_dx, the loss
// kernels' LayerNorm backward).  ACT ids: 1 leaky_relu, 2 tanh, 3 sigmoid, 4 selu.
// =============================================================================================
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f, SELU_ALPHA = 1.6732632423543772848170429916717f;

__device__ __forceinline__ float act_value(float z, int act) {
  switch (act) {
    case 1: return z > 0.f ? z : 0.01f * z;
    case 2: return tanhf(z);
    case 3: return 1.0f / (1.0f + expf(-z));
    default: return z > 0.f ? SELU_SCALE * z : (SELU_SCALE * SELU_ALPHA) * (expf(z) - 1.0f);
  }
}
__device__ __forceinline__ float act_slope_from_value(float a, int act) {
  switch (act) {
    case 1: return a > 0.f ? 1.0f : 0.01f;
    case 2: return 1.0f - a * a;
    case 3: return a * (1.0f - a);
    default: return a > 0.f ? SELU_SCALE : a + SELU_SCALE * SELU_ALPHA;
  }
}

template <int H>
__global__ __launch_bounds__(WG_THREADS) void k_act_ln_fwd(const float *__restrict__ z, long n_slabs, int act,
                                                           float *__restrict__ xhat, float *__restrict__ mean_out,
                                                           float *__restrict__ rstd_out) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float v[H / 2];
    atl_load<H>(z, slab, lane, v);
    float sum = 0.f;
#pragma unroll
    for (int R = 0; R < H / 2; ++R) {
      v[R] = act_value(v[R], act);
      sum += v[R];
    }
    sum = wave_sum32(sum);
    const float mean = sum * (1.0f / H);
    float vs = 0.f;
#pragma unroll
    for (int R = 0; R < H / 2; ++R) {
      v[R] -= mean;
      vs += v[R] * v[R];
    }
    vs = wave_sum32(vs);
    const float rstd = 1.0f / sqrtf(vs * (1.0f / H) + 1e-5f);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) v[R] *= rstd;
    atl_store<H>(xhat, slab, lane, v);
    if (lane < 32) {
      mean_out[slab * SLAB + lane] = mean;
      rstd_out[slab * SLAB + lane] = rstd;
    }
  }
}

template <int H>
__global__ __launch_bounds__(WG_THREADS) void k_act_bwd(float *__restrict__ dz, const float *__restrict__ xhat,
                                                        const float *__restrict__ mean_in, const float *__restrict__ rstd_in,
                                                        long n_slabs, int act) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float d[H / 2], x[H / 2];
    atl_load<H>(dz, slab, lane, d);
    atl_load<H>(xhat, slab, lane, x);
    const float mean = mean_in[slab * SLAB + (lane & 31)], sd = 1.0f / rstd_in[slab * SLAB + (lane & 31)];
#pragma unroll
    for (int R = 0; R < H / 2; ++R) d[R] *= act_slope_from_value(x[R] * sd + mean, act);
    atl_store<H>(dz, slab, lane, d);
  }
}

// forward-mode tangent of [act, LayerNorm] (HATRPO's Fisher-vector product on networks with these activations):
//   a_dot = act'(z) (zd1 + zd2),   x_hat_dot = rstd (a_dot - mean_f(a_dot) - x_hat mean_f(x_hat a_dot))
// zd1 / zd2: the two halves of the pre-activation's tangent (W' x_hat_dot_prev and W'_dot x_hat_prev + b'_dot: two raw GEMMs;
// zd2 may be NULL), act' from the activation value like harl_act_bwd.
template <int H>
__global__ __launch_bounds__(WG_THREADS) void k_act_ln_tangent(const float *__restrict__ zd1, const float *__restrict__ zd2,
                                                               const float *__restrict__ xhat, const float *__restrict__ mean_in,
                                                               const float *__restrict__ rstd_in, long n_slabs, int act,
                                                               float *__restrict__ out) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float d[H / 2], x[H / 2];
    atl_load<H>(zd1, slab, lane, d);
    if (zd2) {
      float e[H / 2];
      atl_load<H>(zd2, slab, lane, e);
#pragma unroll
      for (int R = 0; R < H / 2; ++R) d[R] += e[R];
    }
    atl_load<H>(xhat, slab, lane, x);
    const float mean = mean_in ? mean_in[slab * SLAB + (lane & 31)] : 0.f, rstd = rstd_in[slab * SLAB + (lane & 31)];
    const float sd = 1.0f / rstd;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int R = 0; R < H / 2; ++R) {
      if (act) d[R] *= act_slope_from_value(x[R] * sd + mean, act);
      s1 += d[R];
      s2 += d[R] * x[R];
    }
    s1 = wave_sum32(s1) * (1.0f / H);
    s2 = wave_sum32(s2) * (1.0f / H);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) d[R] = rstd * (d[R] - s1 - x[R] * s2);
    atl_store<H>(out, slab, lane, d);
  }
}

extern "C" int harl_act_ln_tangent(const float *zd1, const float *zd2, const float *xhat, const float *mean, const float *rstd,
                                   long M, int H, int act, float *xhat_dot, void *stream) {
  if (M <= 0) return 0;
  if (act < 0 || act > 4 || (act > 0 && !mean)) {  // 0 = no activation: the LayerNorm tangent alone (rnn.norm of the composed GRU)
    set_error("harl_act_ln_tangent: activation id must be 0 (none), 1 (leaky_relu), 2 (tanh), 3 (sigmoid) or 4 (selu)");
    return -2;
  }
  const long n_slabs = n_slabs_of(M);
  const int grid = persistent_grid(n_slabs, 8);
  if (H == 128) hipLaunchKernelGGL(k_act_ln_tangent<128>, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, zd1, zd2, xhat, mean, rstd, n_slabs, act, xhat_dot);
  else if (H == 64) hipLaunchKernelGGL(k_act_ln_tangent<64>, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, zd1, zd2, xhat, mean, rstd, n_slabs, act, xhat_dot);
  else {
    set_error("harl_act_ln_tangent: width must be 64 or 128");
    return -2;
  }
  return check_launch("harl_act_ln_tangent");
}

extern "C" int harl_act_ln_fwd(const float *z, long M, int H, int act, float *xhat, float *mean, float *rstd, void *stream) {
  if (M <= 0) return 0;
  if (act < 1 || act > 4) {
    set_error("harl_act_ln_fwd: activation id must be 1 (leaky_relu), 2 (tanh), 3 (sigmoid) or 4 (selu)");
    return -2;
  }
  const long n_slabs = n_slabs_of(M);
  const int grid = persistent_grid(n_slabs, 8);
  if (H == 128) hipLaunchKernelGGL(k_act_ln_fwd<128>, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, z, n_slabs, act, xhat, mean, rstd);
  else if (H == 64) hipLaunchKernelGGL(k_act_ln_fwd<64>, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, z, n_slabs, act, xhat, mean, rstd);
  else {
    set_error("harl_act_ln_fwd: width must be 64 or 128");
    return -2;
  }
  return check_launch("harl_act_ln_fwd");
}

extern "C" int harl_act_bwd(float *dz, const float *xhat, const float *mean, const float *rstd, long M, int H, int act,
                            void *stream) {
  if (M <= 0) return 0;
  if (act < 1 || act > 4) {
    set_error("harl_act_bwd: activation id must be 1 (leaky_relu), 2 (tanh), 3 (sigmoid) or 4 (selu)");
    return -2;
  }
  const long n_slabs = n_slabs_of(M);
  const int grid = persistent_grid(n_slabs, 8);
  if (H == 128) hipLaunchKernelGGL(k_act_bwd<128>, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, dz, xhat, mean, rstd, n_slabs, act);
  else if (H == 64) hipLaunchKernelGGL(k_act_bwd<64>, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, dz, xhat, mean, rstd, n_slabs, act);
  else {
    set_error("harl_act_bwd: width must be 64 or 128");
    return -2;
  }
  return check_launch("harl_act_bwd");
}

extern "C" int harl_reduce_partials(const float *part, int n_wg, long elems, float *out, void *stream) {
  if (elems <= 0) return 0;
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                     n_wg, elems, out);
  return check_launch("harl_reduce_partials");
}

__global__ __launch_bounds__(1024) void k_reduce_scalars(const float *__restrict__ ps, int n_blocks,
                                                         double *__restrict__ out) {
  // 16 row-groups x 64 columns; fixed summation order -> deterministic
  __shared__ double sh[16][64];
  const int j = threadIdx.x & 63, rg = threadIdx.x >> 6;
  double s = 0;
  if (j < PS_STRIDE)
    for (int b = rg; b < n_blocks; b += 16) s += (double)ps[(long)b * PS_STRIDE + j];
  sh[rg][j] = s;
  __syncthreads();
  if (threadIdx.x < PS_STRIDE) {
    double t = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sh[g][threadIdx.x];
    out[threadIdx.x] += t;
  }
}

extern "C" int harl_reduce_scalars(const float *part_scalars, int n_blocks, double *scalars, void *stream) {
  hipLaunchKernelGGL(k_reduce_scalars, dim3(1), dim3(1024), 0, (hipStream_t)stream, part_scalars, n_blocks, scalars);
  return check_launch("harl_reduce_scalars");
}

// fp64 loss scalars -> FOUR fp32 pieces on a fixed exponent grid (quanta 2^24, 2^4, 2^-16, 2^-36; every piece is an
// integer multiple n * q_k with |n| < 2^20), written behind the folded gradients so that ONE fp32 SUM all-reduce carries
// gradients and scalars.  Because the grid does not depend on the rank's values, the fp32 sum of each piece over up to 16
// ranks is EXACT (|sum n| < 2^24), so the reduced scalars equal the fp64 sum of the ranks' scalars up to 2^-36 absolute
// per rank -- not merely a per-rank hi/lo pair, whose cross-rank fp32 rounding would be lost.  (|v| >= 2^44 degrades
// gracefully to fp32 accuracy in the first piece.)
constexpr int HILO_PIECES = 4;
__device__ __constant__ double HILO_Q[HILO_PIECES] = {16777216.0, 16.0, 1.0 / 65536.0, 1.0 / 68719476736.0};
__global__ void k_pack_scalars_hilo(const double *__restrict__ scalars, float *__restrict__ hilo) {
  const int t = threadIdx.x;
  if (t < PS_STRIDE) {
    double r = scalars[t];
#pragma unroll
    for (int k = 0; k < HILO_PIECES; ++k) {
      const double p = trunc(r / HILO_Q[k]) * HILO_Q[k];
      hilo[k * PS_STRIDE + t] = (float)p;
      r -= p;
    }
  }
}

extern "C" int harl_pack_scalars_hilo(const double *scalars, float *hilo, void *stream) {
  hipLaunchKernelGGL(k_pack_scalars_hilo, dim3(1), dim3(64), 0, (hipStream_t)stream, scalars, hilo);
  return check_launch("harl_pack_scalars_hilo");
}

// The data-parallel optimiser step needs both, back to back, in front of its collective: the fixed-order fp64 sum of the loss
// kernel's partial rows (k_reduce_scalars, but OVERWRITING `scalars`: no separate clear) and the four fp32 pieces behind the
// folded gradients -- ONE launch instead of a fill + two kernels (each a launch latency on the critical path of every step).
// n_blocks = 0: this rank holds no row of the (global) minibatch: zeros.
__global__ __launch_bounds__(1024) void k_reduce_pack_scalars(const float *__restrict__ ps, int n_blocks, double *__restrict__ out,
                                                             float *__restrict__ hilo) {
  __shared__ double sh[16][64];
  const int j = threadIdx.x & 63, rg = threadIdx.x >> 6;
  double s = 0;
  if (j < PS_STRIDE)
    for (int b = rg; b < n_blocks; b += 16) s += (double)ps[(long)b * PS_STRIDE + j];
  sh[rg][j] = s;
  __syncthreads();
  if (threadIdx.x < PS_STRIDE) {
    double r = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) r += sh[g][threadIdx.x];
    out[threadIdx.x] = r;
#pragma unroll
    for (int k = 0; k < HILO_PIECES; ++k) {
      const double p = trunc(r / HILO_Q[k]) * HILO_Q[k];
      hilo[k * PS_STRIDE + threadIdx.x] = (float)p;
      r -= p;
    }
  }
}

extern "C" int harl_reduce_pack_scalars(const float *part_scalars, int n_blocks, double *scalars, float *hilo, void *stream) {
  if (n_blocks > 0 && !part_scalars) { set_error("harl_reduce_pack_scalars: part_scalars is NULL"); return -2; }
  hipLaunchKernelGGL(k_reduce_pack_scalars, dim3(1), dim3(1024), 0, (hipStream_t)stream, part_scalars, n_blocks, scalars, hilo);
  return check_launch("harl_reduce_pack_scalars");
}

// =============================================================================================
// Layer table (device int32[HARL_TABLE_STRIDE * n_layers]) shared by the fused kernels below.
//   0 w_off  1 b_off  2 gamma_off (-1)  3 beta_off (-1)     offsets into the flat parameter / gradient arena
//   4 out    5 in     6 pack_w_off      7 pack_b_off          offsets into the folded-weight arena
//   8 dwp_off (dense folded gradient: dWp[op][kp] then dbp[op])  9 kp  10 op  11 part_off (per-WG partials arena)
// =============================================================================================
constexpr int TS = 12;

// ---------------------------------------------------------------------------------------------
// Table-driven forms of harl_unfold_linear_grads / harl_fold_linear_tangent: every entry of the layer table in ONE launch.
// HATRPO evaluates ~190 Fisher-vector products per 17-agent update, each of which folds a tangent and unfolds a gradient
// entry by entry (4 + 4 launches of ~5 us: 1 560 launches per update); same arithmetic, same summation order, same bits.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_unfold_table(const float *__restrict__ p, float *__restrict__ g,
                                                     const float *__restrict__ dwp, const int *__restrict__ tab, int n_layers) {
  int k = blockIdx.x, l = 0;
  for (; l < n_layers; ++l) {  // columns of all entries, concatenated
    if (k < tab[l * TS + 5]) break;
    k -= tab[l * TS + 5];
  }
  if (l >= n_layers) return;
  const int *t = tab + l * TS;
  const int O = t[4], K = t[5], kp = t[9], op = t[10], go = t[2], beo = t[3];
  const float *dW_ = dwp + t[8];
  const float *db_ = dW_ + (long)op * kp;
  const float gam = go >= 0 ? p[go + k] : 1.f, bet = beo >= 0 ? p[beo + k] : 0.f;
  for (int o = threadIdx.x; o < O; o += 64) {
    const float d = dW_[(long)o * kp + k], dbo = db_[o];
    g[t[0] + (long)o * K + k] = d * gam + dbo * bet;
    if (k == 0) g[t[1] + o] = dbo;
  }
  if (go < 0) return;
  for (int l2 = 0; l2 < l; ++l2)
    if (tab[l2 * TS + 2] == go) return;  // an earlier entry owns this LayerNorm's gradients
  // owner: this entry and the later ones that share the LayerNorm (the three GRU gate blocks), in table order, each sum
  // rounded to float before the next is added -- the accumulate form of k_unfold
  float accg = 0.f, accb = 0.f;
  bool first = true;
  for (int l2 = l; l2 < n_layers; ++l2) {
    const int *t2 = tab + l2 * TS;
    if (t2[2] != go) continue;
    const int O2 = t2[4], K2 = t2[5], kp2 = t2[9], op2 = t2[10];
    const float *dW2 = dwp + t2[8];
    const float *db2 = dW2 + (long)op2 * kp2;
    double sg = 0.0, sb = 0.0;  // double: heavy cancellation over o (see k_reduce_partials_multi)
    for (int o = threadIdx.x; o < O2; o += 64) {
      const float w = p[t2[0] + (long)o * K2 + k];
      sg += (double)w * (double)dW2[(long)o * kp2 + k];
      sb += (double)w * (double)db2[o];
    }
    sg = wave_reduce_sum_d(sg);
    sb = wave_reduce_sum_d(sb);
    accg = first ? (float)sg : (float)((double)accg + sg);
    accb = first ? (float)sb : (float)((double)accb + sb);
    first = false;
  }
  if (threadIdx.x == 0) {
    g[go + k] = accg;
    g[beo + k] = accb;
  }
}

extern "C" int harl_unfold_table(const float *param, float *grad, const float *dwp, const int *table, int n_layers,
                                 int total_cols, void *stream) {
  if (n_layers <= 0 || total_cols <= 0) return 0;
  hipLaunchKernelGGL(k_unfold_table, dim3(total_cols), dim3(64), 0, (hipStream_t)stream, param, grad, dwp, table, n_layers);
  return check_launch("harl_unfold_table");
}

// every entry of the layer table in ONE launch (one block per output row, k_fold_linear's arithmetic): the fold at the head of
// every update was one launch per Linear -- 11 per network of the 8-agent recurrent workload, 122 per update
__global__ __launch_bounds__(64) void k_fold_table(const float *__restrict__ p, float *__restrict__ packs,
                                                   const int *__restrict__ tab, int n_layers) {
  int o = blockIdx.x, l = 0;
  for (; l < n_layers; ++l) {
    if (o < tab[l * TS + 4]) break;
    o -= tab[l * TS + 4];
  }
  if (l >= n_layers) return;
  const int *t = tab + l * TS;
  const int in_dim = t[5];
  const float *W = p + t[0], *b = p + t[1];
  const float *gamma = t[2] >= 0 ? p + t[2] : nullptr, *beta = t[3] >= 0 ? p + t[3] : nullptr;
  float *Wp = packs + t[6], *bp = packs + t[7];
  float acc = 0.f;
  for (int k = threadIdx.x; k < in_dim; k += 64) {
    float w = W[(long)o * in_dim + k];
    Wp[(long)o * in_dim + k] = gamma ? w * gamma[k] : w;
    if (beta) acc += w * beta[k];
  }
  acc = wave_reduce_sum(acc);
  if (threadIdx.x == 0) bp[o] = b[o] + acc;
}

extern "C" int harl_fold_table(const float *param, float *packs, const int *table, int n_layers, int total_rows, void *stream) {
  if (n_layers <= 0 || total_rows <= 0) return 0;
  hipLaunchKernelGGL(k_fold_table, dim3(total_rows), dim3(64), 0, (hipStream_t)stream, param, packs, table, n_layers);
  return check_launch("harl_fold_table");
}

// tangent of the LayerNorm-affine fold of every entry:  Wp_dot = W_dot*g + W*g_dot ;  bp_dot = b_dot + W_dot.beta + W.beta_dot
// (vec = the tangent direction in the flat parameter layout; one block per output row of every entry)
__global__ __launch_bounds__(64) void k_fold_tangent_table(const float *__restrict__ p, const float *__restrict__ vec,
                                                           float *__restrict__ packd, const int *__restrict__ tab, int n_layers) {
  int o = blockIdx.x, l = 0;
  for (; l < n_layers; ++l) {
    if (o < tab[l * TS + 4]) break;
    o -= tab[l * TS + 4];
  }
  if (l >= n_layers) return;
  const int *t = tab + l * TS;
  const int K = t[5], go = t[2], beo = t[3];
  const float *W = p + t[0] + (long)o * K, *Wd = vec + t[0] + (long)o * K;
  float *Wpd = packd + t[6] + (long)o * K;
  float acc = 0.f;
  for (int k = threadIdx.x; k < K; k += 64) {
    const float w = W[k], wd = Wd[k];
    Wpd[k] = go >= 0 ? wd * p[go + k] + w * vec[go + k] : wd;
    if (beo >= 0) acc += wd * p[beo + k] + w * vec[beo + k];
  }
  acc = wave_reduce_sum(acc);
  if (threadIdx.x == 0) packd[t[7] + o] = vec[t[1] + o] + acc;
}

extern "C" int harl_fold_tangent_table(const float *param, const float *vec, float *pack_d, const int *table, int n_layers,
                                       int total_rows, void *stream) {
  if (n_layers <= 0 || total_rows <= 0) return 0;
  hipLaunchKernelGGL(k_fold_tangent_table, dim3(total_rows), dim3(64), 0, (hipStream_t)stream, param, vec, pack_d, table,
                     n_layers);
  return check_launch("harl_fold_tangent_table");
}

// out[dwp_off_l + e] = sum_w part[part_off_l + w * elems_l + e]   for every layer segment, ONE launch.
// A block owns 64 consecutive elements; its four waves each sum a quarter of the partial rows (w = 4 k + wave) with
// 8 independent accumulators, and the four sums are combined in fixed order through LDS: the kernel is a chain of
// dependent-latency-bound strided loads, so it is n_wg / 32 loads deep instead of n_wg / 8 (37 -> ~15 us at 512 rows).
#ifndef HARL_REDUCE_DEPTH
#define HARL_REDUCE_DEPTH 16
#endif
__global__ __launch_bounds__(256) void k_reduce_partials_multi(const float *__restrict__ part, const int *__restrict__ tab,
                                                               int n_layers, int n_wg, float *__restrict__ dwp) {
  // The per-workgroup partials are summed in DOUBLE (fixed order): the LayerNorm-affine gradients are later formed as
  // sum_o W[o][k] dWp[o][k] -- a dot product with heavy cancellation -- so rounding of dWp is amplified ~sqrt(H) times in
  // them; the rows are fp32 sums over a few thousand samples each, the cross-row sum adds nothing to that.
  __shared__ double sh[4][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  long e = (long)blockIdx.x * 64 + lane;
  double total = 0.0;
  long out_idx = -1;
  for (int l = 0; l < n_layers; ++l) {
    const int *t = tab + l * TS;
    const long elems = (long)t[10] * t[9] + t[10];
    if (e < elems) {
      const float *p = part + (long)t[11] + e;  // part_off is in floats
      // RD independent loads per lane in flight (8 until round 6 session 3: ~3 MB in flight over the chip, 2 TB/s at the ~2 us
      // these strided reads take; HARL_REDUCE_DEPTH A/B: profiles/r06s3_reduce_depth_ab.txt)
      constexpr int RD = HARL_REDUCE_DEPTH;
      double acc[RD];
#pragma unroll
      for (int u = 0; u < RD; ++u) acc[u] = 0.0;
      int w = rg;
      for (; w + 4 * (RD - 1) < n_wg; w += 4 * RD) {
        float v[RD];
#pragma unroll
        for (int u = 0; u < RD; ++u) v[u] = p[(long)(w + 4 * u) * elems];
#pragma unroll
        for (int u = 0; u < RD; ++u) acc[u] += (double)v[u];
      }
      for (; w < n_wg; w += 4) acc[0] += (double)p[(long)w * elems];
#pragma unroll
      for (int st = 1; st < RD; st *= 2)
#pragma unroll
        for (int u = 0; u + st < RD; u += 2 * st) acc[u] += acc[u + st];
      total = acc[0];
      out_idx = t[8] + e;
      break;
    }
    e -= elems;
  }
  sh[rg][lane] = total;
  __syncthreads();
  if (rg == 0 && out_idx >= 0) dwp[out_idx] = (float)((sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]));
}

extern "C" int harl_reduce_partials_multi(const float *part, const int *table, int n_layers, int n_wg, long total_elems,
                                          float *dwp, void *stream) {
  if (total_elems <= 0) return 0;
  hipLaunchKernelGGL(k_reduce_partials_multi, dim3((unsigned)((total_elems + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, part, table, n_layers, n_wg, dwp);
  return check_launch("harl_reduce_partials_multi");
}

// ---------------------------------------------------------------------------------------------
// Fused optimiser epilogue: ADAM_WGS co-resident workgroups, ONE software grid barrier.
//   1  loss scalars (optionally reduced here from the loss kernel's per-block partials); unfold the folded gradients of every
//      table entry into `grad`; the LayerNorm-affine gradients; each workgroup's share of sum g^2                -- barrier --
//   2  gradient scale + training statistics, ||g||, clip; then BY ROWS of the table entries (one wave per output row): Adam on
//      the row's weights and bias and, from the updated values still in registers, the re-folded row W' = W gamma, b' = b + W.beta
//      for the next forward.  The LayerNorm parameters a row needs are updated REDUNDANTLY in registers by every wave that
//      needs them (same inputs, same arithmetic, same bits) and written back once, by the last workgroup to finish -- no
//      barrier between Adam and the re-fold, none between ||g|| and Adam.
// Round 3 ran  unfold | barrier | ||g|| | barrier | Adam | barrier | re-fold : 42 us per launch of pure latency, now 30.
// (Measured and NOT kept in round 4: the split-K combine of the weight-gradient partials as a phase 0 of this launch -- 51 MB
// through 64 workgroups: 103 us against 17 + 30 for the two launches; with 256 workgroups the barriers cost more than the
// launch they save: 123 us.)
// Replaces (per update) ~25 tiny launches of the reference's ATen path: scalar reduce, unfold x L, reciprocal/cast/copy glue,
// grad-norm, Adam, fold x L.  64 workgroups x 256 threads are co-resident on 256 CUs (a few registers each), which is what
// makes the spin barrier safe; other kernels on the chip only delay their arrival.
// ws: [0] arrivals, [1] finished, then doubles from byte 32: [w] per-workgroup sum of squares, [G + r*48 + j] scalar row sums
// ---------------------------------------------------------------------------------------------
#ifndef HARL_ADAM_WGS
#define HARL_ADAM_WGS 64
#endif
constexpr int ADAM_WGS = HARL_ADAM_WGS, ADAM_THREADS = 256, ADAM_SROWS = ADAM_WGS < 64 ? ADAM_WGS : 64;  // (HARL_ADAM_WGS: A/B builds, tools/gpurun_calls_r06_s3n.sh)
static_assert(ADAM_WGS >= ADAM_SROWS && ADAM_WGS % 4 == 0 && 64 + ADAM_WGS * 8 + ADAM_SROWS * PS_STRIDE * 8 <= 32768, "workspace layout");

__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
#ifdef HARL_BARRIER_V1  // A/B: two full fences, release RMW, acquire polling (rounds 3 - 6: ~10 us of fences per barrier)
    __threadfence();  // release: this workgroup's global writes precede the arrival
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    __threadfence();  // acquire: invalidates this CU's vector L1 for the whole workgroup
#else
    // MI355X_MICROARCH.md, "Valid forms": ONE agent release (write back the XCD L2's dirty lines, ~1.7 us) -> explicit
    // s_waitcnt (ROCm 7.2 may drop the one behind buffer_wbl2) -> relaxed arrival; relaxed polls (an acquire load costs a
    // buffer_inv per iteration) -> ONE agent acquire (invalidates this CU's vector L1 for the whole workgroup)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  __syncthreads();
}

struct AdamConsts {
  float coef, lr_over_bc1, omb1, beta2, omb2, eps, wd, bc2_sqrt;
};
// torch.optim.Adam's single-tensor step on one element; returns the new parameter
__device__ __forceinline__ float adam_elem(float g_raw, float pi, float &mi, float &vi, const AdamConsts &c) {
  float gi = g_raw * c.coef;
  if (c.wd != 0.f) gi = gi + c.wd * pi;
  mi = mi + c.omb1 * (gi - mi);
  vi = vi * c.beta2 + (c.omb2 * gi) * gi;
  return pi - c.lr_over_bc1 * (mi / (sqrtf(vi) / c.bc2_sqrt + c.eps));
}

__global__ __launch_bounds__(ADAM_THREADS) void k_adam_fold(
    float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long n,
    const float *__restrict__ dwp, const int *__restrict__ tab, int n_layers,
    float *__restrict__ packs, double *__restrict__ scalars, const float *__restrict__ part_scalars, int n_scalar_blocks,
    const float *__restrict__ scalars_hilo, int mode, float const_scale, int logstd_off, int act_dim, double *__restrict__ info, int use_clip, float max_norm,
    float lr_over_bc1, float beta1, float beta2, float omb1, float omb2, float eps, float wd, float bc2_sqrt,
    unsigned *__restrict__ ws) {
  __shared__ double sh[64];
  const int tid = threadIdx.x, nt = ADAM_THREADS;
  const int G = gridDim.x, blk = blockIdx.x;
  const long gtid = (long)blk * nt + tid, gnt = (long)G * nt;
  const int gw = (int)(gtid >> 6), ln = tid & 63, gnw = (int)(gnt >> 6);  // global wave id / lane / wave count
  double *ws_part = reinterpret_cast<double *>(ws + 8);
  double *ws_rows = ws_part + ADAM_WGS;  // [ADAM_SROWS][PS_STRIDE]
  unsigned bar_target = 0;

  // ---- phase 1.0: loss-kernel partial rows -> ADAM_SROWS row sums (workgroup b < ADAM_SROWS takes rows b, b + 64, ...)
  if (part_scalars && blk < ADAM_SROWS && tid < PS_STRIDE) {
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    int b = blk;
    for (; b + 3 * ADAM_SROWS < n_scalar_blocks; b += 4 * ADAM_SROWS) {
      t0 += (double)part_scalars[(long)b * PS_STRIDE + tid];
      t1 += (double)part_scalars[(long)(b + ADAM_SROWS) * PS_STRIDE + tid];
      t2 += (double)part_scalars[(long)(b + 2 * ADAM_SROWS) * PS_STRIDE + tid];
      t3 += (double)part_scalars[(long)(b + 3 * ADAM_SROWS) * PS_STRIDE + tid];
    }
    for (; b < n_scalar_blocks; b += ADAM_SROWS) t0 += (double)part_scalars[(long)b * PS_STRIDE + tid];
    ws_rows[blk * PS_STRIDE + tid] = (t0 + t1) + (t2 + t3);
  }
  // ---- phase 1.1: unfold  dW = dWp*gamma + dbp (x) beta ; db = dbp ; this thread's share of sum g^2 (unscaled)
  double ss = 0.0;
  for (int l = 0; l < n_layers; ++l) {
    const int *t = tab + l * TS;
    const int O = t[4], K = t[5], kp = t[9], op = t[10];
    const float *dW_ = dwp + t[8];
    const float *db_ = dW_ + (long)op * kp;
    const float *gam = t[2] >= 0 ? p + t[2] : nullptr;
    const float *bet = t[3] >= 0 ? p + t[3] : nullptr;
    for (long e = gtid; e < (long)O * K; e += gnt) {
      const int o = (int)(e / K), k = (int)(e - (long)o * K);
      const float dwp_ = dW_[(long)o * kp + k];
      const float gv = gam ? dwp_ * gam[k] + db_[o] * bet[k] : dwp_;
      g[t[0] + e] = gv;
      ss += (double)gv * (double)gv;
    }
    for (long o = gtid; o < O; o += gnt) {
      const float gv = db_[o];
      g[t[1] + o] = gv;
      ss += (double)gv * (double)gv;
    }
  }
  // ---- phase 1.2: dgamma[k] = sum_{entries sharing gamma} sum_o W[o][k] dWp[o][k] ; dbeta[k] likewise with dbp[o].
  // One wave per (owner entry, column): lanes sweep the rows, then the entries that share the LayerNorm (the three
  // GRU gate blocks) in table order -- fixed order, no atomics, no zero-fill.
  {
    int wslot = 0;
    for (int l = 0; l < n_layers; ++l) {
      const int *t = tab + l * TS;
      const int go = t[2];
      if (go < 0) continue;
      bool owner = true;
      for (int l2 = 0; l2 < l; ++l2) owner = owner && (tab[l2 * TS + 2] != go);
      if (!owner) continue;
      const int K = t[5];
      for (int k = gw - wslot; k < K; k += gnw) {
        if (k < 0) continue;
        double sg = 0.0, sb = 0.0;  // double: these dot products cancel heavily (see k_reduce_partials_multi)
        for (int l2 = l; l2 < n_layers; ++l2) {
          const int *t2 = tab + l2 * TS;
          if (t2[2] != go) continue;
          const int O = t2[4], kp = t2[9], op = t2[10];
          const float *dW_ = dwp + t2[8];
          const float *db_ = dW_ + (long)op * kp;
          for (int o = ln; o < O; o += 64) {
            const double w = p[t2[0] + o * K + k];
            sg += w * (double)dW_[(long)o * kp + k];
            sb += w * (double)db_[o];
          }
        }
        sg = wave_reduce_sum_d(sg);
        sb = wave_reduce_sum_d(sb);
        if (ln == 0) {
          const float fg = (float)sg, fb = (float)sb;
          g[go + k] = fg;
          g[t[3] + k] = fb;
          ss += (double)fg * (double)fg + (double)fb * (double)fb;
        }
      }
      wslot = (wslot + K) % gnw;  // spread the columns of successive LayerNorms over different waves
    }
  }
  {
    ss = wave_reduce_sum_d(ss);
    if (ln == 0) sh[48 + (tid >> 6)] = ss;
    __syncthreads();
    if (tid == 0) ws_part[blk] = (sh[48] + sh[49]) + (sh[50] + sh[51]);
  }
  bar_target += (unsigned)G;
  grid_barrier(ws, bar_target);
  // ---- phase 2.0: every workgroup: the scalar sums (same fixed order everywhere); workgroup 0 publishes them
  if (tid < PS_STRIDE) {
    double t = 0;
    if (part_scalars) {
      double u0 = 0, u1 = 0, u2 = 0, u3 = 0;
      for (int b = 0; b < ADAM_SROWS; b += 4) {
        u0 += ws_rows[(b + 0) * PS_STRIDE + tid];
        u1 += ws_rows[(b + 1) * PS_STRIDE + tid];
        u2 += ws_rows[(b + 2) * PS_STRIDE + tid];
        u3 += ws_rows[(b + 3) * PS_STRIDE + tid];
      }
      t = (u0 + u1) + (u2 + u3);
      if (blk == 0) scalars[tid] = t;
    } else if (scalars_hilo) {  // all-reduced fixed-grid fp32 pieces (harl_pack_scalars_hilo), smallest first
      t = (((double)scalars_hilo[3 * PS_STRIDE + tid] + (double)scalars_hilo[2 * PS_STRIDE + tid]) +
           (double)scalars_hilo[PS_STRIDE + tid]) + (double)scalars_hilo[tid];
      if (blk == 0) scalars[tid] = t;
    } else {
      t = scalars[tid];
    }
    sh[tid] = t;
  }
  __syncthreads();
  float scale;
  if (mode == 0) {  // actor: loss = sum / sum(active)   (happo.py:77-85)
    scale = (float)(1.0 / sh[1]);
    if (blk == 0 && tid == 0 && info) {
      info[0] += (float)(sh[0] / sh[1]);
      info[1] += (float)(sh[2] / sh[1]);
      info[3] += (float)(sh[3] / sh[4]);
    }
  } else {  // critic: mean over the (global) minibatch, times value_loss_coef (v_critic.py:112,146)
    scale = const_scale;
    if (blk == 0 && tid == 0 && info) info[0] += (float)(sh[0] / sh[1]);
  }
  const bool has_ls = logstd_off >= 0;
  // ---- phase 2.1: ||g * scale||  (log_std gradients come straight from the scalar sums), clip coefficient
  AdamConsts C;
  {
    double t = 0;
    for (int b = 0; b < G; b += 4) t += (ws_part[b] + ws_part[b + 1]) + (ws_part[b + 2] + ws_part[b + 3]);  // same order in every workgroup
    if (has_ls)
      for (int d = 0; d < act_dim; ++d) {
        const float gl = (float)sh[8 + d];
        t += (double)gl * (double)gl;
      }
    const float norm = (float)(sqrt(t) * fabs((double)scale));
    float coef = 1.f;
    if (use_clip) {
      coef = max_norm / (norm + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;
    }
    if (blk == 0 && tid == 0 && info) info[2 - mode] += norm;  // actor: info[2] ; critic: info[1]
    C = AdamConsts{coef * scale, lr_over_bc1, omb1, beta2, omb2, eps, wd, bc2_sqrt};
  }
  if (has_ls && blk == 0 && tid < act_dim) {  // log_std: no fold depends on it
    const int i = logstd_off + tid;
    const float gl = (float)sh[8 + tid];
    g[i] = gl;
    float mi = m[i], vi = v[i];
    p[i] = adam_elem(gl, p[i], mi, vi, C);
    m[i] = mi;
    v[i] = vi;
  }
  // ---- phase 2.2: one wave per output row of every entry: Adam on W[o][:], b[o]; re-fold the row from the updated values.
  // gamma / beta of the entry's LayerNorm: updated here in registers only (every wave that needs them: same bits), written
  // back by the last workgroup below -- their stored values are read by other waves until then.
  {
    int rbase = 0;
    for (int l = 0; l < n_layers; ++l) {
      const int *t = tab + l * TS;
      const int O = t[4], K = t[5], go = t[2], beo = t[3];
      for (int o = gw - (rbase % gnw); o < O; o += gnw) {
        if (o < 0) continue;
        float acc = 0.f;
        for (int k = ln; k < K; k += 64) {
          const long i = t[0] + (long)o * K + k;
          float mi = m[i], vi = v[i];
          const float w = adam_elem(g[i], p[i], mi, vi, C);
          p[i] = w;
          m[i] = mi;
          v[i] = vi;
          float wp = w;
          if (go >= 0) {
            float mg = m[go + k], vg = v[go + k];
            wp = w * adam_elem(g[go + k], p[go + k], mg, vg, C);
            float mb = m[beo + k], vb = v[beo + k];
            acc += w * adam_elem(g[beo + k], p[beo + k], mb, vb, C);
          }
          packs[t[6] + (long)o * K + k] = wp;
        }
        acc = wave_reduce_sum(acc);
        if (ln == 0) {
          const long i = t[1] + o;
          float mi = m[i], vi = v[i];
          const float bnew = adam_elem(g[i], p[i], mi, vi, C);
          p[i] = bnew;
          m[i] = mi;
          v[i] = vi;
          packs[t[7] + o] = bnew + acc;
        }
      }
      rbase += O;
    }
  }
  // ---- last workgroup out: write back the LayerNorm parameters (every reader is done), reset the barrier words
  __syncthreads();
  __shared__ unsigned last_flag;
  if (tid == 0) {
#ifdef HARL_BARRIER_V1
    __threadfence();
    const unsigned done = __hip_atomic_fetch_add(ws + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
    // a write-after-read hand-off: the last workgroup overwrites LayerNorm parameters the others have READ (their loads are
    // complete: the values went into the stores above) and needs nothing they wrote after the grid barrier -- a relaxed
    // ticket is enough (the full fence + acq_rel RMW were ~5 us at the tail of every optimiser step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned done = __hip_atomic_fetch_add(ws + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    last_flag = done == (unsigned)G - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (last_flag) {
    for (int l = 0; l < n_layers; ++l) {
      const int *t = tab + l * TS;
      const int go = t[2], beo = t[3], K = t[5];
      if (go < 0) continue;
      bool owner = true;
      for (int l2 = 0; l2 < l; ++l2) owner = owner && (tab[l2 * TS + 2] != go);
      if (!owner) continue;
      for (int k = tid; k < K; k += nt) {
        float mg = m[go + k], vg = v[go + k];
        p[go + k] = adam_elem(g[go + k], p[go + k], mg, vg, C);
        m[go + k] = mg;
        v[go + k] = vg;
        float mb = m[beo + k], vb = v[beo + k];
        p[beo + k] = adam_elem(g[beo + k], p[beo + k], mb, vb, C);
        m[beo + k] = mb;
        v[beo + k] = vb;
      }
    }
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ws + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

extern "C" int harl_adam_fold(float *param, float *grad, float *exp_avg, float *exp_avg_sq, long n, const float *dwp,
                              const int *table, int n_layers, float *packs, double *scalars,
                              const float *part_scalars, int n_scalar_blocks, const float *scalars_hilo, int mode,
                              float const_scale, int logstd_off, int act_dim, double *info, int use_clip, float max_norm,
                              double lr, double beta1, double beta2, float eps, float weight_decay, double bias_correction1,
                              double bias_correction2, void *ws, void *stream) {
  if (!ws) { set_error("harl_adam_fold: workspace (>= 32 KiB, zero-initialised once) is required"); return -2; }
  const float step_size = (float)(lr / bias_correction1);
  const float bc2_sqrt = (float)sqrt(bias_correction2);
  hipLaunchKernelGGL(k_adam_fold, dim3(ADAM_WGS), dim3(ADAM_THREADS), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, dwp, table, n_layers, packs, scalars, part_scalars, n_scalar_blocks,
                     scalars_hilo, mode, const_scale, logstd_off, act_dim, info, use_clip, max_norm, step_size, (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, bc2_sqrt, (unsigned *)ws);
  return check_launch("harl_adam_fold");
}

// =============================================================================================
// HATRPO host-side vector algebra, fused (harl/utils/trpo_util.py:96-158).  The conjugate-gradient loop and the epilogue of
// the Fisher-vector product are a dozen P-sized torch ops each: ~6 500 tiny launches per 17-agent update.  One launch each:
//   harl_trpo_fvp_finish : out = grad / m  (log_std block: 2 (dsigma/dls)^2 / sigma^2 * vec)  + damping * vec
//   harl_trpo_cg_step    : alpha = done ? 0 : rdotr / (p . avp);  x += alpha p;  r -= alpha avp;  new = r . r;
//                          p = done ? p : r + (new / rdotr) p;  rdotr = new;  done |= rdotr < 1e-10
// Element-wise arithmetic in the reference's fp32 operation order (contraction off); dot products accumulate in fp64.
// =============================================================================================
__global__ __launch_bounds__(256) void k_trpo_fvp_finish(const float *__restrict__ grad, const float *__restrict__ vec,
                                                         const float *__restrict__ log_std, float *__restrict__ out, long n,
                                                         float m_global, float damping, long ls_off, int act_dim,
                                                         float xc, float yc) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float o;
    if (ls_off >= 0 && i >= ls_off && i < ls_off + act_dim) {
      const float ls = log_std[i - ls_off];
      const float sg = 1.0f / (1.0f + expf(-(ls / xc)));
      const float sigma = sg * yc;
      const float dsig = yc * sg * (1.0f - sg) / xc;
      o = (2.0f * dsig * dsig / (sigma * sigma)) * vec[i];
    } else {
      o = grad[i] / m_global;
    }
    const float t = damping * vec[i];
    out[i] = o + t;
  }
}

extern "C" int harl_trpo_fvp_finish(const float *grad, const float *vec, const float *log_std, float *out, long n,
                                    float m_global, float damping, long logstd_off, int act_dim, float std_x_coef,
                                    float std_y_coef, void *stream) {
  if (n <= 0) return 0;
  long nb = (n + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_trpo_fvp_finish, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, grad, vec, log_std, out, n,
                     m_global, damping, log_std ? logstd_off : -1, act_dim, std_x_coef, std_y_coef);
  return check_launch("harl_trpo_fvp_finish");
}


// CG_WGS co-resident workgroups with a software grid barrier after each of the two dot products (the scheme of k_adam_fold):
// every thread keeps its 5-6 elements of x, r, p, F p in registers across the three phases -- each vector is read once and
// written once.  Round 3 ran this as ONE workgroup of 1024 threads looping 82 times over dependent loads: 76 us per call, 170
// calls per 17-agent update (5 % of it); now ~15 us.  Partial sums are combined in fixed order by every workgroup: same bits
// every run.
constexpr int CG_WGS = 64, CG_THREADS = 256, CG_PER = 8;  // up to 64 x 256 x 8 = 131 072 elements in registers; more: strided tail
struct CgScratch {
  unsigned bar[8];
  double part[2][CG_WGS];
};

__global__ __launch_bounds__(CG_THREADS) void k_trpo_cg_step(float *__restrict__ x, float *__restrict__ r, float *__restrict__ p,
                                                             const float *__restrict__ avp, long n, float *__restrict__ state,
                                                             CgScratch *__restrict__ ws) {
#pragma clang fp contract(off)
  __shared__ double sh[CG_THREADS / 64];
  const float rdotr = state[0];
  // the reference leaves its loop once rdotr < 1e-10 (trpo_util.py:127-128): later launches are true no-ops (uniform over the
  // whole grid: `state` is only written after the second barrier; multiplying the idle Fisher-vector product by alpha = 0
  // instead would turn a non-finite entry of it into NaN)
  if (state[1] != 0.f) return;
  const int tid = threadIdx.x, blk = blockIdx.x;
  const long gtid = (long)blk * CG_THREADS + tid, gnt = (long)CG_WGS * CG_THREADS;
  auto block_sum = [&](double v) -> double {  // -> every thread of the workgroup
    v = wave_reduce_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < CG_THREADS / 64; ++w) t += sh[w];
    return t;
  };
  auto grid_sum = [&](double v, int which, unsigned target) -> double {
    const double t = block_sum(v);
    if (tid == 0) ws->part[which][blk] = t;
    grid_barrier(ws->bar, target);
    double g = 0;
    for (int b2 = 0; b2 < CG_WGS; ++b2) g += ws->part[which][b2];  // same order in every workgroup
    return g;
  };
  float pv[CG_PER], av[CG_PER];
  double d = 0.0;
#pragma unroll
  for (int k = 0; k < CG_PER; ++k) {
    const long i = gtid + k * gnt;
    pv[k] = i < n ? p[i] : 0.f;
    av[k] = i < n ? avp[i] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < CG_PER; ++k) d += (double)pv[k] * (double)av[k];
  for (long i = gtid + CG_PER * gnt; i < n; i += gnt) d += (double)p[i] * (double)avp[i];  // (vectors beyond 131 072 entries)
  const float pavp = (float)grid_sum(d, 0, (unsigned)CG_WGS);
  const float alpha = rdotr / pavp;
  float rv[CG_PER];
  double rr = 0.0;
#pragma unroll
  for (int k = 0; k < CG_PER; ++k) {
    const long i = gtid + k * gnt;
    if (i < n) {
      const float ap = alpha * pv[k];
      x[i] = x[i] + ap;
      const float aa = alpha * av[k];
      const float rn = r[i] - aa;
      r[i] = rn;
      rv[k] = rn;
      rr += (double)rn * (double)rn;
    } else {
      rv[k] = 0.f;
    }
  }
  for (long i = gtid + CG_PER * gnt; i < n; i += gnt) {
    const float ap = alpha * p[i];
    x[i] = x[i] + ap;
    const float aa = alpha * avp[i];
    const float rn = r[i] - aa;
    r[i] = rn;
    rr += (double)rn * (double)rn;
  }
  const float new_rdotr = (float)grid_sum(rr, 1, (unsigned)(2 * CG_WGS));
  const float beta = new_rdotr / rdotr;
#pragma unroll
  for (int k = 0; k < CG_PER; ++k) {
    const long i = gtid + k * gnt;
    if (i < n) {
      const float bp = beta * pv[k];
      p[i] = rv[k] + bp;
    }
  }
  for (long i = gtid + CG_PER * gnt; i < n; i += gnt) {
    const float bp = beta * p[i];
    p[i] = r[i] + bp;
  }
  // last workgroup out: publish the new residual, reset the barrier words for the next launch
  __syncthreads();
  if (tid == 0) {
    const unsigned done = __hip_atomic_fetch_add(ws->bar + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)CG_WGS - 1) {
      state[0] = new_rdotr;
      state[1] = new_rdotr < 1e-10f ? 1.f : 0.f;
      __hip_atomic_store(ws->bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ws->bar + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

static_assert(sizeof(CgScratch) <= HARL_CG_SCRATCH_BYTES, "HARL_CG_SCRATCH_BYTES too small");

// `scratch`: HARL_CG_SCRATCH_BYTES of caller-owned device memory, ZERO-FILLED once before the first launch that uses it (the
// kernel leaves its barrier words at zero itself: last workgroup out), one block per stream with launches in flight.
extern "C" int harl_trpo_cg_step(float *x, float *r, float *p, const float *avp, long n, float *state, void *scratch, void *stream) {
  if (n <= 0) return 0;
  CgScratch *ws = static_cast<CgScratch *>(scratch);
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 7) != 0) {
    set_error("harl_trpo_cg_step: scratch must be HARL_CG_SCRATCH_BYTES of zero-filled, 8-byte aligned device memory");
    return -2;
  }
  hipLaunchKernelGGL(k_trpo_cg_step, dim3(CG_WGS), dim3(CG_THREADS), 0, (hipStream_t)stream, x, r, p, avp, n, state, ws);
  return check_launch("harl_trpo_cg_step");
}

// =============================================================================================
// HATRPO's scalar glue on the device (hatrpo.py:92-192, trpo_util.py:96-129).  Until round 5 the vector
// setup of the conjugate-gradient solve (r.r), 1/2 x^T F x, the step size, the expected improvement and the line search's accept
// test were torch / rocBLAS ops with `.item()` synchronisations between them (rocblas_dot + ~30 ATen elementwise launches and
// five host round trips per agent).  Now:
//   harl_trpo_begin        g = grad_sum / sum(active), x = 0, r = p = g, cg_state = {r.r, 0}, st[LOSS] = surrogate at theta_old
//   harl_trpo_step         shs = 1/2 x.Fx, step = 1 / sqrt(shs / delta), full_step = step x, params_save = theta_old,
//                          st[EXPECTED] = g . full_step, st[FRACTION] = 1
//   harl_trpo_ls_candidate theta = params_save + fraction full_step
//   harl_trpo_ls_test      kl, improvement, the accept test (hatrpo.py:171-178) and the backtrack bookkeeping -- the host reads the
//                          16-double record ONCE per line-search step and nothing else
// The dot products accumulate in fp64 over fixed-order per-workgroup partials (same bits every run), like harl_trpo_cg_step,
// whose scratch block and grid-barrier scheme they share.  `st` (double[HARL_TRPO_STATE]): see include/harl_hip.h.
// =============================================================================================
namespace {
enum { ST_LOSS = 0, ST_SHS, ST_STEP, ST_EXPECTED, ST_FRACTION, ST_FLAG, ST_BACKTRACKS, ST_KL, ST_IMPROVE, ST_NEW_LOSS, ST_ENTROPY,
       ST_RATIO, ST_EXPECTED0 };

struct GridRed {  // CG_WGS co-resident workgroups of CG_THREADS: block / grid sums over CgScratch, last-workgroup-out reset
  CgScratch *ws;
  double *sh;
  int tid, blk;
  __device__ double block_sum(double v) {
    v = wave_reduce_sum_d(v);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < CG_THREADS / 64; ++w) t += sh[w];
    return t;
  }
  __device__ void put(double v, int which) {
    const double t = block_sum(v);
    if (tid == 0) ws->part[which][blk] = t;
  }
  __device__ double get(int which) {
    double g = 0;
    for (int b2 = 0; b2 < CG_WGS; ++b2) g += ws->part[which][b2];  // same order in every workgroup
    return g;
  }
  template <typename F>
  __device__ void finish(F &&last) {  // the last workgroup out publishes and leaves the barrier words at zero
    __syncthreads();
    if (tid == 0) {
      const unsigned done = __hip_atomic_fetch_add(ws->bar + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (done == (unsigned)CG_WGS - 1) {
        last();
        __hip_atomic_store(ws->bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws->bar + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
};
}  // namespace

__global__ __launch_bounds__(CG_THREADS) void k_trpo_begin(const float *__restrict__ grad_sum, const double *__restrict__ scalars,
                                                           long ls_off, int act_dim, float *__restrict__ g, float *__restrict__ x, float *__restrict__ r,
                                                           float *__restrict__ p, long n, float *__restrict__ cg_state,
                                                           double *__restrict__ st, CgScratch *__restrict__ ws) {
#pragma clang fp contract(off)
  __shared__ double sh[CG_THREADS / 64];
  GridRed R{ws, sh, (int)threadIdx.x, (int)blockIdx.x};
  const float inv = (float)(1.0 / scalars[1]);  // grad * (1 / sum(active)).to(float32)
  const long gtid = (long)blockIdx.x * CG_THREADS + threadIdx.x, gnt = (long)CG_WGS * CG_THREADS;
  double rr = 0.0;
  for (long i = gtid; i < n; i += gnt) {
    // Gaussian policies: d loss / d log_std travels among the loss kernel's scalar sums (entries 8 .. 8 + act_dim)
    const float raw = (ls_off >= 0 && i >= ls_off && i < ls_off + act_dim) ? (float)scalars[8 + (i - ls_off)] : grad_sum[i];
    const float gi = raw * inv;
    g[i] = gi;
    r[i] = gi;
    p[i] = gi;
    x[i] = 0.f;
    rr += (double)gi * (double)gi;
  }
  R.put(rr, 0);
  grid_barrier(ws->bar, (unsigned)CG_WGS);
  const double rdotr = R.get(0);
  R.finish([&] {
    cg_state[0] = (float)rdotr;
    cg_state[1] = 0.f;
    for (int k = 0; k < HARL_TRPO_STATE; ++k) st[k] = 0.0;
    st[ST_LOSS] = scalars[0] / scalars[1];
    st[ST_ENTROPY] = scalars[2] / scalars[1];
    st[ST_RATIO] = scalars[3] / scalars[4];
  });
}

__global__ __launch_bounds__(CG_THREADS) void k_trpo_step(const float *__restrict__ x, const float *__restrict__ fx,
                                                          const float *__restrict__ g, const float *__restrict__ theta,
                                                          float *__restrict__ theta_save, float *__restrict__ full_step, long n,
                                                          float kl_threshold, double *__restrict__ st,
                                                          CgScratch *__restrict__ ws) {
#pragma clang fp contract(off)
  __shared__ double sh[CG_THREADS / 64];
  GridRed R{ws, sh, (int)threadIdx.x, (int)blockIdx.x};
  const long gtid = (long)blockIdx.x * CG_THREADS + threadIdx.x, gnt = (long)CG_WGS * CG_THREADS;
  double d = 0.0;
  for (long i = gtid; i < n; i += gnt) d += (double)x[i] * (double)fx[i];
  R.put(d, 0);
  grid_barrier(ws->bar, (unsigned)CG_WGS);
  const float shs = 0.5f * (float)R.get(0);           // 0.5 * (x * F x).sum()            (hatrpo.py:123)
  const float step = 1.0f / sqrtf(shs / kl_threshold);  // 1 / torch.sqrt(shs / kl_threshold) (hatrpo.py:124)
  double e = 0.0;
  for (long i = gtid; i < n; i += gnt) {
    const float fs = step * x[i];
    full_step[i] = fs;
    theta_save[i] = theta[i];
    e += (double)g[i] * (double)fs;
  }
  R.put(e, 1);
  grid_barrier(ws->bar, (unsigned)(2 * CG_WGS));
  const double expected = (double)(float)R.get(1);  // (loss_grad * full_step).sum(): an fp32 figure in the reference
  R.finish([&] {
    st[ST_SHS] = (double)shs;
    st[ST_STEP] = (double)step;
    st[ST_EXPECTED] = expected;
    st[ST_EXPECTED0] = expected;
    st[ST_FRACTION] = 1.0;
    st[ST_FLAG] = 0.0;
    st[ST_BACKTRACKS] = 0.0;
  });
}

__global__ __launch_bounds__(256) void k_trpo_ls_candidate(const float *__restrict__ theta_save, const float *__restrict__ full_step,
                                                           const double *__restrict__ st, float *__restrict__ theta, long n) {
#pragma clang fp contract(off)
  const float fraction = (float)st[ST_FRACTION];  // python float x fp32 tensor: the scalar is rounded to fp32 first
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float t = fraction * full_step[i];
    theta[i] = theta_save[i] + t;
  }
}

__global__ void k_trpo_ls_test(const double *__restrict__ scalars, double *__restrict__ kl_sum, double m_global, double kl_threshold,
                               double accept_ratio, double backtrack_coeff, double *__restrict__ st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double new_loss = scalars[0] / scalars[1];
  const double kl = kl_sum[0] / m_global;
  kl_sum[0] = 0.0;  // (harl_trpo_kl_sum accumulates: left clean for the next candidate)
  const double improve = new_loss - st[ST_LOSS];
  st[ST_NEW_LOSS] = new_loss;
  st[ST_KL] = kl;
  st[ST_IMPROVE] = improve;
  st[ST_ENTROPY] = scalars[2] / scalars[1];
  st[ST_RATIO] = scalars[3] / scalars[4];
  if (kl < kl_threshold && (improve / st[ST_EXPECTED]) > accept_ratio && improve > 0.0) {  // hatrpo.py:171-178
    st[ST_FLAG] = 1.0;
  } else {
    st[ST_EXPECTED] = st[ST_EXPECTED] * backtrack_coeff;
    st[ST_FRACTION] = st[ST_FRACTION] * backtrack_coeff;
    st[ST_BACKTRACKS] = st[ST_BACKTRACKS] + 1.0;
  }
}

static CgScratch *cg_ws(void *scratch, const char *who) {
  CgScratch *ws = static_cast<CgScratch *>(scratch);
  if (!ws || (reinterpret_cast<uintptr_t>(ws) & 7) != 0) {
    set_error(who);
    return nullptr;
  }
  return ws;
}

extern "C" int harl_trpo_begin(const float *grad_sum, const double *scalars, long logstd_off, int act_dim, float *g, float *x,
                               float *r, float *p, long n, float *cg_state, double *st, void *scratch, void *stream) {
  if (n <= 0) return 0;
  CgScratch *ws = cg_ws(scratch, "harl_trpo_begin: scratch must be HARL_CG_SCRATCH_BYTES of zero-filled, 8-byte aligned device memory");
  if (!ws) return -2;
  hipLaunchKernelGGL(k_trpo_begin, dim3(CG_WGS), dim3(CG_THREADS), 0, (hipStream_t)stream, grad_sum, scalars, logstd_off, act_dim,
                     g, x, r, p, n, cg_state, st, ws);
  return check_launch("harl_trpo_begin");
}

extern "C" int harl_trpo_step(const float *x, const float *fx, const float *g, const float *theta, float *theta_save,
                              float *full_step, long n, float kl_threshold, double *st, void *scratch, void *stream) {
  if (n <= 0) return 0;
  CgScratch *ws = cg_ws(scratch, "harl_trpo_step: scratch must be HARL_CG_SCRATCH_BYTES of zero-filled, 8-byte aligned device memory");
  if (!ws) return -2;
  hipLaunchKernelGGL(k_trpo_step, dim3(CG_WGS), dim3(CG_THREADS), 0, (hipStream_t)stream, x, fx, g, theta, theta_save, full_step, n,
                     kl_threshold, st, ws);
  return check_launch("harl_trpo_step");
}

extern "C" int harl_trpo_ls_candidate(const float *theta_save, const float *full_step, const double *st, float *theta, long n,
                                      void *stream) {
  if (n <= 0) return 0;
  long nb = (n + 255) / 256;
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(k_trpo_ls_candidate, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, theta_save, full_step, st, theta, n);
  return check_launch("harl_trpo_ls_candidate");
}

extern "C" int harl_trpo_ls_test(const double *scalars, double *kl_sum, double m_global, double kl_threshold, double accept_ratio,
                                 double backtrack_coeff, double *st, void *stream) {
  hipLaunchKernelGGL(k_trpo_ls_test, dim3(1), dim3(64), 0, (hipStream_t)stream, scalars, kl_sum, m_global, kl_threshold,
                     accept_ratio, backtrack_coeff, st);
  return check_launch("harl_trpo_ls_test");
}

// out[0..PS_STRIDE) = column sums of the loss kernels' per-block partial rows (harl_reduce_scalars ACCUMULATES into out and
// needs a zeroing launch in front of it; this one overwrites)
__global__ __launch_bounds__(1024) void k_reduce_scalars_set(const float *__restrict__ ps, int n_blocks, double *__restrict__ out) {
  __shared__ double sh[16][64];
  const int j = threadIdx.x & 63, rg = threadIdx.x >> 6;
  double s = 0;
  if (j < PS_STRIDE)
    for (int b = rg; b < n_blocks; b += 16) s += (double)ps[(long)b * PS_STRIDE + j];
  sh[rg][j] = s;
  __syncthreads();
  if (threadIdx.x < PS_STRIDE) {
    double t = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += sh[g][threadIdx.x];
    out[threadIdx.x] = t;
  }
}

extern "C" int harl_reduce_scalars_set(const float *part_scalars, int n_blocks, double *scalars, void *stream) {
  hipLaunchKernelGGL(k_reduce_scalars_set, dim3(1), dim3(1024), 0, (hipStream_t)stream, part_scalars, n_blocks, scalars);
  return check_launch("harl_reduce_scalars_set");
}

// hipMemsetAsync behind the C ABI (the Python side's `tensor.zero_()` is an ATen fill kernel)
extern "C" int harl_zero_bytes(void *p, long bytes, void *stream) {
  if (!p || bytes <= 0) return 0;
  if (hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream) != hipSuccess) {
    set_error("harl_zero_bytes: hipMemsetAsync failed");
    return -2;
  }
  return 0;
}
