// multihead.hip -- MultiDiscrete action heads (harl/models/base/act.py:35-43,56-73,117-141: one Categorical per entry of
// `nvec` on the same trunk output; the reference's LAG environments use MultiDiscrete([41, 41, 41, 30])), gfx950.
//
// The concatenated heads are ordinary Linears: their logits come from the split-bf16 layer GEMM without an epilogue
// (harl_mlp_linear = k_fwd_hidden MODE 1, mlp.hip), the gradient into the trunk is harl_mlp_bwd_dx and the weight gradient
// harl_mlp_dw_partials over the d(loss)/d(logits) image written here -- the same three verified GEMM kernels as a hidden
// layer.  Heads are packed into GROUPS of at most 128 logits (one ATL(64 | 128) image per group; LAG: 123 + 30).
// This file is the per-sample arithmetic in between (HBM-bound, one wave per 32-sample slab, lane (s, h) holds SP/2 logits
// of sample s; per-head reductions = in-lane over the head's registers + one lane^32 exchange):
//   log-softmax per head, log pi(a) = sum over heads (act.py:124-137), entropy, importance ratio against the buffer's
//   [rows, old_w] stored log-probs, clipped surrogate x factor, and d(unscaled loss)/d(logits).
// Reference quirks kept: the summed log-prob [m, 1] is compared with EVERY column of the stored
// [m, n_heads] array, so `prod` raises the ratio to the n_heads-th power; the entropy bonus is (1/m) sum_rows sum_heads H,
// not active-mask weighted (act.py:126-139 broadcasts [m] x [m, 1]).
#include "common.h"
#include "../../include/harl_hip.h"

using namespace harl;

namespace {
int bad(const char *m) {
  set_error(m);
  return -2;
}

constexpr int MAXH = HARL_MD_MAX_HEADS, MAXG = HARL_MD_MAX_GROUPS;

struct MdArgs {
  const float *z[MAXG];  // logits images, ATL(sp[g])
  float *dz[MAXG];       // d(loss)/d(logits) images (may alias z)
  int sp[MAXG];
  int n_groups, n_heads;
  int head_group[MAXH], head_lo[MAXH], head_n[MAXH];  // head k = logits [lo, lo + n) of its group's image
  int head_out_off[MAXH];                             // column of head k in head_out [M, S]
  int S;
  long M, m_valid, m_pad;
  const int64_t *idx;
  const float *actions;   // [rows, n_heads] indices stored as fp32
  const float *old_logp;  // [rows, old_w]
  int old_w;
  const float *adv;
  const double *adv_moments;
  const float *factor_in, *active;
  const float *ent_scale;  // device scalar sum(active) / m (see harl_hip.h), NULL = 1
  float clip_lo, clip_hi, entropy_coef;
  int agg_mean, mode;
  float *logp_out, *factor_out, *head_out, *part_scalars;
  long n_slabs;
};

template <int SP>
__device__ __forceinline__ void group_load(const float *img, long slab, int lane, float (&v)[64]) {
  float t[SP / 2];
  atl_load<SP>(img, slab, lane, t);
#pragma unroll
  for (int R = 0; R < SP / 2; ++R) v[R] = t[R];
}
template <int SP>
__device__ __forceinline__ void group_store(float *img, long slab, int lane, const float (&v)[64]) {
  float t[SP / 2];
#pragma unroll
  for (int R = 0; R < SP / 2; ++R) t[R] = v[R];
  atl_store<SP>(img, slab, lane, t);
}

// statistics of ONE head (logits [lo, hi) of the image held by this lane pair): lse, entropy, log p(action)
template <int SP>
__device__ __forceinline__ void head_stats(const float (&v)[64], int h, int lo, int hi, int a, float &lse, float &ent,
                                           float &lpa) {
  float mx = -3.0e38f;
#pragma unroll
  for (int R = 0; R < SP / 2; ++R) {
    const int f = feat_base(R) + 4 * h;
    if (f >= lo && f < hi) mx = fmaxf(mx, v[R]);
  }
  mx = fmaxf(mx, wave_xor32(mx));
  float se = 0.f;
#pragma unroll
  for (int R = 0; R < SP / 2; ++R) {
    const int f = feat_base(R) + 4 * h;
    if (f >= lo && f < hi) se += expf(v[R] - mx);
  }
  se = wave_sum32(se);
  const float l = mx + logf(se);
  float e = 0.f, la = 0.f;
#pragma unroll
  for (int R = 0; R < SP / 2; ++R) {
    const int f = feat_base(R) + 4 * h;
    if (f >= lo && f < hi) {
      const float lp = v[R] - l;
      e -= fmaxf(lp, -3.4028234663852886e38f) * expf(lp);  // torch Categorical.entropy (distributions.py:7-25)
      if (f == lo + a) la = lp;
    }
  }
  lse = l;
  ent = wave_sum32(e);
  lpa = wave_sum32(la);
}

// second pass over one head: d(loss)/d(logit) into out (TRAIN) or the normalised logits to head_out
template <int SP, bool TRAIN>
__device__ __forceinline__ void head_emit(const float (&v)[64], float (&out)[64], int h, int lo, int hi, int a, float lse,
                                          float ent, float dlp, float ecoef, bool valid, float *ho_row) {
#pragma unroll
  for (int R = 0; R < SP / 2; ++R) {
    const int f = feat_base(R) + 4 * h;
    if (f >= lo && f < hi) {
      const float lp = v[R] - lse;
      if constexpr (TRAIN) {
        const float p = expf(lp);
        const float onehot = f == lo + a ? 1.f : 0.f;
        // d logp_a / dz_c = onehot - p_c ;  d ent / dz_c = -p_c (log p_c + ent)
        out[R] = valid ? dlp * (onehot - p) + ecoef * (-p * (lp + ent)) : 0.f;
      } else if (valid) {
        ho_row[f - lo] = lp;
      }
    }
  }
}

template <bool TRAIN>
__global__ __launch_bounds__(WG_THREADS) void k_md_head(MdArgs A) {
  __shared__ float red[4 * PS_STRIDE];
  __shared__ float st[WAVES_PER_WG][3][MAXH][WAVE];  // per wave, per head, per lane: lse, entropy, log p(a)
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31, h = lane >> 5;
  float adv_mean = 0.f, adv_den = 1.f;
  if (TRAIN && A.adv_moments) {  // happo.py:122-127
    const double cnt = A.adv_moments[2];
    const double m = A.adv_moments[0] / cnt;
    const double var = A.adv_moments[1] / cnt - m * m;
    adv_mean = (float)m;
    adv_den = 1.0f / ((float)sqrt(var > 0 ? var : 0.0) + 1e-5f);
  }
  const float ent_scale = (TRAIN && A.ent_scale) ? A.ent_scale[0] : 1.f;
  float sc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sc[k] = 0.f;

  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < A.n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    const long j = slab * SLAB + i;
    const bool valid = j < A.M && (A.m_pad == 0 || (j % A.m_pad) < A.m_valid);
    const long jc = j < A.M ? j : A.M - 1;
    const long row = A.idx ? A.idx[jc] : jc;
    const long orow = TRAIN ? row : jc;  // log-prob passes address old_logp / factor by batch position
    const float *arow = A.actions ? A.actions + row * A.n_heads : nullptr;

    float v[64];
    for (int g = 0; g < A.n_groups; ++g) {
      const int sp = A.sp[g];
      if (sp == 128) group_load<128>(A.z[g], slab, lane, v);
      else group_load<64>(A.z[g], slab, lane, v);
#pragma unroll 1
      for (int k = 0; k < A.n_heads; ++k) {
        if (A.head_group[k] != g) continue;  // wave-uniform
        const int lo = A.head_lo[k], hi = lo + A.head_n[k], a = arow ? (int)arow[k] : 0;
        float l, e, la;
        if (sp == 128) head_stats<128>(v, h, lo, hi, a, l, e, la);
        else head_stats<64>(v, h, lo, hi, a, l, e, la);
        st[wave][0][k][lane] = l;
        st[wave][1][k][lane] = e;
        st[wave][2][k][lane] = la;
      }
    }
    float LP = 0.f, ENT = 0.f;  // act.py:134-139: cat(...).sum(dim=-1) over the heads, in order
#pragma unroll 1
    for (int k = 0; k < A.n_heads; ++k) {
      ENT += st[wave][1][k][lane];
      LP += st[wave][2][k][lane];
    }
    float imp = 1.f, dimp_dlp = 1.f;
    if (TRAIN || A.old_logp) {  // happo.py:66-70: getattr(torch, aggregation)(exp(logp - old), dim=-1)
      float prod = 1.f, sum = 0.f;
#pragma unroll 1
      for (int k = 0; k < A.old_w; ++k) {
        const float r = expf(LP - A.old_logp[orow * A.old_w + k]);
        prod *= r;
        sum += r;
      }
      imp = A.agg_mean ? sum * (1.0f / (float)A.old_w) : prod;
      dimp_dlp = A.agg_mean ? imp : imp * (float)A.old_w;  // every column depends on the same summed log-prob
    }

    float dlp = 0.f, ecoef = 0.f;
    if constexpr (!TRAIN) {
      if (valid && h == 0) {
        if (A.logp_out) A.logp_out[j] = LP;
        if (A.factor_out) A.factor_out[j] = A.factor_out[j] * imp;  // on_policy_ha_runner.py:116-124
      }
      if (!A.head_out) continue;
    } else {
      if (A.logp_out && valid && h == 0) A.logp_out[j] = LP;
      const float actv = A.active ? A.active[row] : 1.f;
      const float advn = (A.adv[row] - adv_mean) * adv_den;
      const float fct = A.factor_in ? A.factor_in[row] : 1.f;
      const float surr1 = imp * advn;
      const float impc = fminf(fmaxf(imp, A.clip_lo), A.clip_hi);
      const float surr2 = impc * advn;
      const float mn = fminf(surr1, surr2);
      const float inrange = (imp >= A.clip_lo && imp <= A.clip_hi) ? 1.f : 0.f;
      float gsel = surr1 < surr2 ? 1.f : (surr1 > surr2 ? inrange : 0.5f + 0.5f * inrange);  // torch.min ties split evenly
      if (A.mode != 0) gsel = 1.f;  // HAA2C: no clip (haa2c.py:70-80)
      dlp = valid ? -fct * actv * advn * gsel * dimp_dlp : 0.f;
      ecoef = valid ? -A.entropy_coef * ent_scale : 0.f;
      if (valid && h == 0) {
        sc[0] += -fct * (A.mode == 2 ? surr1 : mn) * actv;
        sc[1] += actv;
        sc[2] += ENT * ent_scale;
        sc[3] += imp;
        sc[4] += 1.f;
      }
    }

    // second pass over the logits: d(loss)/d(logit) (TRAIN) or the normalised logits themselves (head_out)
    for (int g = 0; g < A.n_groups; ++g) {
      const int sp = A.sp[g];
      if (sp == 128) group_load<128>(A.z[g], slab, lane, v);
      else group_load<64>(A.z[g], slab, lane, v);
      float out[64];
#pragma unroll
      for (int R = 0; R < 64; ++R) out[R] = 0.f;
#pragma unroll 1
      for (int k = 0; k < A.n_heads; ++k) {
        if (A.head_group[k] != g) continue;
        const int lo = A.head_lo[k], hi = lo + A.head_n[k], a = arow ? (int)arow[k] : 0;
        const float l = st[wave][0][k][lane], e = st[wave][1][k][lane];
        float *ho = (!TRAIN && valid) ? A.head_out + j * A.S + A.head_out_off[k] : nullptr;
        if (sp == 128) head_emit<128, TRAIN>(v, out, h, lo, hi, a, l, e, dlp, ecoef, valid, ho);
        else head_emit<64, TRAIN>(v, out, h, lo, hi, a, l, e, dlp, ecoef, valid, ho);
      }
      if constexpr (TRAIN) {
        if (sp == 128) group_store<128>(A.dz[g], slab, lane, out);
        else group_store<64>(A.dz[g], slab, lane, out);
      }
    }
  }

  if constexpr (TRAIN) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float t = wave_reduce_sum(sc[k]);
      if (lane == 0) red[wave * PS_STRIDE + k] = t;
    }
    __syncthreads();
    if (threadIdx.x < PS_STRIDE) {
      float t = 0.f;
      if (threadIdx.x < 8)
        t = (red[0 * PS_STRIDE + threadIdx.x] + red[1 * PS_STRIDE + threadIdx.x]) +
            (red[2 * PS_STRIDE + threadIdx.x] + red[3 * PS_STRIDE + threadIdx.x]);
      A.part_scalars[(long)blockIdx.x * PS_STRIDE + threadIdx.x] = t;
    }
  }
}

int fill_layout(MdArgs &A, int n_groups, const int *sp, int n_heads, const int *nvec, const int *head_group, const char *who) {
  if (n_groups < 1 || n_groups > MAXG || n_heads < 1 || n_heads > MAXH) {
    set_error(who);
    return -2;
  }
  A.n_groups = n_groups;
  A.n_heads = n_heads;
  int fill[MAXG] = {0, 0, 0, 0};
  int S = 0;
  for (int g = 0; g < n_groups; ++g) {
    if (sp[g] != 64 && sp[g] != 128) return bad("MultiDiscrete head: a group image is ATL(64) or ATL(128)");
    A.sp[g] = sp[g];
  }
  for (int k = 0; k < n_heads; ++k) {
    const int g = head_group[k];
    if (g < 0 || g >= n_groups || nvec[k] < 1 || fill[g] + nvec[k] > sp[g])
      return bad("MultiDiscrete head: heads do not fit their group images");
    A.head_group[k] = g;
    A.head_lo[k] = fill[g];
    A.head_n[k] = nvec[k];
    A.head_out_off[k] = S;
    fill[g] += nvec[k];
    S += nvec[k];
  }
  A.S = S;
  return 0;
}
}  // namespace

extern "C" int harl_md_head_logp(const float *const *z, int n_groups, const int *sp, int n_heads, const int *nvec,
                                 const int *head_group, long M, const float *actions, float *logp_out,
                                 const float *old_logp, int old_w, float *factor, int agg_mean, float *head_out,
                                 long m_valid, long m_pad, void *stream) {
  if (M <= 0) return 0;
  MdArgs A{};
  if (int rc = fill_layout(A, n_groups, sp, n_heads, nvec, head_group, "harl_md_head_logp: at most 4 groups / 8 heads")) return rc;
  if (old_logp && (old_w < 1 || old_w > MAXH)) return bad("harl_md_head_logp: old_w must be 1..8");
  if (factor && !old_logp) return bad("harl_md_head_logp: the factor product needs old_logp");
  for (int g = 0; g < n_groups; ++g) A.z[g] = z[g];
  A.M = M; A.m_valid = m_valid; A.m_pad = m_pad;
  A.actions = actions; A.logp_out = logp_out; A.old_logp = old_logp; A.old_w = old_logp ? old_w : 0;
  A.factor_out = factor; A.agg_mean = agg_mean; A.head_out = head_out;
  A.n_slabs = n_slabs_of(M);
  const int grid = persistent_grid(A.n_slabs, 4);
  hipLaunchKernelGGL(k_md_head<false>, dim3(grid), dim3(WG_THREADS), 0, (hipStream_t)stream, A);
  return check_launch("harl_md_head_logp");
}

extern "C" int harl_md_head_loss(const float *const *z, float *const *dz, int n_groups, const int *sp, int n_heads,
                                 const int *nvec, const int *head_group, long M, const int64_t *idx, const float *actions,
                                 const float *old_logp, int old_w, const float *adv, const double *adv_moments,
                                 const float *factor, const float *active, const float *ent_scale, double clip_param,
                                 float entropy_coef, int agg_mean, int mode, long m_valid, long m_pad, float *logp_out,
                                 float *part_scalars, int n_blocks, void *stream) {
  if (M <= 0) return 0;
  MdArgs A{};
  if (int rc = fill_layout(A, n_groups, sp, n_heads, nvec, head_group, "harl_md_head_loss: at most 4 groups / 8 heads")) return rc;
  if (old_w < 1 || old_w > MAXH) return bad("harl_md_head_loss: old_w must be 1..8");
  if (mode != 0 && mode != 2) return bad("harl_md_head_loss: mode 0 (HAPPO / MAPPO) or 2 (HAA2C); the reference's HATRPO rejects MultiDiscrete");
  if (n_blocks < 1) return bad("harl_md_head_loss: n_blocks must be positive");
  for (int g = 0; g < n_groups; ++g) {
    A.z[g] = z[g];
    A.dz[g] = dz[g];
  }
  A.M = M; A.m_valid = m_valid; A.m_pad = m_pad; A.idx = idx;
  A.actions = actions; A.old_logp = old_logp; A.old_w = old_w; A.adv = adv; A.adv_moments = adv_moments;
  A.factor_in = factor; A.active = active; A.ent_scale = ent_scale;
  A.clip_lo = (float)(1.0 - clip_param); A.clip_hi = (float)(1.0 + clip_param);
  A.entropy_coef = entropy_coef; A.agg_mean = agg_mean; A.mode = mode;
  A.logp_out = logp_out; A.part_scalars = part_scalars;
  A.n_slabs = n_slabs_of(M);
  hipLaunchKernelGGL(k_md_head<true>, dim3(n_blocks), dim3(WG_THREADS), 0, (hipStream_t)stream, A);
  return check_launch("harl_md_head_loss");
}
