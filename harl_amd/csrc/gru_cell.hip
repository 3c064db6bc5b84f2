// gru_cell.hip -- the element-wise half of a GRU step for hidden widths whose recurrent weights do not fit the LDS of the
// fused recurrence kernels (csrc/gru.hip keeps the three bf16 images of W_hh [3H][H] resident: 122 KiB at H = 64, 3 x that at
// H = 128).  For H = 128 (harl/models/base/rnn.py:8-81 on the default `hidden_sizes: [128, 128]`) a step is composed on the host
// (harl_amd/gru_wide.py) from the verified layer GEMM (harl_mlp_linear: gh_g = W_hg h~ + b_hg per gate, ATL images) and the
// kernels below; the input halves gi_g = W_ig' x_hat + b_ig of ALL steps are three harl_mlp_linear launches up front.
//   cell forward :  r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h = (1 - z) n + z h~
//                   (torch.nn.GRU, gate order r, z, n); emits h, the next step's h~ = h * mask_{l+1}, and the saved gates
//   cell backward:  BPTT of one step given G_l = d(loss)/d(h_l) pieces; emits dgi = [dr, dz, dn], dgh_n and G_l * z_l
//   row norm     :  y = (h - mean) * rstd  over all steps (rnn.norm without its affine part, which is folded into the head)
// All tensors are ATL(H) images (common.h): one wave per 32-sequence slab, a lane pair holds a sequence's H features.
// Coverage path: ~4 launches per time step instead of one persistent kernel per chunk.
#include "common.h"
#include "../../include/harl_hip.h"

using namespace harl;

namespace {
int bad(const char *m) {
  set_error(m);
  return -2;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// hpm0 = h0 * mask_0 : row-major [m_pad][H] -> ATL(H) image of step 0
template <int H>
__global__ __launch_bounds__(WG_THREADS) void k_gru_init(const float *__restrict__ h0, const float *__restrict__ mask_rows,
                                                         float *__restrict__ hpm0, long n_slabs) {
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31, h = lane >> 5;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    const long row = slab * SLAB + i;
    const float mk = mask_rows[row];
    float v[H / 2];
#pragma unroll
    for (int R = 0; R < H / 2; ++R) v[R] = h0[row * H + feat_base(R) + 4 * h] * mk;
    atl_store<H>(hpm0, slab, lane, v);
  }
}

template <int H, bool SAVE>
__global__ __launch_bounds__(WG_THREADS) void k_gru_cell_fwd(
    const float *__restrict__ gi_r, const float *__restrict__ gi_z, const float *__restrict__ gi_n,
    const float *__restrict__ gh_r, const float *__restrict__ gh_z, const float *__restrict__ gh_n,
    const float *__restrict__ hpm, const float *__restrict__ mask_next, float *__restrict__ r_out, float *__restrict__ z_out,
    float *__restrict__ n_out, float *__restrict__ hn_out, float *__restrict__ h_out, float *__restrict__ hpm_next,
    float *__restrict__ h_last, long n_slabs) {
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31, h = lane >> 5;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float a[H / 2], b[H / 2], rr[H / 2], zz[H / 2], nn[H / 2], hp[H / 2], hn[H / 2];
    atl_load<H>(gi_r, slab, lane, a);
    atl_load<H>(gh_r, slab, lane, b);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) rr[R] = sigmoidf_(a[R] + b[R]);
    atl_load<H>(gi_z, slab, lane, a);
    atl_load<H>(gh_z, slab, lane, b);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) zz[R] = sigmoidf_(a[R] + b[R]);
    atl_load<H>(gi_n, slab, lane, a);
    atl_load<H>(gh_n, slab, lane, hn);
    atl_load<H>(hpm, slab, lane, hp);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) {
      nn[R] = tanhf(a[R] + rr[R] * hn[R]);
      a[R] = (1.0f - zz[R]) * nn[R] + zz[R] * hp[R];  // h_l
    }
    atl_store<H>(h_out, slab, lane, a);
    if constexpr (SAVE) {
      atl_store<H>(r_out, slab, lane, rr);
      atl_store<H>(z_out, slab, lane, zz);
      atl_store<H>(n_out, slab, lane, nn);
      atl_store<H>(hn_out, slab, lane, hn);
    }
    if (hpm_next) {  // h~ of the next step: the reset mask multiplies the carried state (rnn.py:27-32,61-70)
      const float mk = mask_next[slab * SLAB + i];
#pragma unroll
      for (int R = 0; R < H / 2; ++R) b[R] = a[R] * mk;
      atl_store<H>(hpm_next, slab, lane, b);
    }
    if (h_last) {
      const long row = slab * SLAB + i;
#pragma unroll
      for (int R = 0; R < H / 2; ++R) h_last[row * H + feat_base(R) + 4 * h] = a[R];
    }
  }
}

// one BPTT step.  G_l = dh_out_l + mask_{l+1} * (gz_next + t_r + t_z + t_n)   (second term absent at the last step), where
// gz_next = G_{l+1} z_{l+1} and t_g = W_hg^T dgh_{g, l+1} come from the step after.  In place: gz (in: gz_next, out: G_l z_l).
template <int H>
__global__ __launch_bounds__(WG_THREADS) void k_gru_cell_bwd(
    const float *__restrict__ dh_out, const float *__restrict__ t_r, const float *__restrict__ t_z,
    const float *__restrict__ t_n, const float *__restrict__ mask_next, const float *__restrict__ r_s,
    const float *__restrict__ z_s, const float *__restrict__ n_s, const float *__restrict__ hn_s,
    const float *__restrict__ hpm, float *__restrict__ gz, float *__restrict__ dr, float *__restrict__ dz,
    float *__restrict__ dn, float *__restrict__ dhn, int has_next, long n_slabs) {
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float G[H / 2], a[H / 2], b[H / 2];
    atl_load<H>(dh_out, slab, lane, G);
    if (has_next) {
      const float mk = mask_next[slab * SLAB + i];
      atl_load<H>(gz, slab, lane, a);
      atl_load<H>(t_r, slab, lane, b);
#pragma unroll
      for (int R = 0; R < H / 2; ++R) a[R] += b[R];
      atl_load<H>(t_z, slab, lane, b);
#pragma unroll
      for (int R = 0; R < H / 2; ++R) a[R] += b[R];
      atl_load<H>(t_n, slab, lane, b);
#pragma unroll
      for (int R = 0; R < H / 2; ++R) G[R] += mk * (a[R] + b[R]);
    }
    float rr[H / 2], zz[H / 2], nn[H / 2];
    atl_load<H>(z_s, slab, lane, zz);
    atl_load<H>(n_s, slab, lane, nn);
    atl_load<H>(hpm, slab, lane, a);   // h~_l
#pragma unroll
    for (int R = 0; R < H / 2; ++R) {
      b[R] = G[R] * (a[R] - nn[R]) * zz[R] * (1.0f - zz[R]);   // d(gi_z + gh_z)
      a[R] = G[R] * (1.0f - zz[R]) * (1.0f - nn[R] * nn[R]);   // d(gi_n + r gh_n)
      G[R] = G[R] * zz[R];                                     // the direct path into h~_l
    }
    atl_store<H>(dz, slab, lane, b);
    atl_store<H>(dn, slab, lane, a);
    atl_store<H>(gz, slab, lane, G);
    atl_load<H>(r_s, slab, lane, rr);
    atl_load<H>(hn_s, slab, lane, b);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) {
      b[R] = a[R] * b[R] * rr[R] * (1.0f - rr[R]);  // d(gi_r + gh_r)
      a[R] = a[R] * rr[R];                          // d gh_n
    }
    atl_store<H>(dr, slab, lane, b);
    atl_store<H>(dhn, slab, lane, a);
  }
}

// y = (x - mean) * rstd per row, rstd = 1 / sqrt(var + 1e-5) (biased variance: nn.LayerNorm)
template <int H>
__global__ __launch_bounds__(WG_THREADS) void k_rownorm(const float *__restrict__ x, float *__restrict__ y,
                                                        float *__restrict__ rstd_out, long n_slabs) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float v[H / 2];
    atl_load<H>(x, slab, lane, v);
    float s = 0.f;
#pragma unroll
    for (int R = 0; R < H / 2; ++R) s += v[R];
    s = wave_sum32(s);
    const float mean = s * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int R = 0; R < H / 2; ++R) {
      v[R] -= mean;
      q += v[R] * v[R];
    }
    q = wave_sum32(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / H) + 1e-5f);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) v[R] *= rstd;
    atl_store<H>(y, slab, lane, v);
    if (lane < 32) rstd_out[slab * SLAB + lane] = rstd;
  }
}

// forward-mode tangent of one cell step (HATRPO's Fisher-vector product on the composed GRU path, harl_amd/gru_wide.py):
//   r_dot = r (1 - r) sum(g_r),  z_dot = z (1 - z) sum(g_z),  n_dot = (1 - n^2) (sum(gi_n) + r_dot hn + r sum(gh_n)),
//   h_dot = (1 - z) n_dot + z_dot (h~ - n) + z h~_dot,        h~_dot of the next step = h_dot * mask_next
// every gate tangent arrives as the images of the GEMMs that form it: gia = W_i x_dot, gib = W_i_dot x + b_i_dot (input side),
// gha = W_h h~_dot (NULL at the first step: h0 carries no tangent), ghb = W_h_dot h~ + b_h_dot.
struct GruTanArgs {
  const float *gia[3], *gib[3], *gha[3], *ghb[3];
  const float *r, *z, *n, *hn, *hpm, *hpm_dot, *mask_next;
  float *h_dot, *hpm_dot_next;
  long n_slabs;
};

template <int H>
__global__ __launch_bounds__(WG_THREADS) void k_gru_cell_tangent(GruTanArgs A) {
  const int lane = threadIdx.x & 63, wave = wave_id(), i = lane & 31;
  for (long slab = (long)blockIdx.x * WAVES_PER_WG + wave; slab < A.n_slabs; slab += (long)gridDim.x * WAVES_PER_WG) {
    float gi[H / 2], gh[H / 2], t[H / 2], rr[H / 2], rd[H / 2], zd[H / 2];
    auto sum_in = [&](int g) {
      atl_load<H>(A.gia[g], slab, lane, gi);
      atl_load<H>(A.gib[g], slab, lane, t);
#pragma unroll
      for (int R = 0; R < H / 2; ++R) gi[R] += t[R];
    };
    auto sum_hid = [&](int g) {
      atl_load<H>(A.ghb[g], slab, lane, gh);
      if (A.gha[g]) {
        atl_load<H>(A.gha[g], slab, lane, t);
#pragma unroll
        for (int R = 0; R < H / 2; ++R) gh[R] += t[R];
      }
    };
    sum_in(0);
    sum_hid(0);
    atl_load<H>(A.r, slab, lane, rr);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) rd[R] = rr[R] * (1.0f - rr[R]) * (gi[R] + gh[R]);
    sum_in(1);
    sum_hid(1);
    float zz[H / 2];
    atl_load<H>(A.z, slab, lane, zz);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) zd[R] = zz[R] * (1.0f - zz[R]) * (gi[R] + gh[R]);
    sum_in(2);
    sum_hid(2);
    float nn[H / 2];
    atl_load<H>(A.n, slab, lane, nn);
    atl_load<H>(A.hn, slab, lane, t);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) gi[R] = (1.0f - nn[R] * nn[R]) * (gi[R] + rd[R] * t[R] + rr[R] * gh[R]);  // n_dot
    atl_load<H>(A.hpm, slab, lane, t);
#pragma unroll
    for (int R = 0; R < H / 2; ++R) gi[R] = (1.0f - zz[R]) * gi[R] + zd[R] * (t[R] - nn[R]);
    if (A.hpm_dot) {
      atl_load<H>(A.hpm_dot, slab, lane, t);
#pragma unroll
      for (int R = 0; R < H / 2; ++R) gi[R] += zz[R] * t[R];
    }
    atl_store<H>(A.h_dot, slab, lane, gi);
    if (A.hpm_dot_next) {
      const float mk = A.mask_next[slab * SLAB + i];
#pragma unroll
      for (int R = 0; R < H / 2; ++R) gi[R] *= mk;
      atl_store<H>(A.hpm_dot_next, slab, lane, gi);
    }
  }
}

int grid_of(long n_slabs) { return persistent_grid(n_slabs, 4); }
}  // namespace

extern "C" int harl_gru_cell_tangent(const float *gia_r, const float *gia_z, const float *gia_n, const float *gib_r,
                                     const float *gib_z, const float *gib_n, const float *gha_r, const float *gha_z,
                                     const float *gha_n, const float *ghb_r, const float *ghb_z, const float *ghb_n,
                                     const float *r, const float *z, const float *n, const float *hn, const float *hpm,
                                     const float *hpm_dot, const float *mask_next, int H, long m_pad, float *h_dot,
                                     float *hpm_dot_next, void *stream) {
  if (m_pad <= 0) return 0;
  if (m_pad % SLAB) return bad("harl_gru_cell_tangent: m_pad must be a multiple of 32");
  if (hpm_dot_next && !mask_next) return bad("harl_gru_cell_tangent: the next step's h~_dot needs its reset masks");
  if ((gha_r != nullptr) != (hpm_dot != nullptr) || (gha_r != nullptr) != (gha_z != nullptr) || (gha_r != nullptr) != (gha_n != nullptr))
    return bad("harl_gru_cell_tangent: gha_* and hpm_dot are given together (all NULL at the first step)");
  GruTanArgs A{};
  A.gia[0] = gia_r; A.gia[1] = gia_z; A.gia[2] = gia_n;
  A.gib[0] = gib_r; A.gib[1] = gib_z; A.gib[2] = gib_n;
  A.gha[0] = gha_r; A.gha[1] = gha_z; A.gha[2] = gha_n;
  A.ghb[0] = ghb_r; A.ghb[1] = ghb_z; A.ghb[2] = ghb_n;
  A.r = r; A.z = z; A.n = n; A.hn = hn; A.hpm = hpm; A.hpm_dot = hpm_dot; A.mask_next = mask_next;
  A.h_dot = h_dot; A.hpm_dot_next = hpm_dot_next;
  A.n_slabs = m_pad / SLAB;
  if (H == 128) hipLaunchKernelGGL(k_gru_cell_tangent<128>, dim3(grid_of(A.n_slabs)), dim3(WG_THREADS), 0, (hipStream_t)stream, A);
  else if (H == 64) hipLaunchKernelGGL(k_gru_cell_tangent<64>, dim3(grid_of(A.n_slabs)), dim3(WG_THREADS), 0, (hipStream_t)stream, A);
  else return bad("harl_gru_cell_tangent: H must be 64 or 128");
  return check_launch("harl_gru_cell_tangent");
}

extern "C" int harl_gru_cell_init(const float *h0, const float *mask_rows, int H, long m_pad, float *hpm0, void *stream) {
  if (m_pad <= 0) return 0;
  if (m_pad % SLAB) return bad("harl_gru_cell_init: m_pad must be a multiple of 32");
  const long ns = m_pad / SLAB;
  if (H == 128) hipLaunchKernelGGL(k_gru_init<128>, dim3(grid_of(ns)), dim3(WG_THREADS), 0, (hipStream_t)stream, h0, mask_rows, hpm0, ns);
  else if (H == 64) hipLaunchKernelGGL(k_gru_init<64>, dim3(grid_of(ns)), dim3(WG_THREADS), 0, (hipStream_t)stream, h0, mask_rows, hpm0, ns);
  else return bad("harl_gru_cell_init: H must be 64 or 128");
  return check_launch("harl_gru_cell_init");
}

extern "C" int harl_gru_cell_fwd(const float *gi_r, const float *gi_z, const float *gi_n, const float *gh_r, const float *gh_z,
                                 const float *gh_n, const float *hpm, const float *mask_next, int H, long m_pad, float *r,
                                 float *z, float *n, float *hn, float *h, float *hpm_next, float *h_last, void *stream) {
  if (m_pad <= 0) return 0;
  if (m_pad % SLAB) return bad("harl_gru_cell_fwd: m_pad must be a multiple of 32");
  if (hpm_next && !mask_next) return bad("harl_gru_cell_fwd: the next step's h~ needs its reset masks");
  const long ns = m_pad / SLAB;
  const bool save = r != nullptr;
  hipStream_t s = (hipStream_t)stream;
#define L(HH, SV)                                                                                                       \
  hipLaunchKernelGGL((k_gru_cell_fwd<HH, SV>), dim3(grid_of(ns)), dim3(WG_THREADS), 0, s, gi_r, gi_z, gi_n, gh_r, gh_z, gh_n, \
                     hpm, mask_next, r, z, n, hn, h, hpm_next, h_last, ns)
  if (H == 128) { if (save) L(128, true); else L(128, false); }
  else if (H == 64) { if (save) L(64, true); else L(64, false); }
  else return bad("harl_gru_cell_fwd: H must be 64 or 128");
#undef L
  return check_launch("harl_gru_cell_fwd");
}

extern "C" int harl_gru_cell_bwd(const float *dh_out, const float *t_r, const float *t_z, const float *t_n,
                                 const float *mask_next, const float *r, const float *z, const float *n, const float *hn,
                                 const float *hpm, int H, long m_pad, float *gz, float *dr, float *dz, float *dn, float *dhn,
                                 void *stream) {
  if (m_pad <= 0) return 0;
  if (m_pad % SLAB) return bad("harl_gru_cell_bwd: m_pad must be a multiple of 32");
  const long ns = m_pad / SLAB;
  const int has_next = t_r != nullptr;
  if (has_next && !(t_z && t_n && mask_next)) return bad("harl_gru_cell_bwd: the carried gradient needs t_r, t_z, t_n and the masks");
  hipStream_t s = (hipStream_t)stream;
  if (H == 128)
    hipLaunchKernelGGL(k_gru_cell_bwd<128>, dim3(grid_of(ns)), dim3(WG_THREADS), 0, s, dh_out, t_r, t_z, t_n, mask_next, r, z, n,
                       hn, hpm, gz, dr, dz, dn, dhn, has_next, ns);
  else if (H == 64)
    hipLaunchKernelGGL(k_gru_cell_bwd<64>, dim3(grid_of(ns)), dim3(WG_THREADS), 0, s, dh_out, t_r, t_z, t_n, mask_next, r, z, n,
                       hn, hpm, gz, dr, dz, dn, dhn, has_next, ns);
  else return bad("harl_gru_cell_bwd: H must be 64 or 128");
  return check_launch("harl_gru_cell_bwd");
}

extern "C" int harl_rownorm(const float *x, long M, int H, float *y, float *rstd, void *stream) {
  if (M <= 0) return 0;
  const long ns = n_slabs_of(M);
  if (H == 128) hipLaunchKernelGGL(k_rownorm<128>, dim3(grid_of(ns)), dim3(WG_THREADS), 0, (hipStream_t)stream, x, y, rstd, ns);
  else if (H == 64) hipLaunchKernelGGL(k_rownorm<64>, dim3(grid_of(ns)), dim3(WG_THREADS), 0, (hipStream_t)stream, x, y, rstd, ns);
  else return bad("harl_rownorm: H must be 64 or 128");
  return check_launch("harl_rownorm");
}
