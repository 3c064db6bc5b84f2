// comm.hip -- one-shot SUM all-reduce of the data-parallel update's small messages over peer-mapped buffers (gfx950, xGMI).
//
// every optimiser step of the sharded update exchanges ONE message of < 100 KiB ([folded gradients |
// four exact scalar pieces], harl_amd/dist.py) and every agent one of 24 bytes (the advantage moments): latency-bound.  A ring
// pays 2 (P - 1) hops of the 7-link point-to-point fabric; here every rank PUSHES its message straight into a slot of every
// peer's buffer (P - 1 concurrent writes, one per link), raises a flag behind it, waits for the P flags in its OWN buffer and
// sums the P local slots in rank order -- one hop, and the same bits on every rank (the replicated parameters must not drift).
//
//   buffer of rank r (device memory of r, mapped into every peer through hipIpc):
//     data  [2 sets][P slots][cap bytes]      slot q of set s = the message of rank q at an epoch of parity s
//     flags [2 sets][P ranks][NBLK blocks]    epoch of the chunk that has landed (written by the owner of the slot)
//     epoch [NBLK]                            this rank's launch counter, one copy per block (private)
//     err   [1]                               non-zero after a flag wait timed out
//
// Block b of the launch owns chunk b of the message: it copies the chunk into slot `rank` of every peer (and of itself), fences
// at system scope, releases flags[set][rank][b] = epoch on every peer, acquires flags[set][q][b] == epoch for all q locally, then
// reduces its chunk.  No grid barrier, no host state: the epoch lives in device memory, so the launch can sit in a hipGraph.
// Two sets suffice: a block reaches epoch k + 1 only after it has seen every peer's flag of epoch k, which a peer raises only
// after its own launch k - 1 -- the last reader of set (k + 1) & 1 -- has completed.
//
// The buffer is allocated uncached / fine-grained where the runtime allows it (remote writes must not hide behind the local
// L2); flags and data are accessed with system-scope release / acquire.  Validated here with several processes sharing ONE
// MI355X (no multi-GPU box was available to the builder); the default exchange remains RCCL (HARL_ALLREDUCE=oneshot opts in).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <new>
#include "common.h"
#include "../../include/harl_hip.h"

using namespace harl;

namespace {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
constexpr int MAX_RANKS = 16, MAX_BLK = 32, CM_THREADS = 256;
// A flag wait gives up after `wait_ticks` of the constant 100 MHz counter (err word set, NaN result) -- 0 = wait for ever, as
// RCCL does.  The default is 10 minutes (harl_comm_set_timeout / HARL_ONESHOT_TIMEOUT_S): a rank that arrives seconds late at a
// collective (rank-0 evaluation, a checkpoint, a first-call build) is ordinary and must not be turned into NaN gradients.  A
// time-out is FATAL for the communicator: the late peer's flags and this rank's epochs are out of step from then on, every
// later exchange times out as well -- the Python side checks harl_comm_status() at every host synchronisation point of
// compute() / train() and raises (harl_amd/dist.py).
constexpr long long DEFAULT_WAIT_TICKS = 600LL * 100000000LL;

struct CommDev {
  char *peer[MAX_RANKS];  // base address of every rank's buffer in THIS process (peer[rank] = the local one)
  int world, rank, nblk;
  long cap;               // bytes per slot
  long long wait_ticks;   // flag-wait limit in 100 MHz ticks, 0 = none
};

struct CommHost {
  CommDev d;
  void *local;
  int kind;  // 2 uncached, 1 fine-grained, 0 plain device memory
  bool opened[MAX_RANKS];
};

__host__ __device__ inline long data_bytes(int world, long cap) { return 2L * world * cap; }
__host__ __device__ inline long flags_off(int world, long cap) { return data_bytes(world, cap); }
__host__ __device__ inline long epoch_off(int world, long cap) { return flags_off(world, cap) + 2L * world * MAX_BLK * 4; }
__host__ __device__ inline long err_off(int world, long cap) { return epoch_off(world, cap) + MAX_BLK * 4; }
inline long total_bytes(int world, long cap) { return err_off(world, cap) + 64; }

int bad(const char *m) {
  set_error(m);
  return -2;
}

template <typename T>
__global__ __launch_bounds__(CM_THREADS) void k_oneshot_allreduce(CommDev c, T *__restrict__ msg, long n) {
  constexpr int VEC = 16 / sizeof(T);
  const int b = blockIdx.x, P = c.world, tid = threadIdx.x;
  char *loc = c.peer[c.rank];
  unsigned *ep = reinterpret_cast<unsigned *>(loc + epoch_off(P, c.cap)) + b;
  const unsigned epoch = *ep + 1u;  // (private word: only this block of this rank's launches touches it)
  const int set = epoch & 1;
  // chunk b, in units of 16 bytes where the length allows it
  const long nv = (n + VEC - 1) / VEC, per = (nv + c.nblk - 1) / c.nblk;
  const long v0 = (long)b * per, v1 = v0 + per < nv ? v0 + per : nv;
  const long slot_me = ((long)set * P + c.rank) * c.cap;
  // ---- push: my chunk into slot `rank` of every buffer, the next rank first (every rank starts on a different link)
  for (int k = 1; k <= P; ++k) {
    const int q = (c.rank + k) % P;
    T *dst = reinterpret_cast<T *>(c.peer[q] + slot_me);
    for (long v = v0 + tid; v < v1; v += CM_THREADS) {
      const long e = v * VEC;
      if (e + VEC <= n) {
        const u32x4v x = *reinterpret_cast<const u32x4v *>(msg + e);
        __builtin_nontemporal_store(x, reinterpret_cast<u32x4v *>(dst + e));
      } else {
        for (long j = e; j < n; ++j) dst[j] = msg[j];
      }
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid < P) {
    unsigned *f = reinterpret_cast<unsigned *>(c.peer[tid] + flags_off(P, c.cap)) + ((long)set * P + c.rank) * MAX_BLK + b;
    __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- wait for the P chunks to land here
  __shared__ int timed_out;
  if (tid == 0) timed_out = 0;
  __syncthreads();
  if (tid < P) {
    unsigned *f = reinterpret_cast<unsigned *>(loc + flags_off(P, c.cap)) + ((long)set * P + tid) * MAX_BLK + b;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (c.wait_ticks > 0 && (long long)__builtin_amdgcn_s_memrealtime() - t0 > c.wait_ticks) {
        timed_out = 1;
        __hip_atomic_store(reinterpret_cast<int *>(loc + err_off(P, c.cap)), 1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // ---- reduce: the P local slots in rank order (the same order, hence the same bits, on every rank)
  const bool fail = timed_out != 0;
  for (long v = v0 + tid; v < v1; v += CM_THREADS) {
    const long e = v * VEC;
    const int m = e + VEC <= n ? VEC : (int)(n - e);
    T acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = T(0);
    for (int q = 0; q < P; ++q) {
      const T *src = reinterpret_cast<const T *>(loc + ((long)set * P + q) * c.cap) + e;
      if (m == VEC) {
        const u32x4v x = *reinterpret_cast<const u32x4v *>(src);
        T t[VEC];
        memcpy(t, &x, 16);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = q == 0 ? t[j] : acc[j] + t[j];
      } else {
        for (int j = 0; j < m; ++j) acc[j] = q == 0 ? src[j] : acc[j] + src[j];
      }
    }
    for (int j = 0; j < m; ++j) msg[e + j] = fail ? T(__builtin_nanf("")) : acc[j];
  }
  if (tid == 0) *ep = epoch;
}

}  // namespace

// Collective set-up, step 1 (host pointers): allocate this rank's buffer for messages of up to cap_bytes, write its 64-byte
// hipIpc handle to handle_out and the context pointer to *ctx_out.  Returns the allocation kind (2 uncached, 1 fine-grained,
// 0 plain device memory) or a negative error.
extern "C" int harl_comm_create(int world, int rank, long cap_bytes, int n_blocks, void *handle_out, void *ctx_out) {
  if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world || cap_bytes <= 0 || n_blocks < 1 || n_blocks > MAX_BLK ||
      !handle_out || !ctx_out)
    return bad("harl_comm_create: bad arguments (world <= 16, n_blocks <= 32)");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  CommHost *h = new (std::nothrow) CommHost();
  if (!h) return bad("harl_comm_create: out of host memory");
  memset(h, 0, sizeof(*h));
  h->d.world = world;
  h->d.rank = rank;
  h->d.nblk = n_blocks;
  h->d.cap = (cap_bytes + 255) / 256 * 256;
  h->d.wait_ticks = DEFAULT_WAIT_TICKS;
  const size_t bytes = (size_t)total_bytes(world, h->d.cap);
  hipIpcMemHandle_t hd;
  const unsigned flags[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, hipDeviceMallocDefault};
  int kind = -1;
  for (int a = 0; a < 3 && kind < 0; ++a) {
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, flags[a]) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    if (hipIpcGetMemHandle(&hd, p) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(p);
      continue;
    }
    h->local = p;
    kind = 2 - a;
  }
  if (kind < 0) {
    delete h;
    return bad("harl_comm_create: no exportable device allocation (hipIpcGetMemHandle failed; HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
  }
  if (hipMemset(h->local, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(h->local);
    delete h;
    return bad("harl_comm_create: clearing the buffer failed");
  }
  h->kind = kind;
  h->d.peer[rank] = static_cast<char *>(h->local);
  memcpy(handle_out, &hd, 64);
  *static_cast<void **>(ctx_out) = h;
  return kind;
}

// Step 2, after the ranks have exchanged their handles (all_handles: world x 64 bytes, rank-major, host memory): map the peers.
// Every rank's harl_comm_create has returned by then, so every buffer is cleared before the first message arrives.
extern "C" int harl_comm_connect(void *ctx, const void *all_handles) {
  CommHost *h = static_cast<CommHost *>(ctx);
  if (!h || !all_handles) return bad("harl_comm_connect: bad arguments");
  for (int q = 0; q < h->d.world; ++q) {
    if (q == h->d.rank) continue;
    hipIpcMemHandle_t hd;
    memcpy(&hd, static_cast<const char *>(all_handles) + 64L * q, 64);
    void *p = nullptr;
    if (hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      (void)hipGetLastError();
      return bad("harl_comm_connect: hipIpcOpenMemHandle failed");
    }
    h->d.peer[q] = static_cast<char *>(p);
    h->opened[q] = true;
  }
  return 0;
}

// In-place SUM over the ranks of msg[0..n) (fp32, or fp64 when is_f64), enqueued on `stream`; every rank must call it with the
// same n, in the same order.  n * element size <= the capacity given to harl_comm_create.
extern "C" int harl_comm_allreduce(void *ctx, void *msg, long n, int is_f64, void *stream) {
  CommHost *h = static_cast<CommHost *>(ctx);
  if (!h || !msg || n < 0) return bad("harl_comm_allreduce: bad arguments");
  if (n == 0) return 0;
  if (n * (is_f64 ? 8 : 4) > h->d.cap) return bad("harl_comm_allreduce: message longer than the capacity of the communicator");
  if ((reinterpret_cast<uintptr_t>(msg) & 15) != 0) return bad("harl_comm_allreduce: message must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (is_f64) hipLaunchKernelGGL(k_oneshot_allreduce<double>, dim3(h->d.nblk), dim3(CM_THREADS), 0, s, h->d, static_cast<double *>(msg), n);
  else hipLaunchKernelGGL(k_oneshot_allreduce<float>, dim3(h->d.nblk), dim3(CM_THREADS), 0, s, h->d, static_cast<float *>(msg), n);
  return check_launch("harl_comm_allreduce");
}

// Flag-wait limit of every later harl_comm_allreduce launch of this communicator, in seconds (0 = wait for ever).  Host-side
// state only: no device work, no synchronisation.
extern "C" int harl_comm_set_timeout(void *ctx, double seconds) {
  CommHost *h = static_cast<CommHost *>(ctx);
  if (!h || !(seconds >= 0.0) || seconds > 1e6) return bad("harl_comm_set_timeout: bad arguments (0 <= seconds <= 1e6)");
  h->d.wait_ticks = (long long)(seconds * 1e8);
  return 0;
}

// 0 = healthy; q + 1 = a wait for rank q's flag timed out (results of that launch are NaN).  Synchronises the device.
extern "C" int harl_comm_status(void *ctx) {
  CommHost *h = static_cast<CommHost *>(ctx);
  if (!h) return bad("harl_comm_status: bad arguments");
  int e = 0;
  if (hipMemcpy(&e, static_cast<char *>(h->local) + err_off(h->d.world, h->d.cap), 4, hipMemcpyDeviceToHost) != hipSuccess)
    return bad("harl_comm_status: read-back failed");
  return e;
}

extern "C" int harl_comm_destroy(void *ctx) {
  CommHost *h = static_cast<CommHost *>(ctx);
  if (!h) return 0;
  (void)hipDeviceSynchronize();
  for (int q = 0; q < h->d.world; ++q)
    if (h->opened[q]) (void)hipIpcCloseMemHandle(h->d.peer[q]);
  if (h->local) (void)hipFree(h->local);
  delete h;
  return 0;
}
