"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The update shards ``n_rollout_threads`` across ranks (columns are independent in GAE, in the forward/backward
and in the factor product -- SURVEY.md §8e); parameters, Adam state and ValueNorm statistics are replicated.
The only exchange steps are SUM all-reduces of (a) the flat gradient arena with the loss scalars packed behind
it, (b) the advantage moments, (c) the ValueNorm batch sums.  With world_size == 1 everything is a no-op.
``HARL_ALLREDUCE=oneshot`` / ``auto`` send them through the hand-written one-hop exchange of csrc/comm.hip instead of RCCL's
ring (``auto``: only where every rank can, decided collectively -- see ``Comm``).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def _device_identity() -> str:
    """What tells two ranks' GPUs apart (the one-shot exchange may use plain device memory only when every rank sits on the SAME
    device -- the several-processes-on-one-GPU tests)."""
    import socket
    d = torch.cuda.current_device()
    try:
        ident = str(torch.cuda.get_device_properties(d).uuid)
    except Exception:  # noqa: BLE001 -- older builds have no uuid attribute
        ident = "%s/%d" % (os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")), d)
    return socket.gethostname() + ":" + ident


class Comm:
    """Thin wrapper so the algorithm code is identical for 1 and N ranks (and for gloo in CPU tests).

    ``HARL_ALLREDUCE`` selects the exchange of the update's small messages: ``rccl`` (default) = the backend's all-reduce;
    ``oneshot`` = the hand-written one-hop exchange of csrc/comm.hip, an error where it cannot be set up; ``auto`` = the
    one-shot exchange when EVERY rank obtained an uncached / fine-grained exportable buffer (or all ranks share one device)
    and every peer mapping succeeded, the backend's all-reduce otherwise -- decided collectively, so all ranks take the
    same path."""

    def __init__(self, group=None):
        self.enabled = (dist.is_available() and dist.is_initialized()
                        and (dist.get_world_size(group) > 1 or os.environ.get("HARL_DIST_SINGLE") == "1"))
        self.group = group
        self.world_size = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.oneshot = None  # (lib, ctx, capacity, allocation kind) of the one-shot exchange, see enable_oneshot
        self.oneshot_info = None  # what enable_oneshot decided and why (kinds per rank, devices, fallback reason)
        mode = os.environ.get("HARL_ALLREDUCE", "rccl")
        if mode not in ("rccl", "oneshot", "auto"):
            raise ValueError(f"HARL_ALLREDUCE={mode!r}: expected rccl, oneshot or auto")
        if self.enabled and mode != "rccl":
            self.enable_oneshot(required=mode == "oneshot")

    def second_group(self) -> "Comm":
        """A communicator of its own over the same ranks (``dist.new_group`` is COLLECTIVE: every rank must call this at the
        same point of its program).  The callers are the CONSTRUCTORS of ``harl_amd.runner.OnPolicyHARunner`` /
        ``OnPolicyMARunner`` and of the drop-in's runner classes (``harl_amd.dropin``: right after the reference's own
        constructor) -- never a lazy path, where ranks may arrive at different times.
        Opt-in (``HARL_CRITIC_GROUP=1``): with it the critic's update keeps a stream of its own under data parallelism,
        its collectives living in a queue ordered only among themselves.  Two communicators in flight on one device are
        outside what NCCL / RCCL document as safe (the two collectives may start in a different order on different ranks and
        hang if their kernels cannot co-run), and no multi-GPU box has validated it, so the DEFAULT is one communicator and
        one stream under data parallelism (the round-3 behaviour; single-GPU runs keep the critic stream either way).
        Returns ``self`` when there is nothing to split."""
        if not self.enabled or os.environ.get("HARL_CRITIC_GROUP", "0") != "1":
            return self
        g = dist.new_group(ranks=None if self.group is None else dist.get_process_group_ranks(self.group),
                           backend=dist.get_backend(self.group))
        return Comm(g)

    def enable_oneshot(self, cap_bytes: int = 1 << 20, n_blocks: int = 8, required: bool = True) -> "Comm":
        """Route SUM all-reduces of fp32 / fp64 device tensors of up to ``cap_bytes`` through the hand-written one-shot exchange
        (csrc/comm.hip: every rank pushes its message into a slot of every peer's hipIpc-mapped buffer and sums the slots in
        rank order -- one hop over xGMI instead of RCCL's ring for messages that are latency-bound).
        COLLECTIVE (two ``all_gather_object`` rounds + a barrier): call it at the same point on every rank -- ``Comm.__init__``
        does when ``HARL_ALLREDUCE`` is ``oneshot`` / ``auto``, i.e. wherever a ``Comm()`` is constructed (the runners'
        constructors).  Every decision below is taken from the GATHERED table, so all ranks agree:
          * a rank whose buffer is plain (coarse-grained) device memory -- kind 0 -- is accepted only when all ranks sit on ONE
            device (the several-processes-per-GPU tests): a peer's xGMI writes into coarse-grained memory are not guaranteed
            to be visible to a running kernel's loads;
          * any rank that could not allocate, export or map -> nobody uses the exchange.
        ``required`` (HARL_ALLREDUCE=oneshot): raise on every rank instead of falling back to the backend's all-reduce.
        Falls back to the backend for anything the exchange does not take (CPU tensors, larger messages, other dtypes)."""
        if not self.enabled or self.oneshot is not None or not torch.cuda.is_available():
            return self
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        handle = C.create_string_buffer(64)
        ctx = C.c_void_p()
        kind = lib.harl_comm_create(self.world_size, self.rank, cap_bytes, n_blocks, handle, C.byref(ctx))
        err = "" if kind >= 0 else "harl_comm_create: " + lib.harl_last_error().decode()
        gathered = [None] * self.world_size
        dist.all_gather_object(gathered, (self.rank, bytes(handle.raw), int(kind), _device_identity(), err), group=self.group)
        gathered.sort()
        kinds = [g[2] for g in gathered]
        devices = [g[3] for g in gathered]
        one_device = len(set(devices)) == 1
        reason = None
        if min(kinds) < 0:
            reason = "; ".join(f"rank {g[0]}: {g[4]}" for g in gathered if g[2] < 0)
        elif min(kinds) == 0 and not one_device:
            reason = ("rank(s) %s obtained only plain (coarse-grained) device memory for the exchange buffer and the ranks span "
                      "%d devices: remote writes would not be guaranteed visible" % ([g[0] for g in gathered if g[2] == 0], len(set(devices))))
        rc = 0
        if reason is None:
            rc = lib.harl_comm_connect(ctx, b"".join(g[1] for g in gathered))
        rcs = [None] * self.world_size
        dist.all_gather_object(rcs, (self.rank, int(rc), "" if rc == 0 else "harl_comm_connect: " + lib.harl_last_error().decode()),
                               group=self.group)
        if reason is None and any(r[1] != 0 for r in rcs):
            reason = "; ".join(f"rank {r[0]}: {r[2]}" for r in sorted(rcs) if r[1] != 0)
        self.oneshot_info = dict(kinds=kinds, devices=len(set(devices)), enabled=reason is None, fallback_reason=reason)
        if reason is not None:
            if kind >= 0:
                lib.harl_comm_destroy(ctx)
            if required:
                raise RuntimeError("HARL_ALLREDUCE=oneshot: the one-shot exchange cannot be set up -- " + reason)
            if self.rank == 0:
                import warnings
                warnings.warn("HARL_ALLREDUCE=auto: falling back to the backend's all-reduce -- " + reason)
            return self
        t_out = os.environ.get("HARL_ONESHOT_TIMEOUT_S")
        if t_out is not None and lib.harl_comm_set_timeout(ctx, float(t_out)) != 0:
            raise RuntimeError("harl_comm_set_timeout: " + lib.harl_last_error().decode())
        dist.barrier(group=self.group)
        self.oneshot = (lib, ctx, cap_bytes, kind)
        return self

    def oneshot_status(self) -> int:
        """0, or q + 1 once a wait for rank q's message has timed out (synchronises the device)."""
        return 0 if self.oneshot is None else int(self.oneshot[0].harl_comm_status(self.oneshot[1]))

    def check(self) -> None:
        """Raise if a flag wait of the one-shot exchange has timed out (its results were NaN from then on, and the
        communicator is unusable: flags and epochs are out of step).  Called by compute() / train() right after their own
        host synchronisation, so the status read costs one 4-byte copy; a no-op on every other path."""
        if self.oneshot is None:
            return
        st = self.oneshot_status()
        if st != 0:
            raise RuntimeError(
                f"one-shot all-reduce: rank {self.rank} gave up waiting for rank {st - 1}'s message (time-out "
                f"{os.environ.get('HARL_ONESHOT_TIMEOUT_S', '600')} s; HARL_ONESHOT_TIMEOUT_S=0 waits for ever).  The reduced "
                "gradients of that step were NaN and the communicator cannot be used again: stop the job.")

    def close(self) -> None:
        if self.oneshot is not None:
            lib, ctx = self.oneshot[0], self.oneshot[1]
            self.oneshot = None
            torch.cuda.synchronize()
            dist.barrier(group=self.group)  # nobody unmaps a buffer a peer's last launch still writes to
            lib.harl_comm_destroy(ctx)

    def _all_reduce(self, t: torch.Tensor) -> None:
        # the path is chosen from RANK-INVARIANT properties only (device kind, dtype, byte count): a tensor that is misaligned
        # or non-contiguous on one rank goes through an aligned staging copy instead of another backend
        if (self.oneshot is not None and t.is_cuda and t.dtype in (torch.float32, torch.float64)
                and 0 < t.numel() * t.element_size() <= self.oneshot[2]):
            lib, ctx = self.oneshot[0], self.oneshot[1]
            direct = t.is_contiguous() and t.data_ptr() % 16 == 0
            buf = t if direct else t.reshape(-1).clone(memory_format=torch.contiguous_format)  # (allocator blocks are 512-byte aligned)
            rc = lib.harl_comm_allreduce(ctx, buf.data_ptr(), buf.numel(), int(t.dtype == torch.float64),
                                         torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError("harl_comm_allreduce: " + lib.harl_last_error().decode())
            if not direct:
                t.copy_(buf.view(t.shape))
            return
        # RCCL reduces device tensors in place; the gloo backend (CPU tests, and the 2-ranks-on-1-GPU parity test)
        # is routed through host memory, which works for every build of gloo
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.enabled:
            self._all_reduce(t)
        return t

    def all_reduce_message(self, msg: torch.Tensor) -> None:
        """One collective per optimiser step over the contiguous fp32 message [folded gradients | 4 x scalar pieces]
        (nets.dwp_msg): the gradients are produced in place at its head, the fp64 loss scalars are split by
        harl_pack_scalars_hilo into four fp32 pieces on a FIXED exponent grid (each an integer multiple of its quantum
        below 2^20, so the fp32 SUM over <= 16 ranks is exact: the reduced value is the fp64 sum of the ranks' scalars to
        2^-36 absolute), and harl_adam_fold reads both straight from the reduced message -- no staging copies around the
        collective."""
        if self.enabled:
            self._all_reduce(msg)


SCALAR_QUANTA = (2.0 ** 24, 2.0 ** 4, 2.0 ** -16, 2.0 ** -36)


def pack_message_reference(flat_grad: torch.Tensor, scalars64: torch.Tensor) -> torch.Tensor:
    """Host restatement of the message format (used by the CPU tests): [grad | piece_0 | .. | piece_3] with
    piece_k = trunc(r_k / q_k) * q_k, r_0 = s, r_{k+1} = r_k - piece_k  (k_pack_scalars_hilo, csrc/elementwise.hip)."""
    r = scalars64.to(torch.float64).clone()
    pieces = []
    for q in SCALAR_QUANTA:
        p = torch.trunc(r / q) * q
        pieces.append(p.to(torch.float32))
        r = r - p
    return torch.cat([flat_grad.to(torch.float32)] + pieces)


def unpack_message_reference(msg: torch.Tensor, n: int, k: int):
    s = torch.zeros(k, dtype=torch.float64)
    for i in reversed(range(len(SCALAR_QUANTA))):  # smallest first
        s = s + msg[n + i * k:n + (i + 1) * k].to(torch.float64)
    return msg[:n], s


def shard_columns(n_rollout_threads: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Rank r owns rollout columns [lo, hi): contiguous, sizes differ by at most one."""
    base, rem = divmod(n_rollout_threads, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_minibatch_rows(global_idx: torch.Tensor, n_global: int, lo: int, hi: int, agents: int = 1) -> torch.Tensor:
    """Rows of a GLOBAL minibatch permutation (row = t*N + n over the unsharded buffer, identical on every rank
    and bit-identical to the reference's draw) that fall into this rank's column range, re-indexed to the
    local [T, hi-lo] buffer.  Order within the minibatch is preserved.  ``agents`` > 1: FP critic buffers, whose rows
    are (t*N + n)*A + a."""
    ncol = n_global * agents
    t = torch.div(global_idx, ncol, rounding_mode="floor")
    c = global_idx - t * ncol
    n = torch.div(c, agents, rounding_mode="floor") if agents > 1 else c
    keep = (n >= lo) & (n < hi)
    a = c[keep] - n[keep] * agents if agents > 1 else 0
    return (t[keep] * (hi - lo) + (n[keep] - lo)) * agents + a


def init_from_env(backend: Optional[str] = None) -> Comm:
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    # HARL_DIST_SINGLE=1: build the group even for ONE rank, so that the RCCL branch of the update (message packing,
    # all_reduce, hi/lo scalar pieces) runs on a one-GPU box exactly as it does on eight
    single = ws == 1 and os.environ.get("HARL_DIST_SINGLE") == "1" and "RANK" in os.environ
    if (ws > 1 or single) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)
    return Comm()
