"""Device-resident on-policy rollout buffers with the reference's attribute names and shapes.

Reference: harl/common/buffers/on_policy_actor_buffer.py:9-178, on_policy_critic_buffer_ep.py:8-250.
Storage is ``[T(+1), N, D]`` row-major fp32 *device* tensors (time-major, thread-minor, feature-contiguous:
flattening the first two axes gives row = t*N + n, the indexing the reference's generators use), so the update
kernels read them in place: no per-minibatch gather copy, no host<->device traffic inside ``train()``.
Minibatch sampling is bit-exact with the reference: the same ``torch.randperm`` draws on the global CPU generator
in the same order; only the resulting int64 index arrays are uploaded.
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream
from .valuenorm import ValueNorm, _as_dev


def _obs_shape(space) -> Tuple[int, ...]:
    name = space.__class__.__name__
    if name == "Box":
        shp = tuple(space.shape)
    elif name == "list":
        shp = tuple(space)
    else:
        raise NotImplementedError(name)
    if isinstance(shp[-1], list):
        shp = shp[:1]
    return shp


def _act_shape(space) -> int:
    name = space.__class__.__name__
    if name == "Discrete":
        return 1
    return int(space.shape[0])


_ADVANCE_OK: Optional[bool] = None


def _advance_matches_randperm() -> bool:
    """One-time self-check (global RNG state saved/restored): does drawing n-1 int32 randoms advance the CPU
    generator exactly like torch.randperm(n)?  (ATen's randperm_cpu draws generator->random() once per swap.)"""
    global _ADVANCE_OK
    if _ADVANCE_OK is None:
        st = torch.get_rng_state()
        try:
            torch.manual_seed(987654321)
            torch.randperm(1031)
            a = torch.randperm(17)
            torch.manual_seed(987654321)
            torch.empty(1030, dtype=torch.int32).random_()
            b = torch.randperm(17)
            _ADVANCE_OK = bool(torch.equal(a, b))
        finally:
            torch.set_rng_state(st)
    return _ADVANCE_OK


_SKIP_OK: Optional[bool] = None


def _skip_matches_random() -> bool:
    """One-time self-check (global RNG state saved/restored): does libharl_hip.so::harl_rng_advance leave the generator
    exactly where ``Tensor.random_`` over as many int32 elements does (mid-block start, several block refills)?"""
    global _SKIP_OK
    if _SKIP_OK is None:
        st = torch.get_rng_state()
        try:
            ok = True
            lib = _lib.load()
            for seed, n in ((987654321, 1030), (5, 70000), (7, 624 * 2600 + 11)):  # the last one takes the polynomial jump
                torch.manual_seed(seed)
                torch.empty(17, dtype=torch.int32).random_()  # start mid-block (no torch.randperm here: callers may tap it)
                mid = torch.get_rng_state()
                torch.empty(n, dtype=torch.int32).random_()
                want = torch.get_rng_state()
                out = torch.empty_like(mid)
                rc = lib.harl_rng_advance(mid.data_ptr(), mid.numel(), n, out.data_ptr())
                ok = ok and rc == 0 and bool(torch.equal(out, want))
            _SKIP_OK = ok
        except Exception:  # noqa: BLE001 -- library without the entry point / another state layout: use Tensor.random_
            _SKIP_OK = False
        finally:
            torch.set_rng_state(st)
    return _SKIP_OK


def _advance_generator(n: int) -> None:
    """Advance the global CPU generator by n 32-bit draws.  The C skip-ahead only refreshes the mt19937 state blocks
    (~0.1 ms per 819200 draws) and jumps ahead polynomially beyond ~1.5 M draws; ``Tensor.random_`` tempers and stores every
    output (~1 ms per 819200)."""
    if os.environ.get("HARL_RNG_SKIP", "1") != "0" and _skip_matches_random():
        st = torch.get_rng_state()
        out = torch.empty_like(st)
        if _lib.load().harl_rng_advance(st.data_ptr(), st.numel(), n, out.data_ptr()) == 0:
            torch.set_rng_state(out)
            return
    # fallback (self-check failed / HARL_RNG_SKIP=0): draw and discard, in pieces -- the total of an 8-GPU train() is ~130 M
    # draws, 0.5 GB as one tensor
    chunk = 1 << 20
    buf = torch.empty(min(n, chunk), dtype=torch.int32)
    left = n
    while left > 0:
        m = min(left, chunk)
        buf[:m].random_()
        left -= m


# CONTRACT of the deferred advances below: between a runner-internal train(_defer=True) and the next rng_sync() the global CPU
# generator is BEHIND by the pending draws.  Every public entry point of this package that returns to user code -- the runner's
# train() / collect() / warmup(), HAPPO/HATRPO/MAPPO/VCritic.train(), the buffers' generators -- calls rng_sync() first or last,
# so user code (callbacks, custom runners, anything that draws from torch's CPU generator) always sees the state the reference
# would have; only code that calls the underscore-prefixed internals itself has to call ``buffers.rng_sync()`` before drawing.
# Deferred generator advances: with one full-buffer minibatch the permutation of a sampler call is never looked at, only the
# generator has to end up where torch.randperm would leave it -- and the state after several such calls depends on the TOTAL
# number of draws only.  consume_randperm() therefore just adds to this counter; rng_sync() -- called before ANY use of the
# global CPU generator inside this package and at the end of every train() -- applies the total in ONE harl_rng_advance, which
# jumps ahead in ~0.1-0.3 ms whatever the count is (GF(2) polynomial jump, csrc/host_rng.hip; the polynomial of a given total is
# built once and cached, and the total of a train() is the same every time).  Round 2 advanced call by call on a background
# thread: 0.8 ms per call for the 6.5 M draws of an 8-GPU run's global batch, 20 calls per update.
_PENDING_DRAWS = 0


def rng_sync() -> None:
    """Apply every deferred generator advance (call before ANY use of the global CPU RNG)."""
    global _PENDING_DRAWS
    if _PENDING_DRAWS:
        n, _PENDING_DRAWS = _PENDING_DRAWS, 0
        _advance_generator(n)


def consume_randperm(batch_size: int, deferred: bool = True) -> None:
    """Advance the global CPU generator exactly as ``torch.randperm(batch_size)`` would, without materialising
    the permutation.  Used when ``num_mini_batch == 1``: the single minibatch is the whole buffer, so the update
    does not depend on the order, but every later draw of the run (agent order, next epoch's permutation) must
    still see the same generator state as in the reference.  randperm(819200) costs ~30-70 ms of host time per
    draw (20 draws per train()); deferred advances are summed and applied by the next ``rng_sync()`` in one jump.
    Falls back to a real randperm if the self-check fails."""
    global _PENDING_DRAWS
    if batch_size <= 1:
        return
    if not _advance_matches_randperm():
        rng_sync()
        torch.randperm(batch_size)
        return
    if not deferred:
        rng_sync()
        _advance_generator(batch_size - 1)
        return
    _PENDING_DRAWS += batch_size - 1


_REPLAY_OK: Optional[bool] = None
_REPLAY_BUFS: dict = {}
PERM_TAP = None  # test hook: called with every permutation this module materialises (CPU int64 copy)


def _replay_raw(n: int, set_state: bool = True) -> torch.Tensor:
    """ATen's randperm_cpu replayed by libharl_hip.so::harl_randperm_replay from the current generator state; advances
    the generator exactly like torch.randperm(n).  Returns a reusable CPU int32 buffer (valid until the next call)."""
    lib = _lib.load()
    if n not in _REPLAY_BUFS:
        pin = torch.cuda.is_available()
        _REPLAY_BUFS[n] = (torch.empty(n, dtype=torch.int32, pin_memory=pin), torch.empty(n, dtype=torch.int32))
    out, scratch = _REPLAY_BUFS[n]
    st = torch.get_rng_state()
    st2 = torch.empty_like(st)
    rc = lib.harl_randperm_replay(st.data_ptr(), st.numel(), n, out.data_ptr(), scratch.data_ptr(), st2.data_ptr())
    if rc != 0:
        raise RuntimeError("harl_randperm_replay: unexpected CPU generator state layout")
    if set_state:
        torch.set_rng_state(st2)
    return out


def _replay_matches_randperm() -> bool:
    """One-time self-check (global RNG state saved/restored): permutation AND final generator state of the replay are
    those of torch.randperm, from a state that is mid-block and across several mt19937 block refills."""
    global _REPLAY_OK
    if _REPLAY_OK is None:
        st = torch.get_rng_state()
        try:
            ok = True
            for seed, n in ((987654321, 1031), (5, 70001)):
                torch.manual_seed(seed)
                torch.randperm(17)
                mid = torch.get_rng_state()
                a = _replay_raw(n).clone()
                sa = torch.get_rng_state()
                torch.set_rng_state(mid)
                b = torch.randperm(n)
                ok = ok and bool(torch.equal(a.long(), b)) and bool(torch.equal(sa, torch.get_rng_state()))
            _REPLAY_OK = ok
        except Exception:  # noqa: BLE001 -- any surprise (torch build with another state layout): use torch.randperm
            _REPLAY_OK = False
        finally:
            torch.set_rng_state(st)
    return _REPLAY_OK


def draw_permutation(n: int, device=None) -> torch.Tensor:
    """``torch.randperm(n)`` on the global CPU generator (bit-identical values and generator state), as an int64 tensor
    on ``device`` (CPU if None).  Large n goes through the C replay: ATen fans the permutation's initialisation out to
    its thread pool, whose wake-up made each 819200-element draw cost ~27 ms inside train() on the 256-thread GPU host;
    the replay costs ~5 ms and uploads 4 bytes per index instead of 8."""
    rng_sync()
    if n >= 65536 and _replay_matches_randperm():
        p32 = _replay_raw(n)
        if PERM_TAP is not None:
            PERM_TAP(p32.long())
        # (NumPy for the widening copy on the host: ATen fans a 65 536-element `.long()` out to its intra-op pool, and waking
        # 128 threads cost 30 - 70 ms per draw on the GPU box's host -- 45 draws per recurrent update at 4096 rollout threads
        # were 250 ms of host time around 120 ms of GPU work, round 6)
        return torch.from_numpy(p32.numpy().astype(np.int64)) if device is None else p32.to(device).long()
    p = torch.randperm(n)
    if PERM_TAP is not None:
        PERM_TAP(p.clone())
    return p if device is None else p.to(device)


def minibatch_indices(batch_size: int, num_mini_batch: int, device=None) -> List[torch.Tensor]:
    """One ``torch.randperm(batch_size)`` draw on the global CPU generator, remainder rows dropped
    (on_policy_actor_buffer.py:121-135).  Returns int64 tensors (views of one permutation) on ``device`` (CPU if None)."""
    assert batch_size >= num_mini_batch, (
        f"batch size ({batch_size}) must be >= the number of mini batches ({num_mini_batch})")
    m = batch_size // num_mini_batch
    rand = draw_permutation(batch_size, device)
    return [rand[i * m:(i + 1) * m] for i in range(num_mini_batch)]


def recurrent_first_rows(T: int, N: int, num_mini_batch: int, data_chunk_length: int, naive: bool):
    """Sequence starts of the reference's recurrent samplers, as rows of the t-major flattening (row = t*N + n).
    chunked (on_policy_actor_buffer.py:223-326): arrays are cast thread-major [N*T] and chunk c is rows [c*L, (c+1)*L)
    there -> thread n = (c*L)//T, start time t0 = (c*L)%T; one randperm(T*N//L) on the global CPU generator.
    naive (:180-221): whole columns, L = T, one randperm(N).  Yields (first_rows CPU int64 [m], L)."""
    if naive:
        assert N >= num_mini_batch, f"n_rollout_threads ({N}) must be >= num_mini_batch ({num_mini_batch})"
        per = N // num_mini_batch
        rng_sync()
        perm = torch.randperm(N)
        for b in range(num_mini_batch):
            yield perm[b * per:(b + 1) * per], T
        return
    L = data_chunk_length
    assert T % L == 0, "episode_length must be a multiple of data_chunk_length"
    for chunks in minibatch_indices((T * N) // L, num_mini_batch):
        start = chunks.numpy() * L  # (host-side index arithmetic in NumPy: see draw_permutation)
        yield torch.from_numpy((start % T) * N + start // T), L


def _recurrent_seqs(device, T: int, n_local: int, agents: int, H: int, num_mini_batch: int, data_chunk_length: int,
                    naive: bool, shard, h0_src: torch.Tensor, masks_src: torch.Tensor, cache: Optional[dict] = None):
    """Shared body of the buffers' ``recurrent_batches``: draws the reference's sequence starts over the GLOBAL column
    count (identical on every rank), keeps the sequences whose rollout thread lives on this rank and re-indexes them to
    the local buffers.  Yields nets.build_seq dicts with ``m_global`` (sequences in the global minibatch) added, or
    ``dict(empty=True, ...)`` when none of them is local.

    ONE minibatch (the tuned recurrent configurations: ``actor_num_mini_batch`` = ``critic_num_mini_batch`` = 1): the minibatch
    is EVERY sequence of the buffer, and the update -- sums and means over its rows, each sequence unrolled from its own stored
    state -- does not depend on the order the permutation puts them in.  So, as the feed-forward samplers have done since round
    1 (``consume_randperm``), only the GENERATOR is advanced by the reference's ``torch.randperm`` draw; the sequences are taken
    in buffer order.  That layout is the same in every epoch: the device-side tables (``harl_build_seq``) are built once per
    update and handed out again (``cache``: keyed on the source tensors' addresses and version counters), and because the row
    index tensor is the same OBJECT every epoch, the networks' normalised-input image (nets._x0n_image) is reused as well --
    per optimiser step of the 8-agent SMAC update that is two launches (8 + 33 us), a host-side permutation of 8 192 starts and
    its upload less.  ``HARL_RNN_ORDERED=0`` materialises the permutation as before (the results differ by summation order)."""
    from .nets import build_seq
    n_global, lo, hi = shard if shard else (n_local, 0, n_local)
    ncol_g, ncol_l = n_global * agents, n_local * agents
    if num_mini_batch == 1 and os.environ.get("HARL_RNN_ORDERED", "1") != "0":
        import numpy as np
        L = T if naive else data_chunk_length
        assert T % L == 0, "episode_length must be a multiple of data_chunk_length"
        n_seq = ncol_g if naive else (T * ncol_g) // L
        consume_randperm(n_seq)  # the generator advance of the reference's single torch.randperm(n_seq)
        key = (str(device), T, n_local, agents, H, L, bool(naive), tuple(shard) if shard else None, h0_src.data_ptr(),
               h0_src._version, masks_src.data_ptr(), masks_src._version)
        if cache is not None and cache.get("key") == key:
            yield cache["seq"]
            return
        start = np.arange(n_seq, dtype=np.int64) * L  # sequence c: thread-major rows [c L, (c + 1) L) (recurrent_first_rows)
        first = start if naive else (start % T) * ncol_g + start // T
        if naive:
            first = np.arange(n_seq, dtype=np.int64)
        if shard:
            t0 = first // ncol_g
            c = first - t0 * ncol_g
            n = c // agents
            keep = (n >= lo) & (n < hi)
            first = t0[keep] * ncol_l + (n[keep] - lo) * agents + (c[keep] - n[keep] * agents)
        if first.size == 0:
            seq = dict(empty=True, L=L, m=0, m_global=n_seq)
        else:
            seq = build_seq(device, L, int(first.size), H, first_rows=torch.from_numpy(np.ascontiguousarray(first)), stride=ncol_l,
                            h0_src=h0_src, masks_src=masks_src)
            seq["m_global"] = n_seq
        if cache is not None:
            cache.clear()
            cache.update(key=key, seq=seq, src=(h0_src, masks_src))  # (the sources stay alive: their addresses are the key)
        yield seq
        return
    for first, L in recurrent_first_rows(T, ncol_g, num_mini_batch, data_chunk_length, naive):
        m_global = first.numel()
        if shard:
            t0 = torch.div(first, ncol_g, rounding_mode="floor")
            c = first - t0 * ncol_g
            n = torch.div(c, agents, rounding_mode="floor")
            keep = (n >= lo) & (n < hi)
            first = t0[keep] * ncol_l + (n[keep] - lo) * agents + (c[keep] - n[keep] * agents)
        if first.numel() == 0:
            yield dict(empty=True, L=L, m=0, m_global=m_global)
            continue
        seq = build_seq(device, L, first.numel(), H, first_rows=first, stride=ncol_l, h0_src=h0_src, masks_src=masks_src)
        seq["m_global"] = m_global
        yield seq


class OnPolicyActorBuffer:
    def __init__(self, args: dict, obs_space, act_space, device=None):
        self.device = torch.device(device) if device is not None else _lib.default_device()
        _lib.require_gpu(self.device)
        self.episode_length = T = args["episode_length"]
        self.n_rollout_threads = N = args["n_rollout_threads"]
        self.hidden_sizes = args["hidden_sizes"]
        self.rnn_hidden_size = self.hidden_sizes[-1]
        self.recurrent_n = args["recurrent_n"]
        obs_shape = _obs_shape(obs_space)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)  # noqa: E731
        self.obs = z(T + 1, N, *obs_shape)
        self.rnn_states = z(T + 1, N, self.recurrent_n, self.rnn_hidden_size)
        if act_space.__class__.__name__ == "Discrete":
            self.available_actions = torch.ones(T + 1, N, act_space.n, dtype=torch.float32, device=self.device)
        else:
            self.available_actions = None
        a = _act_shape(act_space)
        self.actions = z(T, N, a)
        self.action_log_probs = z(T, N, a)
        self.masks = torch.ones(T + 1, N, 1, dtype=torch.float32, device=self.device)
        self.active_masks = torch.ones_like(self.masks)
        self.factor = None
        self.step = 0

    def update_factor(self, factor):
        """Save factor for this actor (actor_buffer.py:78-80); a device tensor is aliased, not copied --
        the runner hands over a fresh tensor per agent."""
        self.factor = _as_dev(factor, self.device).reshape(self.episode_length, self.n_rollout_threads, 1)

    def insert(self, obs, rnn_states, actions, action_log_probs, masks, active_masks=None, available_actions=None):
        s = self.step
        self.obs[s + 1].copy_(_as_dev(obs, self.device))
        self.rnn_states[s + 1].copy_(_as_dev(rnn_states, self.device))
        self.actions[s].copy_(_as_dev(actions, self.device))
        self.action_log_probs[s].copy_(_as_dev(action_log_probs, self.device))
        self.masks[s + 1].copy_(_as_dev(masks, self.device))
        if active_masks is not None:
            self.active_masks[s + 1].copy_(_as_dev(active_masks, self.device))
        if available_actions is not None:
            self.available_actions[s + 1].copy_(_as_dev(available_actions, self.device))
        self.step = (s + 1) % self.episode_length

    def after_update(self):
        self.obs[0].copy_(self.obs[-1])
        self.rnn_states[0].copy_(self.rnn_states[-1])
        self.masks[0].copy_(self.masks[-1])
        self.active_masks[0].copy_(self.active_masks[-1])
        if self.available_actions is not None:
            self.available_actions[0].copy_(self.available_actions[-1])

    # flat [T*N, .] views (row = t*N + n), zero-copy
    def flat(self, name: str) -> torch.Tensor:
        t = getattr(self, name)
        if name in ("obs", "masks", "active_masks", "available_actions", "rnn_states"):
            t = t[:-1]
        return t.reshape(self.episode_length * self.n_rollout_threads, -1)

    def recurrent_batches(self, num_mini_batch: int, data_chunk_length: int, naive: bool = False, shard=None):
        """GRU-layout minibatches (nets.build_seq) for the chunked / naive recurrent samplers: only the m sequence
        starts travel to the device; the kernels gather rows first + l*N in place.  ``shard`` = (n_global, lo, hi):
        the GLOBAL sampler is drawn and filtered to this rank's rollout threads."""
        T, N = self.actions.shape[:2]
        return _recurrent_seqs(self.device, T, N, 1, self.rnn_hidden_size * self.recurrent_n, num_mini_batch, data_chunk_length, naive, shard,
                               self.rnn_states.reshape((T + 1) * N, -1), self.masks.reshape(-1),
                               cache=self.__dict__.setdefault("_seq_cache", {}))

    def feed_forward_generator_actor(self, advantages, actor_num_mini_batch=None, mini_batch_size=None):
        """API-compatible generator (actor_buffer.py:114-178): yields gathered device tensors in the reference's
        tuple order.  ``HAPPO.train`` does NOT go through this (it hands the index arrays to the kernels, which
        gather in place); this exists for callers that iterate minibatches themselves."""
        T, N = self.actions.shape[:2]
        B = T * N
        if mini_batch_size is None:
            sampler = minibatch_indices(B, actor_num_mini_batch)
        else:
            rand = draw_permutation(B)
            sampler = [rand[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(actor_num_mini_batch)]
        adv = None if advantages is None else _as_dev(advantages, self.device).reshape(-1, 1)
        for ind in sampler:
            i = ind.to(self.device)
            out = [self.flat("obs")[i], self.flat("rnn_states").reshape(B, self.recurrent_n, -1)[i], self.flat("actions")[i],
                   self.flat("masks")[i], self.flat("active_masks")[i], self.flat("action_log_probs")[i],
                   None if adv is None else adv[i],
                   None if self.available_actions is None else self.flat("available_actions")[i]]
            if self.factor is not None:
                out.append(self.factor.reshape(B, -1)[i])
            yield tuple(out)


    def _recurrent_api_generator(self, advantages, num_mini_batch, data_chunk_length, naive):
        T, N = self.actions.shape[:2]
        B = T * N
        adv = None if advantages is None else _as_dev(advantages, self.device).reshape(B, 1)
        for first, L in recurrent_first_rows(T, N, num_mini_batch, data_chunk_length, naive):
            first = first.to(self.device)
            rows = (first[None, :] + torch.arange(L, device=self.device)[:, None] * N).reshape(-1)  # l-major, like _flatten
            f = lambda name: self.flat(name)[rows]  # noqa: E731
            out = [f("obs"), self.rnn_states.reshape((T + 1) * N, self.recurrent_n, -1)[first], f("actions"), f("masks"),
                   f("active_masks"), f("action_log_probs"), None if adv is None else adv[rows],
                   None if self.available_actions is None else f("available_actions")]
            if self.factor is not None:
                out.append(self.factor.reshape(B, -1)[rows])
            yield tuple(out)

    def naive_recurrent_generator_actor(self, advantages, actor_num_mini_batch):
        """API-compatible generator (actor_buffer.py:180-221): whole columns, full-length sequences, rnn_states[0]."""
        return self._recurrent_api_generator(advantages, actor_num_mini_batch, 0, True)

    def recurrent_generator_actor(self, advantages, actor_num_mini_batch, data_chunk_length):
        """API-compatible generator (actor_buffer.py:223-326): chunks of ``data_chunk_length`` steps, rows l-major."""
        return self._recurrent_api_generator(advantages, actor_num_mini_batch, data_chunk_length, False)


class OnPolicyCriticBufferEP:
    def __init__(self, args: dict, share_obs_space, device=None):
        self.device = torch.device(device) if device is not None else _lib.default_device()
        _lib.require_gpu(self.device)
        self.episode_length = T = args["episode_length"]
        self.n_rollout_threads = N = args["n_rollout_threads"]
        self.hidden_sizes = args["hidden_sizes"]
        self.rnn_hidden_size = self.hidden_sizes[-1]
        self.recurrent_n = args["recurrent_n"]
        self.gamma = args["gamma"]
        self.gae_lambda = args["gae_lambda"]
        self.use_gae = args["use_gae"]
        self.use_proper_time_limits = args["use_proper_time_limits"]
        so = _obs_shape(share_obs_space)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)  # noqa: E731
        self.share_obs = z(T + 1, N, *so)
        self.rnn_states_critic = z(T + 1, N, self.recurrent_n, self.rnn_hidden_size)
        self.value_preds = z(T + 1, N, 1)
        self.returns = z(T + 1, N, 1)
        self.rewards = z(T, N, 1)
        self.masks = torch.ones(T + 1, N, 1, dtype=torch.float32, device=self.device)
        self.bad_masks = torch.ones_like(self.masks)
        self.advantages = z(T, N, 1)  # filled by compute_returns (fused on_policy_ha_runner.py:26-33)
        self.step = 0

    def insert(self, share_obs, rnn_states_critic, value_preds, rewards, masks, bad_masks):
        s = self.step
        self.share_obs[s + 1].copy_(_as_dev(share_obs, self.device))
        self.rnn_states_critic[s + 1].copy_(_as_dev(rnn_states_critic, self.device))
        self.value_preds[s].copy_(_as_dev(value_preds, self.device))
        self.rewards[s].copy_(_as_dev(rewards, self.device))
        self.masks[s + 1].copy_(_as_dev(masks, self.device))
        self.bad_masks[s + 1].copy_(_as_dev(bad_masks, self.device))
        self.step = (s + 1) % self.episode_length

    def after_update(self):
        self.share_obs[0].copy_(self.share_obs[-1])
        self.rnn_states_critic[0].copy_(self.rnn_states_critic[-1])
        self.masks[0].copy_(self.masks[-1])
        self.bad_masks[0].copy_(self.bad_masks[-1])

    def get_mean_rewards(self):
        return float(self.rewards.mean().item())

    def compute_returns(self, next_value, value_normalizer: Optional[ValueNorm] = None):
        """GAE / discounted returns (critic_buffer_ep.py:97-200, all 8 branches) as ONE reverse-scan kernel,
        bit-identical to the reference's NumPy loop; also writes ``self.advantages = returns[:-1] -
        denormalize(value_preds[:-1])`` (on_policy_ha_runner.py:26-33) in the same pass."""
        T, N = self.episode_length, self.n_rollout_threads
        nv = _as_dev(next_value, self.device).reshape(N)
        vn = None if value_normalizer is None else value_normalizer.stats
        gamma32 = float(np.float32(self.gamma))
        gl32 = float(np.float32(self.gamma * self.gae_lambda))  # python-double product, then cast
        call("harl_gae_returns", ptr(self.rewards), ptr(self.value_preds), ptr(self.masks), ptr(self.bad_masks), ptr(nv),
             ptr(vn), ptr(self.returns), ptr(self.advantages), T, N, gamma32, gl32, int(self.use_gae),
             int(self.use_proper_time_limits), 0, stream(), tag="gae_returns")

    def flat(self, name: str) -> torch.Tensor:
        t = getattr(self, name)
        if name in ("share_obs", "masks", "bad_masks", "value_preds", "returns", "rnn_states_critic"):
            t = t[:-1]
        return t.reshape(self.episode_length * self.n_rollout_threads, -1)

    def recurrent_batches(self, num_mini_batch: int, data_chunk_length: int, naive: bool = False, shard=None):
        """See OnPolicyActorBuffer.recurrent_batches (critic_buffer_ep.py:252-369).  FP buffers
        (critic_buffer_fp.py:262-390): the same samplers over N*A columns -- ``_ma_cast`` orders (thread, agent) pairs
        as column c = n*A + a, which is exactly the row order of the [T, N, A, .] flattening used here."""
        T, N = self.rewards.shape[:2]
        agents = getattr(self, "num_agents", None) or 1
        return _recurrent_seqs(self.device, T, N, agents, self.rnn_hidden_size * self.recurrent_n, num_mini_batch, data_chunk_length, naive,
                               shard, self.rnn_states_critic.reshape((T + 1) * N * agents, -1), self.masks.reshape(-1),
                               cache=self.__dict__.setdefault("_seq_cache", {}))

    def _recurrent_api_generator(self, num_mini_batch, data_chunk_length, naive):
        T, N = self.rewards.shape[:2]
        ncol = N * (getattr(self, "num_agents", None) or 1)
        for first, L in recurrent_first_rows(T, ncol, num_mini_batch, data_chunk_length, naive):
            first = first.to(self.device)
            rows = (first[None, :] + torch.arange(L, device=self.device)[:, None] * ncol).reshape(-1)
            f = lambda name: self.flat(name)[rows]  # noqa: E731
            yield (f("share_obs"), self.rnn_states_critic.reshape((T + 1) * ncol, self.recurrent_n, -1)[first],
                   f("value_preds"), f("returns"), f("masks"))

    def naive_recurrent_generator_critic(self, critic_num_mini_batch):
        """API-compatible generator (critic_buffer_ep.py:252-283, critic_buffer_fp.py:262-304)."""
        return self._recurrent_api_generator(critic_num_mini_batch, 0, True)

    def recurrent_generator_critic(self, critic_num_mini_batch, data_chunk_length):
        """API-compatible generator (critic_buffer_ep.py:285-369, critic_buffer_fp.py:306-390)."""
        return self._recurrent_api_generator(critic_num_mini_batch, data_chunk_length, False)

    def feed_forward_generator_critic(self, critic_num_mini_batch=None, mini_batch_size=None):
        """API-compatible generator (critic_buffer_ep.py:202-250); ``VCritic.train`` uses index arrays instead."""
        T, N = self.rewards.shape[:2]
        B = T * N
        if mini_batch_size is None:
            sampler = minibatch_indices(B, critic_num_mini_batch)
        else:
            rand = draw_permutation(B)
            sampler = [rand[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(critic_num_mini_batch)]
        for ind in sampler:
            i = ind.to(self.device)
            yield (self.flat("share_obs")[i], self.flat("rnn_states_critic").reshape(B, self.recurrent_n, -1)[i],
                   self.flat("value_preds")[i], self.flat("returns")[i], self.flat("masks")[i])


class OnPolicyCriticBufferFP(OnPolicyCriticBufferEP):
    """Critic buffer for the Feature-Pruned (per-agent) state type: every array carries an agent axis
    ``[T(+1), N, A, .]`` (reference: harl/common/buffers/on_policy_critic_buffer_fp.py:10-260).  The GAE scan runs over
    N*A columns; the flattened batch is T*N*A rows with row = (t*N + n)*A + a, as in the reference generators."""

    def __init__(self, args: dict, share_obs_space, num_agents: int, device=None):
        self.device = torch.device(device) if device is not None else _lib.default_device()
        _lib.require_gpu(self.device)
        self.episode_length = T = args["episode_length"]
        self.n_rollout_threads = N = args["n_rollout_threads"]
        self.num_agents = A = num_agents
        self.hidden_sizes = args["hidden_sizes"]
        self.rnn_hidden_size = self.hidden_sizes[-1]
        self.recurrent_n = args["recurrent_n"]
        self.gamma = args["gamma"]
        self.gae_lambda = args["gae_lambda"]
        self.use_gae = args["use_gae"]
        self.use_proper_time_limits = args["use_proper_time_limits"]
        so = _obs_shape(share_obs_space)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)  # noqa: E731
        self.share_obs = z(T + 1, N, A, *so)
        self.rnn_states_critic = z(T + 1, N, A, self.recurrent_n, self.rnn_hidden_size)
        self.value_preds = z(T + 1, N, A, 1)
        self.returns = z(T + 1, N, A, 1)
        self.rewards = z(T, N, A, 1)
        self.masks = torch.ones(T + 1, N, A, 1, dtype=torch.float32, device=self.device)
        self.bad_masks = torch.ones_like(self.masks)
        self.advantages = z(T, N, A, 1)
        self.step = 0

    def compute_returns(self, next_value, value_normalizer: Optional[ValueNorm] = None):
        """Same scan over N*A columns; the ValueNorm + proper-time-limits GAE branch multiplies in the FP buffer's
        order ``gamma*lambda*gae*mask`` (on_policy_critic_buffer_fp.py:130)."""
        T, cols = self.episode_length, self.n_rollout_threads * self.num_agents
        nv = _as_dev(next_value, self.device).reshape(cols)
        vn = None if value_normalizer is None else value_normalizer.stats
        fp_order = int(self.use_gae and self.use_proper_time_limits and value_normalizer is not None)
        call("harl_gae_returns", ptr(self.rewards), ptr(self.value_preds), ptr(self.masks), ptr(self.bad_masks), ptr(nv),
             ptr(vn), ptr(self.returns), ptr(self.advantages), T, cols, float(np.float32(self.gamma)),
             float(np.float32(self.gamma * self.gae_lambda)), int(self.use_gae), int(self.use_proper_time_limits), fp_order,
             stream(), tag="gae_returns")

    def flat(self, name: str) -> torch.Tensor:
        t = getattr(self, name)
        if name in ("share_obs", "masks", "bad_masks", "value_preds", "returns", "rnn_states_critic"):
            t = t[:-1]
        return t.reshape(self.episode_length * self.n_rollout_threads * self.num_agents, -1)

    def feed_forward_generator_critic(self, critic_num_mini_batch=None, mini_batch_size=None):
        B = self.episode_length * self.n_rollout_threads * self.num_agents
        if mini_batch_size is None:
            sampler = minibatch_indices(B, critic_num_mini_batch)
        else:
            rand = draw_permutation(B)
            sampler = [rand[i * mini_batch_size:(i + 1) * mini_batch_size] for i in range(critic_num_mini_batch)]
        for ind in sampler:
            i = ind.to(self.device)
            yield (self.flat("share_obs")[i], self.flat("rnn_states_critic").reshape(B, self.recurrent_n, -1)[i],
                   self.flat("value_preds")[i], self.flat("returns")[i], self.flat("masks")[i])
