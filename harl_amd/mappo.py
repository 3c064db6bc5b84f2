"""MAPPO on MI355X (reference: harl/algorithms/actors/mappo.py:10-234).

``update``/``train`` are HAPPO's with the sequential-update factor fixed to 1 (the reference's two files differ only in
that term), so they run on exactly the same kernels (``harl_actor_head_loss`` with ``factor = NULL``).  Parameter
sharing (``share_param_train``, mappo.py:149-234) concatenates one minibatch per agent into a single update; here every
agent's segment is pushed through forward / loss / backward in place (its own buffers, its own index array) and the
UNSCALED folded gradients and loss sums are accumulated before ONE optimiser step -- the same sums the concatenated
batch gives, without materialising it.
"""
from __future__ import annotations

from typing import List

import torch

from . import _lib
from ._lib import call, ptr, stream
from .buffers import OnPolicyActorBuffer, consume_randperm, minibatch_indices, rng_sync
from .dist import local_minibatch_rows
from .happo import HAPPO
from .valuenorm import _as_dev


class MAPPO(HAPPO):
    def update(self, sample):
        """8-tuple of mappo.py:47-56 (no factor)."""
        return super().update(tuple(sample) + (None,)) if len(sample) == 8 else super().update(sample)

    def share_param_train(self, actor_buffer: List[OnPolicyActorBuffer], advantages, num_agents: int, state_type: str,
                          _moments=None, _defer: bool = False):
        """ppo_epoch x actor_num_mini_batch updates of the ONE shared actor on all agents' data (mappo.py:149-234).
        EP: advantages [T, N, 1] are normalised with ONE mean/std over every agent's active entries (:165-183), passed to
        the loss kernel as the summed fp64 moments; FP: per-agent slices of the runner-normalised [T, N, A, 1] tensor."""
        dev, net = self.device, self.actor
        net.invalidate_caches()
        A = num_agents
        T, N = actor_buffer[0].actions.shape[:2]
        B = T * N
        adv = _as_dev(advantages, dev)
        if self.use_recurrent_policy or self.use_naive_recurrent_policy:
            return self._share_param_train_recurrent(actor_buffer, adv, A, state_type, _moments, _defer)
        moments = None
        if state_type == "EP":
            adv_a = [adv.reshape(B).contiguous()] * A
            if _moments is None:
                mom = torch.zeros(A, 3, dtype=torch.float64, device=dev)
                for a in range(A):
                    self.masked_moments(actor_buffer[a], adv_a[a], mom[a])
                self.comm.all_reduce_sum(mom)
                _moments = mom.sum(0)
            moments = _moments
        else:
            adv_a = [adv[:, :, a].reshape(B).contiguous() for a in range(A)]
        self._info.zero_()
        net.fold()
        acc = torch.zeros_like(net.dwp)
        n_global = self.shard[0] * T if self.shard else B
        k = self.actor_num_mini_batch
        s = stream()
        for _ in range(self.ppo_epoch):
            # every agent's generator draws its own permutation, in agent order, when the first minibatch is requested
            samplers = []
            for a in range(A):
                if k == 1:
                    consume_randperm(n_global)
                    samplers.append([None])
                else:
                    samplers.append(minibatch_indices(n_global, k, dev))
            for b in range(k):
                acc.zero_()
                net._ensure_ws(B)
                net.scalars.zero_()
                segs = []
                for a in range(A):
                    ind = samplers[a][b]
                    if ind is not None and self.shard:
                        ind = local_minibatch_rows(ind, self.shard[0], self.shard[1], self.shard[2])
                    segs.append(None if ind is None else ind.to(dev))
                if net.md and self.use_policy_active_masks:
                    # MultiDiscrete: sum(active) / rows of the CONCATENATED minibatch (happo._md_ent_scale; mappo.py:185-234)
                    st = torch.zeros(2, dtype=torch.float32, device=dev)
                    for a in range(A):
                        act_a = actor_buffer[a].flat("active_masks").reshape(B)
                        st[0] += (act_a if segs[a] is None else act_a[segs[a]]).sum()
                        st[1] += float(B if segs[a] is None else segs[a].numel())
                    self.comm.all_reduce_sum(st)
                    self._md_ent_override = (st[0] / st[1]).reshape(1).contiguous()
                for a in range(A):
                    buf = actor_buffer[a]
                    idx = segs[a]
                    m = B if idx is None else idx.numel()
                    nblk = self._forward_backward(
                        buf.flat("obs"), idx, m, buf.flat("actions"),
                        None if buf.available_actions is None else buf.flat("available_actions"),
                        buf.flat("action_log_probs"), adv_a[a], moments, None,
                        buf.flat("active_masks").reshape(B) if self.use_policy_active_masks else None)
                    acc.add_(net.dwp)
                    call("harl_reduce_scalars", ptr(net.part_scalars), nblk, ptr(net.scalars), s)  # accumulates
                self._md_ent_override = None
                net.dwp.copy_(acc)
                self._optimizer_step(None)
        n_upd = self.ppo_epoch * k
        if _defer:
            return self._info / n_upd
        rng_sync()
        vals = (self._info / n_upd).cpu().tolist()
        return dict(zip(self._INFO_KEYS, vals))

    def _share_param_train_recurrent(self, actor_buffer, adv, A: int, state_type: str, _moments, _defer: bool):
        """Parameter sharing with GRU policies (mappo.py:185-234).  The reference concatenates the agents' recurrent samples
        along axis 0 -- [agent 0: L*m rows | agent 1: L*m rows | ...] next to rnn_states [A*m] -- and RNNLayer.forward then
        reads that array as (T = L, N = A*m) (rnn.py:40-44: x.view(T, N, -1)), i.e. row l*(A*m) + j is "time l of sequence j":
        the rows of one agent land in ONE time step.  That is what the golden vectors record, so it is what runs here: the
        gathered samples of the buffers' generator API (same draws, in agent order, on first use of every generator) are
        concatenated and go through the gathered-sample update (HAPPO.update -> nets.build_seq with L and A*m)."""
        if self.shard:
            raise NotImplementedError("share_param with recurrent policies under data parallelism")
        dev = self.device
        T, N = actor_buffer[0].actions.shape[:2]
        B = T * N
        if state_type == "EP":  # ONE mean / std over every agent's active entries (mappo.py:165-183)
            if _moments is None:
                mom = torch.zeros(A, 3, dtype=torch.float64, device=dev)
                for a in range(A):
                    self.masked_moments(actor_buffer[a], adv.reshape(B).contiguous(), mom[a])
                _moments = mom.sum(0)
            normed = torch.empty(B, dtype=torch.float32, device=dev)
            call("harl_adv_normalize", ptr(adv.reshape(B).contiguous()), ptr(_moments.contiguous()), ptr(normed), B, stream())
            adv_a = [normed.reshape(T, N, 1)] * A
        else:
            adv_a = [adv[:, :, a].contiguous() for a in range(A)]
        self._info.zero_()
        k = self.actor_num_mini_batch
        for _ in range(self.ppo_epoch):
            gens = [(actor_buffer[a].recurrent_generator_actor(adv_a[a], k, self.data_chunk_length) if self.use_recurrent_policy
                     else actor_buffer[a].naive_recurrent_generator_actor(adv_a[a], k)) for a in range(A)]
            for _b in range(k):
                samples = [next(g) for g in gens]
                batch = tuple(None if samples[0][i] is None else torch.cat([smp[i] for smp in samples], dim=0) for i in range(8))
                self.update(batch)  # accumulates into self._info
        n_upd = self.ppo_epoch * k
        if _defer:
            return self._info / n_upd
        rng_sync()
        vals = (self._info / n_upd).cpu().tolist()
        return dict(zip(self._INFO_KEYS, vals))
