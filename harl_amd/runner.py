"""OnPolicyHARunner: the sequential-update training step on MI355X.

Reference: harl/runners/on_policy_ha_runner.py:11-130 (train), harl/runners/on_policy_base_runner.py:462-497
(compute / after_update), :712-763 (prep_*, save / restore).  Environment stepping (collect/insert/eval/render) is
the reference's host-side rollout loop and stays there; ``from_spaces`` builds the update-side
objects directly from the spaces, which is also what ``bench.py`` and the tests use.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import _lib
from .buffers import OnPolicyActorBuffer, OnPolicyCriticBufferEP, OnPolicyCriticBufferFP, rng_sync
from ._lib import call, ptr, stream
from .dist import Comm, shard_columns
from .happo import HAA2C, HAPPO
from .hatrpo import HATRPO
from .mappo import MAPPO
from .v_critic import VCritic
from .valuenorm import ValueNorm

ALGO_REGISTRY = {"happo": HAPPO, "hatrpo": HATRPO, "haa2c": HAA2C, "mappo": MAPPO}


class OnPolicyHARunner:
    # feed-forward, one-minibatch updates: the critic's chain gets a stream of its own up to this many rows per minibatch (train())
    CRITIC_STREAM_MAX_ROWS = 150_000

    def __init__(self, args: dict, algo_args: dict, env_args: Optional[dict] = None, *, obs_spaces=None,
                 share_obs_space=None, act_spaces=None, device: Optional[torch.device] = None,
                 comm: Optional[Comm] = None, envs=None, logger=None, eval_envs=None, save_dir: Optional[str] = None):
        """``args``/``algo_args`` as in examples/train.py:87-91.  Spaces must be given explicitly or through ``envs``
        (a vectorised environment with the reference's ``ShareVecEnv`` surface: ``observation_space``,
        ``share_observation_space``, ``action_space`` lists and ``reset()/step()``, envs/env_wrappers.py); this class
        does not create environments.  With an initialised process group, ``algo_args['train']['n_rollout_threads']`` is
        the GLOBAL thread count and this rank keeps its contiguous column shard."""
        self.envs, self.logger, self.eval_envs, self.save_dir = envs, logger, eval_envs, save_dir
        if envs is not None and obs_spaces is None:
            obs_spaces, act_spaces = list(envs.observation_space), list(envs.action_space)
            share_obs_space = envs.share_observation_space[0]
        if obs_spaces is None or share_obs_space is None or act_spaces is None:
            raise ValueError("OnPolicyHARunner needs obs_spaces / share_obs_space / act_spaces (no env creation here)")
        self.args = args
        self.algo_args = algo_args
        self.env_args = env_args
        self.device = torch.device(device if device is not None else "cuda:0")
        _lib.require_gpu(self.device)
        self.comm = comm if comm is not None else Comm()
        algo = args.get("algo", "happo")
        if algo not in ALGO_REGISTRY:
            raise NotImplementedError(f"algo {algo}: this round implements {sorted(ALGO_REGISTRY)}")
        self.num_agents = len(obs_spaces)
        self.state_type = (env_args or {}).get("state_type", "EP")
        if self.state_type not in ("EP", "FP"):
            raise ValueError(f"state_type {self.state_type}")
        self.share_param = bool(algo_args["algo"].get("share_param", False))
        if self.share_param and algo != "mappo":
            raise NotImplementedError("share_param is a MAPPO feature (HAPPO/HATRPO/HAA2C need per-agent actors)")
        self.fixed_order = algo_args["algo"].get("fixed_order", True)
        self.action_aggregation = algo_args["algo"]["action_aggregation"]
        if self.share_param and self.comm.world_size > 1 and (algo_args["model"].get("use_recurrent_policy")
                                                             or algo_args["model"].get("use_naive_recurrent_policy")):
            # refused before any rollout instead of in the first train() (configs.unsupported_reason has the why)
            raise NotImplementedError("share_param with recurrent policies under data parallelism is not implemented")

        n_global = algo_args["train"]["n_rollout_threads"]
        lo, hi = shard_columns(n_global, self.comm.rank, self.comm.world_size)
        self.n_global, self.col_lo, self.col_hi = n_global, lo, hi
        train_local = dict(algo_args["train"])
        train_local["n_rollout_threads"] = hi - lo
        margs = {**algo_args["model"], **algo_args["algo"]}
        self.actor: List[HAPPO] = []
        for a in range(self.num_agents):  # construction order = reference (actors 0..A-1, then critic) for RNG parity
            if self.share_param and a > 0:  # ONE actor object in every slot (on_policy_base_runner.py:96-113)
                self.actor.append(self.actor[0])
            else:
                self.actor.append(ALGO_REGISTRY[algo](margs, obs_spaces[a], act_spaces[a], device=self.device))
        self.actor_buffer = [OnPolicyActorBuffer({**train_local, **algo_args["model"]}, obs_spaces[a], act_spaces[a],
                                                 device=self.device) for a in range(self.num_agents)]
        self.critic = VCritic(margs, share_obs_space, device=self.device)
        cb_args = {**train_local, **algo_args["model"], **algo_args["algo"]}
        if self.state_type == "EP":
            self.critic_buffer = OnPolicyCriticBufferEP(cb_args, share_obs_space, device=self.device)
        else:
            self.critic_buffer = OnPolicyCriticBufferFP(cb_args, share_obs_space, self.num_agents, device=self.device)
        self.value_normalizer = ValueNorm(1, device=self.device) if algo_args["train"]["use_valuenorm"] else None
        self._critic_comm = self.comm.second_group()  # collective: every rank passes here, in constructor order
        self._init_update_state(n_global)
        if algo_args["train"].get("model_dir") is not None:  # on_policy_base_runner.py:168-169
            self.restore(algo_args["train"]["model_dir"])

    def _init_update_state(self, n_global: Optional[int] = None, comm: Optional[Comm] = None) -> None:
        """Everything train() / compute() need beyond the reference runner's own attributes (the drop-in subclasses of the
        reference's OnPolicyBaseRunner get it lazily through ``_ensure_update_state``): the communicator, this rank's column
        shard, and the pinned / device scratch of train()."""
        if not hasattr(self, "comm") or self.comm is None:
            self.comm = comm if comm is not None else Comm()
        if n_global is None:
            n_global = self.algo_args["train"]["n_rollout_threads"]
        if not hasattr(self, "col_lo"):
            self.n_global = n_global
            self.col_lo, self.col_hi = shard_columns(n_global, self.comm.rank, self.comm.world_size)
        shard = (self.n_global, self.col_lo, self.col_hi) if self.comm.enabled else None
        for x in list(self.actor) + [self.critic]:
            x.comm, x.shard = self.comm, shard
        # HARL_CRITIC_GROUP=1: the critic's collectives go through a communicator of their own and its update keeps its own
        # stream under data parallelism.  The group is created by the constructors (collective call: never from this lazy
        # path, where ranks may arrive at different times); without one the critic shares the actors' communicator + stream.
        self.critic.comm = getattr(self, "_critic_comm", None) or self.comm
        self._logp_old = None
        self._counts_host = None
        self._update_state_ready = True

    def _ensure_update_state(self) -> None:
        if not getattr(self, "_update_state_ready", False):
            self._init_update_state()

    def _check_comms(self) -> None:
        """After train()'s read-back (the device is idle): a timed-out one-shot exchange has put NaNs into that step's
        gradients -- raise instead of training on.  No-op unless HARL_ALLREDUCE selected the one-shot path."""
        self.comm.check()
        if self.critic.comm is not self.comm:
            self.critic.comm.check()

    # ---- on_policy_base_runner.py:462-484 -------------------------------------------------------
    @torch.no_grad()
    def compute(self):
        self._ensure_update_state()
        cb = self.critic_buffer
        if self.state_type == "EP":
            next_value, _ = self.critic.get_values(cb.share_obs[-1], cb.rnn_states_critic[-1], cb.masks[-1])
        else:  # FP: all (thread, agent) rows in one batch (np.concatenate over threads, base_runner.py:472-481)
            so = cb.share_obs[-1]
            rows = so.shape[0] * so.shape[1]
            next_value, _ = self.critic.get_values(so.reshape(rows, -1), cb.rnn_states_critic[-1].reshape(rows, 1, -1),
                                                   cb.masks[-1].reshape(rows, 1))
        cb.compute_returns(next_value, self.value_normalizer)

    # ---- on_policy_ha_runner.py:11-130 ------------------------------------------------------------
    @torch.no_grad()
    def train(self):
        self._ensure_update_state()
        T = self.algo_args["train"]["episode_length"]
        N = self.col_hi - self.col_lo
        B = T * N
        dev = self.device
        actor_train_infos = []
        for x in self.actor:  # cached normalised-input images never outlive one update (nets.invalidate_caches)
            x.actor.invalidate_caches()
        factor = torch.ones(T, N, 1, dtype=torch.float32, device=dev)
        advantages = self.critic_buffer.advantages  # returns[:-1] - denormalize(value_preds[:-1]), fused into the GAE scan
        if self.state_type == "FP":
            advantages = self._fp_normalised_advantages(advantages)
        if self.fixed_order:
            agent_order = list(range(self.num_agents))
        else:
            rng_sync()
            agent_order = [int(a) for a in torch.randperm(self.num_agents).numpy()]  # first CPU-RNG draw of train()
        # Host syncs per train(): ONE read-back of every agent's active-entry count up front (early-out test,
        # happo.py:119-120) and ONE of all training statistics at the end; nothing in between waits for the GPU.
        fast = [hasattr(a, "masked_moments") for a in self.actor]
        mom_all = torch.zeros(self.num_agents, 3, dtype=torch.float64, device=dev)
        for a in range(self.num_agents):
            if fast[a]:
                adv_a = advantages if self.state_type == "EP" else advantages[:, :, a].contiguous()
                self.actor[a].masked_moments(self.actor_buffer[a], adv_a, mom_all[a])
        self.comm.all_reduce_sum(mom_all)
        # The only mid-train host read: the active-entry counts (early-out test).  It is an asynchronous copy into pinned
        # memory followed by an event; when the critic update may run first (it is independent of the actors, and with one
        # full-buffer minibatch everywhere no permutation is ever materialised, so the CPU generator only advances by
        # counts that do not depend on the order) its kernels are enqueued BEFORE the host waits on that event: the GPU
        # works through them while the host queues the first actor's launches instead of idling behind a drained stream.
        if self._counts_host is None or self._counts_host.numel() != self.num_agents:
            self._counts_host = torch.empty(self.num_agents, dtype=torch.float64, pin_memory=torch.cuda.is_available())
        self._counts_host.copy_(mom_all[:, 2], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        cinfo, critic_done = None, None
        if self._critic_first_ok(fast):
            # ... and on a stream of its own (single-GPU runs): the critic's twenty-odd launches are independent of the actors',
            # and every one of these persistent kernels ends with a tail in which most CUs have run out of slabs (kernel time
            # 0.19-0.26 ms against 0.14-0.20 ms for a wave's own slab loop, tools/phase_cycles.py) -- the other chain's next
            # kernel fills it.  HARL_CRITIC_STREAM=0 keeps one stream.
            # Under data parallelism the second stream needs the critic's own communicator (two chains issuing into ONE
            # communicator would have to interleave their collectives identically on every rank).
            # Round 6, session 3: ... and only where the launches are short.  From ~150 000 rows per minibatch every heavy kernel
            # of either chain fills the chip for 0.2 - 0.5 ms (one workgroup per CU, LDS-bound occupancy): the other chain's light
            # kernels then wait for a whole heavy launch to drain, the sum of the kernel times is the update either way, and
            # one stream measured as fast or faster (16.06 against 16.18 ms at 4096 threads, 8.90 / 9.00 at 2048, 5.45 / 5.58 at
            # 1024 -- and 3.90 against 3.73 at 512 threads, where the second stream stays; profiles/r06s3_critic_stream_ab.md).
            # HARL_CRITIC_STREAM=1 forces the second stream, 0 forbids it.
            cs_mode = os.environ.get("HARL_CRITIC_STREAM", "auto")
            if (dev.type == "cuda" and (not self.comm.enabled or self.critic.comm is not self.comm)
                    and cs_mode != "0" and (cs_mode == "1" or B <= self.CRITIC_STREAM_MAX_ROWS)):
                if getattr(self, "_critic_stream", None) is None:
                    self._critic_stream = torch.cuda.Stream(device=dev)
                main_s = torch.cuda.current_stream(dev)
                self._critic_stream.wait_stream(main_s)
                with torch.cuda.stream(self._critic_stream):
                    cinfo = self.critic.train(self.critic_buffer, self.value_normalizer, _defer=True)
                    critic_done = torch.cuda.Event()
                    critic_done.record(self._critic_stream)
            else:
                cinfo = self.critic.train(self.critic_buffer, self.value_normalizer, _defer=True)
        ev.synchronize()
        counts = self._counts_host.tolist()
        # Round 6: the critic's update NEXT TO the actors' also where permutations are materialised (recurrent policies, several
        # minibatches).  In the reference the critic's samplers draw from the global CPU generator AFTER every actor's, so the
        # generator is fast-forwarded over the actors' draws first (their sizes are known: HAPPO.rng_footprint; agents without an
        # active entry draw nothing, happo.py:119-120), the critic's train() -- on a stream of its own -- takes its permutations
        # from exactly the state the reference's critic finds, and the generator goes back to where the actors start; at the end
        # it is set to the state behind the critic's draws.  Every permutation and the final state are bit-identical to the
        # in-order run.  What it buys: with GRU policies the main stream idles ~0.4 ms per agent behind the previous agent's
        # post-update log-prob pass (a full-length chain on 6 % of the chip: 3.9 of the SMAC update's 29.5 ms, rocprofv3 trace of
        # round 6) and then ran the critic's 2.4 ms at the very end; now the critic's launches fill those gaps.
        rng_after = None
        if (cinfo is None and dev.type == "cuda" and all(hasattr(a, "rng_footprint") for a in self.actor)
                and (not self.comm.enabled or self.critic.comm is not self.comm)
                and os.environ.get("HARL_CRITIC_STREAM", "1") != "0" and os.environ.get("HARL_CRITIC_EARLY", "1") != "0"):
            from .buffers import consume_randperm
            rng_sync()
            rng_start = torch.get_rng_state()
            for a in agent_order:
                if counts[a] > 0.0:
                    for n in self.actor[a].rng_footprint(self.actor_buffer[a]):
                        consume_randperm(n)  # (the generator advance of torch.randperm(n), no permutation materialised)
            rng_sync()
            if getattr(self, "_critic_stream", None) is None:
                self._critic_stream = torch.cuda.Stream(device=dev)
            self._critic_stream.wait_stream(torch.cuda.current_stream(dev))
            # ... one EPOCH at a time (VCritic._train_epochs), epoch k enqueued right behind agent k's update and ordered behind
            # its event: that is where the main stream goes idle (agent k + 1's loss waits for agent k's post-update pass); run
            # all at once up front the critic only competed with the first agent's kernels (27.7 against 28.3 ms, SMAC 3s5z).
            # The generator keeps TWO positions: the actors' (restored after every critic epoch) and the critic's.
            staggered = dict(gen=self.critic._train_epochs(self.critic_buffer, self.value_normalizer), state=torch.get_rng_state(),
                             left=self.critic.critic_epoch)
            torch.set_rng_state(rng_start)

            def critic_epoch(after_event=None, _st=staggered):
                if _st["left"] <= 0:
                    return
                rng_sync()
                actor_state = torch.get_rng_state()
                torch.set_rng_state(_st["state"])
                if after_event is not None:
                    self._critic_stream.wait_event(after_event)
                with torch.cuda.stream(self._critic_stream):
                    next(_st["gen"])
                _st["left"] -= 1
                rng_sync()
                _st["state"] = torch.get_rng_state()
                torch.set_rng_state(actor_state)
            rng_after = staggered
        # Pre-update log-probs that cannot come out of the first epoch's forward (recurrent policies: full-length unroll from
        # rnn_states[0]; several minibatches) depend only on the agent's OWN pre-update weights, not on the agents before it
        # (on_policy_ha_runner.py:66-83) -- so all of them are enqueued up front on a side stream, where the long, narrow
        # recurrences (N/32 dependent chains of T steps) run next to the update kernels of the agents in front instead of in
        # series with them; agent k's update waits for event k.
        old_all, old_ev = {}, {}
        side_agents = [a for a in agent_order
                       if not (fast[a] and self.actor[a].fuses_old_logp() and counts[a] > 0.0) and os.environ.get("HARL_SIDE_STREAM", "1") != "0"]
        def enqueue_old_logp(a):
            """Agent a's pre-update pass on the side stream (+ the event its update waits for)."""
            act_a, buf_a = self.actor[a], self.actor_buffer[a]
            with torch.cuda.stream(self._side_stream):
                act_a.actor.fold()
                kw_a = dict(rnn_states=buf_a.rnn_states[0], masks=buf_a.flat("masks")) if act_a.actor.recurrent else {}
                act_a._logp_pass(buf_a.flat("obs"), buf_a.flat("actions"),
                                 None if buf_a.available_actions is None else buf_a.flat("available_actions"), B, old_all[a], **kw_a)
                old_ev[a] = torch.cuda.Event()
                old_ev[a].record(self._side_stream)

        # ... ONE AGENT AHEAD of the main stream (round 6, second session): enqueueing all of them up front is ~0.3 ms of host time
        # each -- the raw rocprofv3 trace of the 8-agent recurrent update showed the main queue EMPTY for the first 2.6 of its 22.7
        # ms while the host was still feeding the side stream.  The first agent's pass goes out here, agent k + 1's right behind
        # the launches of agent k's update (2.5 ms of queued GPU work for a 0.47 ms pass): same kernels, same operands per agent.
        # HARL_SIDE_AHEAD=0: all up front, as before.
        side_lazy = os.environ.get("HARL_SIDE_AHEAD", "1") != "0"
        side_pending = []
        if side_agents:
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            self._side_stream.wait_stream(main)
            for a in side_agents:
                act_a, buf_a = self.actor[a], self.actor_buffer[a]
                old_all[a] = torch.empty(B, act_a.actor.act_w, dtype=torch.float32, device=dev)
            side_pending = list(side_agents)
            for a in (side_pending[:1] if side_lazy else list(side_pending)):
                enqueue_old_logp(a)
                side_pending.remove(a)
        pending = []
        # Post-update log-probs of RECURRENT policies (full-length unroll: N/32 dependent chains of T steps, ~0.5 ms on 32 waves
        # of the chip at the SMAC sizes) go to a stream of their own: the next agent's first forward -- sequence tables, input
        # image, MLP layers, the training GRU forward -- needs nothing from them; only its first LOSS launch consumes the factor
        # and waits for the event (HAPPO._await_factor).  Same kernels, same operands, same order per tensor: bit-identical
        # results.  HARL_POST_STREAM=0 keeps the pass on the main stream.
        post_ok = dev.type == "cuda" and os.environ.get("HARL_POST_STREAM", "1") != "0"
        factor_ev, keep_alive, prev_overlapped = None, [], False
        for pos, agent_id in enumerate(agent_order):
            buf, actor = self.actor_buffer[agent_id], self.actor[agent_id]
            if agent_id in old_ev:  # this agent's networks / workspaces are in use on the side stream until then
                torch.cuda.current_stream(dev).wait_event(old_ev[agent_id])
            buf.update_factor(factor)
            if prev_overlapped:  # the post stream still reads the previous agent's pre-update log-probs: do not write into them
                keep_alive.append(self._logp_old)
                self._logp_old = None
            obs, actions = buf.flat("obs"), buf.flat("actions")
            avail = None if buf.available_actions is None else buf.flat("available_actions")
            if self._logp_old is None or self._logp_old.shape != (B, actor.actor.act_w):
                self._logp_old = torch.empty(B, actor.actor.act_w, dtype=torch.float32, device=dev)
            actor.actor.fold()
            rnn_kw = dict(rnn_states=buf.rnn_states[0], masks=buf.flat("masks")) if actor.actor.recurrent else {}
            adv_a = advantages if self.state_type == "EP" else advantages[:, :, agent_id].contiguous()
            kw = dict(_pre=(mom_all[agent_id], counts[agent_id]), _defer=True) if fast[agent_id] else {}
            # pre-update log-probs (:66-83).  With one full-buffer minibatch the first epoch's forward inside train()
            # computes exactly these (same rows, same parameters), so they are taken from there.
            fused_old = fast[agent_id] and actor.fuses_old_logp() and counts[agent_id] > 0.0
            if fused_old:
                kw["_old_logp_out"] = self._logp_old
            elif agent_id in old_all:
                self._logp_old = old_all.pop(agent_id)
            else:
                actor._logp_pass(obs, actions, avail, B, self._logp_old, **rnn_kw)
            if factor_ev is not None and hasattr(actor, "_factor_ready"):
                actor._factor_ready = factor_ev
            info = actor.train(buf, adv_a, self.state_type, **kw)               # :86-93
            if side_pending:  # the next agent's pre-update pass, behind this agent's queued launches (see enqueue_old_logp)
                enqueue_old_logp(side_pending.pop(0))
            if rng_after is not None and rng_after["left"] > 0:  # one critic epoch behind this agent's optimiser steps
                ev_a = torch.cuda.Event()
                ev_a.record(torch.cuda.current_stream(dev))
                critic_epoch(ev_a)
            if factor_ev is not None:  # (an update that never reached a loss launch has not waited yet)
                if hasattr(actor, "_factor_ready"):
                    actor._factor_ready = None
                torch.cuda.current_stream(dev).wait_event(factor_ev)
                factor_ev = None
            pending.append((len(actor_train_infos), info, actor._INFO_KEYS) if torch.is_tensor(info) else None)
            actor_train_infos.append(info)
            nxt = agent_order[pos + 1] if pos + 1 < len(agent_order) else None
            prev_overlapped = bool(post_ok and nxt is not None and actor.actor.recurrent and hasattr(self.actor[nxt], "_factor_ready"))
            if prev_overlapped:
                if getattr(self, "_post_stream", None) is None:
                    self._post_stream = torch.cuda.Stream(device=dev)
                ps = self._post_stream
                ps.wait_stream(torch.cuda.current_stream(dev))  # this agent's optimiser steps are complete
                keep_alive += [factor, self._logp_old]
                with torch.cuda.stream(ps):
                    new_factor = factor.clone()
                    actor._logp_pass(obs, actions, avail, B, None, old_logp=self._logp_old, factor=new_factor.reshape(B), **rnn_kw)
                    factor_ev = torch.cuda.Event()
                    factor_ev.record(ps)
                factor = new_factor
                continue
            new_factor = factor.clone()
            # post-update log-probs fused with factor *= agg(exp(new - old))   (:96-124)
            actor._logp_pass(obs, actions, avail, B, None, old_logp=self._logp_old, factor=new_factor.reshape(B), **rnn_kw)
            factor = new_factor
        if keep_alive:  # tensors of the main stream's allocator pool that the post stream has read
            torch.cuda.current_stream(dev).wait_stream(self._post_stream)
            keep_alive.clear()
        if rng_after is not None:  # epochs not handed out yet (fewer agents than critic epochs), then the critic's statistics
            while rng_after["left"] > 0:
                critic_epoch(None)
            for _ in rng_after["gen"]:  # (run the generator to its end)
                pass
            with torch.cuda.stream(self._critic_stream):
                cinfo = self.critic._train_result(True)
                critic_done = torch.cuda.Event()
                critic_done.record(self._critic_stream)
        if cinfo is None:
            cinfo = self.critic.train(self.critic_buffer, self.value_normalizer, _defer=True)
        if critic_done is not None:
            torch.cuda.current_stream(dev).wait_event(critic_done)
        rng_sync()  # the global CPU generator is exactly where the reference leaves it
        if rng_after is not None:  # ... i.e. behind the critic's draws, which were taken from their own position (see above)
            torch.set_rng_state(rng_after["state"])
        dev_infos = [p[1] for p in pending if p is not None] + [cinfo]
        flat = torch.cat([t.reshape(-1) for t in dev_infos]).cpu().tolist()  # the single end-of-train read-back
        self._check_comms()
        off = 0
        for p in pending:
            if p is not None:
                i, _, keys = p
                actor_train_infos[i] = dict(zip(keys, flat[off:off + len(keys)]))
                off += len(keys)
        critic_train_info = {"value_loss": flat[off], "critic_grad_norm": flat[off + 1]}
        return actor_train_infos, critic_train_info

    def _critic_first_ok(self, fast) -> bool:
        """May the critic update be enqueued before the actors'?  Only when nothing in train() materialises a permutation
        (one full-buffer minibatch everywhere, feed-forward nets, HAPPO-family actors): then every sampler merely advances
        the CPU generator by a count, the final generator state does not depend on the order of those advances, and the
        critic reads nothing the actors write."""
        if not all(fast) or os.environ.get("HARL_CRITIC_FIRST", "1") == "0":
            return False
        c = self.critic
        if c.critic_num_mini_batch != 1 or c.use_recurrent_policy or c.use_naive_recurrent_policy:
            return False
        return all(hasattr(a, "fuses_old_logp") and a.fuses_old_logp() for a in self.actor)

    def after_update(self):
        for b in self.actor_buffer:
            b.after_update()
        self.critic_buffer.after_update()

    def _fp_normalised_advantages(self, advantages):
        """Global masked normalisation over every agent's active entries (on_policy_ha_runner.py:36-45,
        on_policy_ma_runner.py:25-36)."""
        dev = self.device
        active = torch.stack([b.active_masks[:-1] for b in self.actor_buffer], dim=2).contiguous()  # [T,N,A,1]
        mom = torch.zeros(3, dtype=torch.float64, device=dev)
        n = advantages.numel()
        call("harl_masked_moments", ptr(advantages), ptr(active), n, ptr(mom), _lib.scratch("mm"), stream())
        self.comm.all_reduce_sum(mom)
        adv_n = torch.empty_like(advantages)
        call("harl_adv_normalize", ptr(advantages), ptr(mom), ptr(adv_n), n, stream())
        return adv_n

    # ---- rollout side (on_policy_base_runner.py:171-460): the loop is the reference's; what changes is where the data
    # lives.  Observations go up once per step, actions come down once per step; log-probs, values, hidden states, masks
    # and everything the update reads never leave the device.
    def warmup(self):
        """Reset the environments and fill slot 0 of the buffers (on_policy_base_runner.py:269-283)."""
        obs, share_obs, available_actions = self.envs.reset()
        obs = torch.as_tensor(obs, dtype=torch.float32).to(self.device)
        for a in range(self.num_agents):
            self.actor_buffer[a].obs[0].copy_(obs[:, a])
            if self.actor_buffer[a].available_actions is not None:
                av = torch.as_tensor(available_actions, dtype=torch.float32).to(self.device)
                self.actor_buffer[a].available_actions[0].copy_(av[:, a])
        so = torch.as_tensor(share_obs, dtype=torch.float32).to(self.device)
        self.critic_buffer.share_obs[0].copy_(so[:, 0] if self.state_type == "EP" else so)

    @torch.no_grad()
    def collect(self, step: int):
        """Sample actions from every actor and values from the critic at buffer slot ``step`` (:285-343).  Returns DEVICE
        tensors (values [N,1] | [N,A,1], actions [N,A,act_w], action_log_probs, rnn_states [N,A,1,H],
        rnn_states_critic [N,1,H] | [N,A,1,H])."""
        acts, logps, rnns = [], [], []
        for a in range(self.num_agents):
            b = self.actor_buffer[a]
            action, logp, rnn = self.actor[a].get_actions(
                b.obs[step], b.rnn_states[step], b.masks[step],
                b.available_actions[step] if b.available_actions is not None else None)
            acts.append(action)
            logps.append(logp)
            rnns.append(rnn.reshape(rnn.shape[0], b.recurrent_n, -1))
        actions, action_log_probs, rnn_states = torch.stack(acts, 1), torch.stack(logps, 1), torch.stack(rnns, 1)
        cb = self.critic_buffer
        if self.state_type == "EP":
            values, rnn_c = self.critic.get_values(cb.share_obs[step], cb.rnn_states_critic[step], cb.masks[step])
            rnn_c = rnn_c.reshape(values.shape[0], cb.recurrent_n, -1)
        else:
            so = cb.share_obs[step]
            N, A = so.shape[:2]
            values, rnn_c = self.critic.get_values(so.reshape(N * A, -1), cb.rnn_states_critic[step].reshape(N * A, 1, -1),
                                                   cb.masks[step].reshape(N * A, 1))
            values, rnn_c = values.reshape(N, A, 1), rnn_c.reshape(N, A, cb.recurrent_n, -1)
        return values, actions, action_log_probs, rnn_states, rnn_c

    @torch.no_grad()
    def insert(self, data):
        """Write one environment step into the buffers (:345-460): hidden states reset and masks 0 where the whole
        environment is done, active_masks 0 for agents that died, bad_masks 0 on truncation (``bad_transition``)."""
        (obs, share_obs, rewards, dones, infos, available_actions, values, actions, action_log_probs, rnn_states,
         rnn_states_critic) = data
        dev, A = self.device, self.num_agents
        up = lambda x: torch.as_tensor(x, dtype=torch.float32).to(dev)  # noqa: E731
        dones = torch.as_tensor(dones).to(dev).bool()
        N = dones.shape[0]
        dones_env = dones.all(dim=1)
        keep = (~dones_env).to(torch.float32)
        rnn_states = rnn_states * keep.view(N, 1, 1, 1)
        rnn_states_critic = rnn_states_critic * (keep.view(N, 1, 1) if self.state_type == "EP" else keep.view(N, 1, 1, 1))
        masks = keep.view(N, 1, 1).expand(N, A, 1)
        active_masks = torch.where(dones & ~dones_env.view(N, 1), 0.0, 1.0).view(N, A, 1)
        if self.state_type == "EP":
            bad = [[0.0] if info[0].get("bad_transition", False) is True else [1.0] for info in infos]
        else:
            bad = [[[0.0] if info[a].get("bad_transition", False) is True else [1.0] for a in range(A)] for info in infos]
        bad_masks = torch.tensor(bad, dtype=torch.float32, device=dev)
        obs, rewards, share_obs = up(obs), up(rewards), up(share_obs)
        avail = None if (available_actions is None or available_actions[0] is None) else up(available_actions)
        for a in range(A):
            self.actor_buffer[a].insert(obs[:, a], rnn_states[:, a], actions[:, a], action_log_probs[:, a], masks[:, a],
                                        active_masks[:, a], None if avail is None else avail[:, a])
        if self.state_type == "EP":
            self.critic_buffer.insert(share_obs[:, 0], rnn_states_critic, values, rewards[:, 0], masks[:, 0], bad_masks)
        else:
            self.critic_buffer.insert(share_obs, rnn_states_critic, values, rewards, masks, bad_masks)

    def run(self, num_episodes: Optional[int] = None):
        """The reference's training loop (:171-267) over ``self.envs``: warmup, then per episode lr decay, T x
        (collect -> envs.step -> insert), compute, train, [log], every ``eval_interval`` episodes [eval +] save,
        after_update.  Returns the per-episode (actor_train_infos, critic_train_info, mean step reward) list; ``logger``
        callbacks are invoked if one was given; checkpoints go to ``self.save_dir`` when it is set."""
        if self.envs is None:
            raise RuntimeError("run() needs a vectorised environment (envs=...)")
        tr = self.algo_args["train"]
        T = tr["episode_length"]
        episodes = num_episodes if num_episodes is not None else int(tr["num_env_steps"]) // T // tr["n_rollout_threads"]
        self.warmup()
        if self.logger is not None:
            self.logger.init(episodes)
        history = []
        for episode in range(1, episodes + 1):
            if tr.get("use_linear_lr_decay", False):
                for a in self.actor:
                    a.lr_decay(episode, episodes)
                self.critic.lr_decay(episode, episodes)
            if self.logger is not None:
                self.logger.episode_init(episode)
            self.prep_rollout()
            for step in range(T):
                values, actions, logp, rnn, rnn_c = self.collect(step)
                obs, share_obs, rewards, dones, infos, avail = self.envs.step(actions.cpu().numpy())
                data = (obs, share_obs, rewards, dones, infos, avail, values, actions, logp, rnn, rnn_c)
                if self.logger is not None:
                    self.logger.per_step(data)
                self.insert(data)
            self.compute()
            self.prep_training()
            infos_a, info_c = self.train()
            history.append((infos_a, info_c, self.critic_buffer.get_mean_rewards()))
            if self.logger is not None and episode % tr.get("log_interval", 1) == 0:
                self.logger.episode_log(infos_a, info_c, self.actor_buffer, self.critic_buffer)
            if episode % tr.get("eval_interval", 25) == 0:  # on_policy_base_runner.py:252-258
                if self.algo_args.get("eval", {}).get("use_eval", False) and self.eval_envs is not None:
                    self.prep_rollout()
                    self.eval()
                if getattr(self, "save_dir", None):
                    self.save(self.save_dir)
            self.after_update()
        return history

    @torch.no_grad()
    def eval(self):
        """Deterministic evaluation episodes on ``self.eval_envs`` (on_policy_base_runner.py:499-590): actions from
        ``actor.act(..., deterministic=True)``, hidden states reset and masks 0 where an environment finished, logger
        callbacks ``eval_init / eval_per_step / eval_thread_done / eval_log``; returns the mean episode reward."""
        import numpy as np
        ev = self.algo_args["eval"]
        n_thr, A = ev["n_eval_rollout_threads"], self.num_agents
        lg = self.logger
        if lg is not None:
            lg.eval_init()
        obs, _share, avail = self.eval_envs.reset()
        H = self.algo_args["model"]["hidden_sizes"][-1]
        rn = self.algo_args["model"]["recurrent_n"]
        rnn = torch.zeros(n_thr, A, rn, H, dtype=torch.float32, device=self.device)
        masks = torch.ones(n_thr, A, 1, dtype=torch.float32, device=self.device)
        done_eps, ep_rewards, cur = 0, [], np.zeros(n_thr)
        while True:
            acts = []
            obs_d = torch.as_tensor(np.asarray(obs), dtype=torch.float32).to(self.device)
            av_d = None if (avail is None or avail[0] is None) else torch.as_tensor(np.asarray(avail), dtype=torch.float32).to(self.device)
            for a in range(A):
                act, r_ = self.actor[a].act(obs_d[:, a], rnn[:, a], masks[:, a], None if av_d is None else av_d[:, a],
                                            deterministic=True)
                rnn[:, a] = r_.reshape(n_thr, rn, H)
                acts.append(act)
            actions = torch.stack(acts, 1).cpu().numpy()
            obs, share, rewards, dones, infos, avail = self.eval_envs.step(actions)
            if lg is not None:
                lg.eval_per_step((obs, share, rewards, dones, infos, avail))
            cur += np.mean(np.asarray(rewards), axis=1).reshape(n_thr)
            dones_env = np.all(np.asarray(dones), axis=1)
            keep = torch.as_tensor(~dones_env).to(self.device).float()
            rnn = rnn * keep.view(n_thr, 1, 1, 1)
            masks = keep.view(n_thr, 1, 1).expand(n_thr, A, 1).contiguous()
            for i in range(n_thr):
                if dones_env[i]:
                    done_eps += 1
                    ep_rewards.append(cur[i])
                    cur[i] = 0.0
                    if lg is not None:
                        lg.eval_thread_done(i)
            if done_eps >= ev["eval_episodes"]:
                if lg is not None:
                    lg.eval_log(done_eps)
                return float(np.mean(ep_rewards))

    @torch.no_grad()
    def render(self, expand_dims: bool = False, manual_render: bool = False, delay: float = 0.0) -> List[float]:
        """Roll the deterministic policy through ``algo_args['render']['render_episodes']`` episodes of ``self.envs``
        (on_policy_base_runner.py:594-710).  ``expand_dims``: the environment is a single instance without the parallel-
        environment axis (the reference's ``manual_expand_dims`` environments); ``manual_render`` calls ``envs.render()``
        after every step, ``delay`` sleeps between steps.  Returns the episode returns (the reference prints them)."""
        import time

        import numpy as np
        A = self.num_agents
        H = self.algo_args["model"]["hidden_sizes"][-1]
        rn = self.algo_args["model"]["recurrent_n"]
        returns = []
        for _ in range(self.algo_args["render"]["render_episodes"]):
            obs, _share, avail = self.envs.reset()
            n = 1 if expand_dims else len(obs)
            rnn = torch.zeros(n, A, rn, H, dtype=torch.float32, device=self.device)
            masks = torch.ones(n, A, 1, dtype=torch.float32, device=self.device)
            total = 0.0
            while True:
                obs_d = torch.as_tensor(np.asarray(obs), dtype=torch.float32).to(self.device).reshape(n, A, -1)
                no_av = avail is None or (not expand_dims and avail[0] is None)
                av_d = None if no_av else torch.as_tensor(np.asarray(avail), dtype=torch.float32).to(self.device).reshape(n, A, -1)
                acts = []
                for a in range(A):
                    act, r_ = self.actor[a].act(obs_d[:, a], rnn[:, a], masks[:, a], None if av_d is None else av_d[:, a],
                                                deterministic=True)
                    rnn[:, a] = r_.reshape(n, rn, H)
                    acts.append(act)
                actions = torch.stack(acts, 1).cpu().numpy()
                obs, _share, rewards, dones, _infos, avail = self.envs.step(actions[0] if expand_dims else actions)
                r0, d0 = np.asarray(rewards), np.asarray(dones)
                total += float(r0.reshape(-1)[0])
                if manual_render:
                    self.envs.render()
                if delay > 0.0:
                    time.sleep(delay)
                if bool(d0.reshape(-1)[0]):
                    print(f"total reward of this episode: {total}")
                    returns.append(total)
                    break
        return returns

    def close(self):
        for e in (self.envs, getattr(self, "eval_envs", None)):
            if e is not None and hasattr(e, "close"):
                e.close()

    def prep_rollout(self):
        for a in self.actor:
            a.prep_rollout()
        self.critic.prep_rollout()

    def prep_training(self):
        for a in self.actor:
            a.prep_training()
        self.critic.prep_training()

    # ---- on_policy_base_runner.py:724-763: same file names, same state_dict keys ---------------------
    def save(self, save_dir: Optional[str] = None):
        if save_dir is None:
            save_dir = self.save_dir
        save_dir = str(save_dir)
        os.makedirs(save_dir, exist_ok=True)
        for a in range(self.num_agents):
            torch.save(self.actor[a].actor.state_dict(), os.path.join(save_dir, f"actor_agent{a}.pt"))
        torch.save(self.critic.critic.state_dict(), os.path.join(save_dir, "critic_agent.pt"))
        if self.value_normalizer is not None:
            torch.save(self.value_normalizer.state_dict(), os.path.join(save_dir, "value_normalizer.pt"))

    def restore(self, model_dir: Optional[str] = None):
        if model_dir is None:
            model_dir = self.algo_args["train"]["model_dir"]
        for a in range(self.num_agents):
            self.actor[a].actor.load_state_dict(torch.load(os.path.join(model_dir, f"actor_agent{a}.pt"), map_location=self.device))
        self.critic.critic.load_state_dict(torch.load(os.path.join(model_dir, "critic_agent.pt"), map_location=self.device))
        p = os.path.join(model_dir, "value_normalizer.pt")
        if self.value_normalizer is not None and os.path.exists(p):
            self.value_normalizer.load_state_dict(torch.load(p, map_location=self.device))


class OnPolicyMARunner(OnPolicyHARunner):
    """MAPPO: simultaneous (not sequential) actor updates without the factor (runners/on_policy_ma_runner.py:10-64),
    optionally with ONE parameter-shared actor."""

    @torch.no_grad()
    def train(self):
        self._ensure_update_state()
        dev = self.device
        A = self.num_agents
        for x in self.actor:
            x.actor.invalidate_caches()
        advantages = self.critic_buffer.advantages
        if self.state_type == "FP":
            advantages = self._fp_normalised_advantages(advantages)
        mom_all = torch.zeros(A, 3, dtype=torch.float64, device=dev)
        for a in range(A):
            adv_a = advantages if self.state_type == "EP" else advantages[:, :, a].contiguous()
            self.actor[a].masked_moments(self.actor_buffer[a], adv_a, mom_all[a])
        self.comm.all_reduce_sum(mom_all)
        dev_infos = []
        if self.share_param:
            info = self.actor[0].share_param_train(self.actor_buffer, advantages, A, self.state_type,
                                                   _moments=mom_all.sum(0), _defer=True)
            rng_sync()
            torch.randperm(A)  # on_policy_ma_runner.py:42-43 iterates over a fresh permutation: one more CPU-RNG draw
            dev_infos = [info]
            slots = [0] * A
        else:
            counts = mom_all[:, 2].cpu().tolist()
            slots = []
            for a in range(A):
                buf = self.actor_buffer[a]
                buf.factor = None
                adv_a = advantages if self.state_type == "EP" else advantages[:, :, a].contiguous()
                info = self.actor[a].train(buf, adv_a, self.state_type, _pre=(mom_all[a], counts[a]), _defer=True)
                if torch.is_tensor(info):
                    slots.append(len(dev_infos))
                    dev_infos.append(info)
                else:
                    slots.append(info)  # early-out dict (no active entries)
        cinfo = self.critic.train(self.critic_buffer, self.value_normalizer, _defer=True)
        rng_sync()
        flat = torch.cat([t.reshape(-1) for t in dev_infos] + [cinfo.reshape(-1)]).cpu().tolist()
        self._check_comms()
        keys = self.actor[0]._INFO_KEYS
        actor_train_infos = [s if isinstance(s, dict) else dict(zip(keys, flat[4 * s:4 * s + 4])) for s in slots]
        off = 4 * len(dev_infos)
        return actor_train_infos, {"value_loss": flat[off], "critic_grad_norm": flat[off + 1]}


RUNNER_REGISTRY = {"happo": OnPolicyHARunner, "hatrpo": OnPolicyHARunner, "haa2c": OnPolicyHARunner,
                   "mappo": OnPolicyMARunner}
