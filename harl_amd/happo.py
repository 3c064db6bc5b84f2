"""HAPPO on MI355X: same class surface as the reference (harl/algorithms/actors/on_policy_base.py:8-137,
harl/algorithms/actors/happo.py:10-158), arithmetic in libharl_hip.so.

One ``update`` = fold -> trunk forward (MFMA) -> fused head/loss/backward-to-dz_L -> weight-gradient partials
(MFMA) -> deterministic reduce + unfold -> [data-parallel all-reduce] -> fused grad-norm/clip/Adam.  Nothing
returns to the host inside ``train()`` except the int64 minibatch permutations going up and one small read-back
of the accumulated training statistics at the end.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import PS_STRIDE, call, ptr, stream
from .buffers import OnPolicyActorBuffer, consume_randperm, minibatch_indices, rng_sync
from .dist import Comm, local_minibatch_rows
from .nets import FusedAdam, StochasticPolicy, build_seq, seq_compact
from .valuenorm import _as_dev


class OnPolicyBase:
    """Actor + optimiser (reference: actors/on_policy_base.py:8-137)."""

    def __init__(self, args, obs_space, act_space, device=torch.device("cuda:0")):
        self.args = args
        self.device = torch.device(device)
        _lib.require_gpu(self.device)
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        self.data_chunk_length = args["data_chunk_length"]
        self.use_recurrent_policy = args["use_recurrent_policy"]
        self.use_naive_recurrent_policy = args["use_naive_recurrent_policy"]
        self.use_policy_active_masks = args["use_policy_active_masks"]
        self.action_aggregation = args["action_aggregation"]
        self.lr = args["lr"]
        self.opti_eps = args["opti_eps"]
        self.weight_decay = args["weight_decay"]
        self.obs_space = obs_space
        self.act_space = act_space
        self.actor = StochasticPolicy(args, obs_space, act_space, self.device)
        self.actor_optimizer = FusedAdam(self.actor, self.lr, self.opti_eps, self.weight_decay)
        self.comm = Comm()
        self.shard = None  # (n_global, lo, hi) when n_rollout_threads is sharded across ranks
        self._old_logp: Optional[torch.Tensor] = None

    def lr_decay(self, episode, episodes):  # utils/models_tools.py:77-87
        lr = self.lr - (self.lr * ((episode - 1) / float(episodes)))
        for g in self.actor_optimizer.param_groups:
            g["lr"] = lr

    # ---- log-prob passes over a whole [T*N] batch (on_policy_ha_runner.py:66-83,96-113) --------------
    def _logp_pass(self, obs, actions, avail, M, logp_out, old_logp=None, factor=None, head_out=None,
                   rnn_states=None, masks=None, h_last=False, reuse_trunk=False):
        """Forward + head over rows 0..M-1.  Recurrent nets follow the reference's convention (rnn.py:24-42): with
        rnn_states [m, 1, H], M = L*m rows are L steps of m sequences (l-major) unrolled from rnn_states with mask
        resets.  Returns the final hidden state [m, H] when ``h_last`` is set."""
        net = self.actor
        agg = int(self.action_aggregation == "mean")
        if net.md and not net.recurrent:  # MultiDiscrete: trunk -> logits of every group -> per-head log-softmax
            if not reuse_trunk:
                net.forward_trunk(obs, None, M, for_backward=False)
            net.md_logits(M)
            call("harl_md_head_logp", *net.md_layout(), M, ptr(actions), ptr(logp_out), ptr(old_logp),
                 0 if old_logp is None else old_logp.shape[1], ptr(factor), agg, ptr(head_out), 0, 0, stream(),
                 tag="md_head_logp")
            return None
        if not reuse_trunk and net.fused_update_ok(None, train=False):  # forward + head in one launch, x_hat_2 never leaves the chip
            call("harl_update_logp", *net.fused_args(obs, M), ptr(net.log_std()), net.std_x_coef, net.std_y_coef,
                 int(net.discrete), net.act_dim, ptr(actions), ptr(avail), ptr(logp_out), ptr(old_logp), ptr(factor), agg,
                 ptr(head_out), stream(), tag="update_logp")
            return None
        if not net.recurrent:
            if not reuse_trunk:  # reuse_trunk: x_hat_L of these rows under the current weights is still in the workspace
                net.forward_trunk(obs, None, M, for_backward=False)
            Wp, bp = net._packs[-1]
            call("harl_actor_head_logp", ptr(net.xh[-1]), M, net.hidden_sizes[-1], ptr(Wp), ptr(bp), ptr(net.log_std()),
                 net.std_x_coef, net.std_y_coef, int(net.discrete), net.act_dim, ptr(actions), ptr(avail), ptr(logp_out),
                 ptr(old_logp), ptr(factor), agg, ptr(head_out), 0, 0, stream(), tag="actor_head_logp")
            return None
        H = net.hidden_sizes[-1]
        m = rnn_states.shape[0]
        L = M // m
        HS = H * net.recurrent_n  # state row: the layers' hidden vectors side by side (rnn_states [m, recurrent_n, H])
        seq = build_seq(self.device, L, m, HS, h0=_as_dev(rnn_states, self.device).reshape(m, HS),
                        masks_src=_as_dev(masks, self.device), want_h_last=h_last)
        idx, Mp = seq["idx"], L * seq["m_pad"]
        padded = idx is not None
        g = (lambda t: None if t is None else t[idx].contiguous()) if padded else (lambda t: t)  # noqa: E731
        tmp = (lambda t: None if t is None else torch.empty(Mp, *t.shape[1:], **self.tpdv)) if padded else (lambda t: t)  # noqa: E731
        a_p, av_p, old_p, f_p = g(actions), g(avail), g(old_logp), g(factor)
        lo_p, ho_p = tmp(logp_out), tmp(head_out)
        net.forward_trunk(obs, idx, Mp, for_backward=False, seq=seq)
        fx, _, _, fh = net.feat()
        Wp, bp = net._packs[-1]
        if net.md:
            net.md_logits(Mp)
            call("harl_md_head_logp", *net.md_layout(), Mp, ptr(a_p), ptr(lo_p), ptr(old_p),
                 0 if old_p is None else old_p.shape[1], ptr(f_p), agg, ptr(ho_p), m, seq["m_pad"], stream(),
                 tag="md_head_logp")
        else:
            call("harl_actor_head_logp", ptr(fx), Mp, fh, ptr(Wp), ptr(bp), ptr(net.log_std()), net.std_x_coef,
                 net.std_y_coef, int(net.discrete), net.act_dim, ptr(a_p), ptr(av_p), ptr(lo_p), ptr(old_p), ptr(f_p), agg,
                 ptr(ho_p), m, seq["m_pad"], stream(), tag="actor_head_logp")
        if padded:
            for dst, src in ((logp_out, lo_p), (head_out, ho_p), (factor, f_p)):
                if dst is not None:
                    dst.copy_(seq_compact(src, seq))
        return None if not h_last else seq["h_last"][:m]

    def _dist_rows(self, head, M, *, noise=None, actions=None, logp=None, probs=None, argmax_out=None, ent_rows=None,
                   sigma_out=None):
        """harl_dist_rows on this policy's head outputs (Gaussian mean / normalised logits of one or several heads)."""
        net = self.actor
        if net.md and getattr(self, "_head_off", None) is None:
            off = [0]
            for n in net.nvec:
                off.append(off[-1] + int(n))
            self._head_off = torch.tensor(off, dtype=torch.int32, device=self.device)
        call("harl_dist_rows", ptr(head), M, net.act_dim, int(net.discrete), None if net.discrete else ptr(net.log_std()),
             net.std_x_coef, net.std_y_coef, ptr(noise), ptr(actions), ptr(self._head_off) if net.md else None,
             len(net.nvec) if net.md else 1, 1, ptr(logp), ptr(probs), ptr(argmax_out), ptr(ent_rows), ptr(sigma_out), stream(),
             tag="dist_rows")

    def evaluate_actions(self, obs, rnn_states_actor, action, masks, available_actions=None, active_masks=None):
        """(action_log_probs [B, act_w], dist_entropy 0-d, action_distribution) as the reference returns them
        (stochastic_policy.py:88-127, act.py:104-157): the entropy is the active-mask-weighted mean over rows when
        ``use_policy_active_masks`` and masks are given, the plain mean otherwise; the distribution is a
        ``torch.distributions`` object built from the head outputs (Gaussian mean / sigma, or normalised masked logits).
        Device tensors, no autograd graph (the update path differentiates inside the kernels)."""
        obs = _as_dev(obs, self.device)
        obs = obs.reshape(obs.shape[0], -1)
        action = _as_dev(action, self.device).reshape(obs.shape[0], -1)
        avail = None if available_actions is None else _as_dev(available_actions, self.device).reshape(obs.shape[0], -1)
        M = obs.shape[0]
        net = self.actor
        out = torch.empty(M, net.act_w, **self.tpdv)
        head = torch.empty(M, net.act_dim, **self.tpdv)
        net.fold()
        self._logp_pass(obs, action, avail, M, out, head_out=head, rnn_states=rnn_states_actor, masks=masks)
        # entropy rows, their (active-mask-weighted) mean and sigma behind the C ABI (harl_dist_rows / harl_masked_moments /
        # harl_moments_mean); the distribution OBJECT the reference returns is built from the kernel outputs
        ent_rows = torch.empty(M, **self.tpdv)
        sigma = None if net.discrete else torch.empty(net.act_dim, **self.tpdv)
        self._dist_rows(head, M, ent_rows=ent_rows, sigma_out=sigma)
        am = None
        if not net.md and active_masks is not None and self.use_policy_active_masks:  # (act.py:117-141: never for MultiDiscrete)
            am = _as_dev(active_masks, self.device).reshape(M).contiguous()
        mom = torch.zeros(3, dtype=torch.float64, device=self.device)
        entropy = torch.empty((), **self.tpdv)
        call("harl_masked_moments", ptr(ent_rows), ptr(am), M, ptr(mom), _lib.scratch("mm"), stream())
        call("harl_moments_mean", ptr(mom), ptr(entropy), stream())
        if net.md:
            return out, entropy, None
        dist = (torch.distributions.Categorical(logits=head) if net.discrete
                else torch.distributions.Normal(head, sigma.expand_as(head)))
        return out, entropy, dist

    @torch.no_grad()
    def get_actions(self, obs, rnn_states_actor, masks, available_actions=None, deterministic=False):
        """Rollout-side sampling (on_policy_base.py:52-69 -> StochasticPolicy.forward, act.py:45-86).  The trunk, GRU
        step and head run on the HIP kernels (head outputs via ``harl_actor_head_logp(head_out=...)``); the draw itself
        uses torch's device generator, which is what the reference's ``Normal.sample()/Categorical.sample()`` use on a
        GPU.  Returns device tensors (actions [B, act_w], action_log_probs [B, act_w], rnn_states [B, 1, H])."""
        net = self.actor
        x = _as_dev(obs, self.device)
        x = x.reshape(x.shape[0], -1)
        M = x.shape[0]
        avail = None if available_actions is None else _as_dev(available_actions, self.device).reshape(M, -1)
        net.fold()
        head = torch.empty(M, net.act_dim, **self.tpdv)
        # (always a device tensor: the reference's callers apply _t2n to it, on_policy_base_runner.py:310-312,533)
        rnn_out = None if rnn_states_actor is None else _as_dev(rnn_states_actor, self.device)
        h = self._logp_pass(x, None, avail, M, None, head_out=head, rnn_states=rnn_states_actor, masks=masks,
                            h_last=True)  # head_out only: no actions needed
        if net.recurrent:
            rnn_out = h.reshape(M, net.recurrent_n, -1).clone()
        # everything around the draw runs in harl_dist_rows; the draw itself is torch's device generator
        if net.discrete:  # head = normalised logits (masked entries ~ -1e10); MultiDiscrete: one draw per head, log-probs summed
            nh = len(net.nvec) if net.md else 1
            actions = torch.empty(M, nh, **self.tpdv)
            logp = torch.empty(M, 1, **self.tpdv)
            if deterministic:
                self._dist_rows(head, M, argmax_out=actions, logp=logp)  # (actions NULL: the log-prob of the argmax)
            else:
                probs = torch.empty_like(head)
                self._dist_rows(head, M, probs=probs)
                if net.md:
                    lo = 0
                    for k, n in enumerate(net.nvec):  # act.py:56-73, one Categorical.sample() per head, in order
                        actions[:, k:k + 1] = torch.multinomial(probs[:, lo:lo + n], 1)
                        lo += n
                else:
                    actions.copy_(torch.multinomial(probs, 1))
                self._dist_rows(head, M, actions=actions, logp=logp)
        else:
            noise = None if deterministic else torch.randn_like(head)
            actions = torch.empty_like(head)
            logp = torch.empty_like(head)
            self._dist_rows(head, M, noise=noise, actions=actions, logp=logp)
        return actions, logp, rnn_out

    @torch.no_grad()
    def act(self, obs, rnn_states_actor, masks, available_actions=None, deterministic=False):
        actions, _, rnn = self.get_actions(obs, rnn_states_actor, masks, available_actions, deterministic)
        return actions, rnn

    def prep_training(self):
        self.actor.train()
        self.actor.invalidate_caches()

    def prep_rollout(self):
        # phase boundary = cache boundary: the folded weights and the normalised-input image are re-derived on first use of
        # every rollout / training phase, so a parameter write that torch's version counter cannot see (``p.data.copy_()``,
        # a raw-pointer kernel, load through DLPack) is picked up at the next prep_rollout() / prep_training() / train()
        self.actor.eval()
        self.actor.invalidate_caches()


class HAPPO(OnPolicyBase):
    def __init__(self, args, obs_space, act_space, device=torch.device("cuda:0")):
        super().__init__(args, obs_space, act_space, device)
        self.clip_param = args["clip_param"]
        self.ppo_epoch = args["ppo_epoch"]
        self.actor_num_mini_batch = args["actor_num_mini_batch"]
        self.entropy_coef = args["entropy_coef"]
        self.use_max_grad_norm = args["use_max_grad_norm"]
        self.max_grad_norm = args["max_grad_norm"]
        self._surrogate_mode = 0  # harl_actor_head_loss `trpo` argument: 0 = clipped (HAPPO)
        self._info = torch.zeros(4, dtype=torch.float64, device=self.device)  # fp64 sums of the per-update fp32 policy_loss, dist_entropy, grad_norm, ratio (the reference sums .item() values: happo.py:145-150)
        self._grad_tap = None
        self._trace = None  # test hook: list receiving a clone of the running statistics after every optimiser step
        self._state_tap = None  # test hook: list receiving (parameters, exp_avg, exp_avg_sq, step) as they stand BEFORE every optimiser step
        # runner-internal: the event behind which this agent's sequential-update factor is complete (it is produced on the
        # runner's post-update stream while this agent's first forward already runs, runner.train); awaited in front of the
        # first loss launch -- the only consumer of the factor
        self._factor_ready = None

    def _await_factor(self) -> None:
        ev = self._factor_ready
        if ev is not None:
            self._factor_ready = None
            torch.cuda.current_stream(self.device).wait_event(ev)

    # ---- one optimiser step on rows idx[0..m) of the flat [T*N, .] tensors (happo.py:28-102) ---------
    def _forward_backward(self, obs, idx, m, actions, avail, old_logp, adv, adv_moments, factor, active, seq=None,
                          logp_out=None):
        """Forward, loss and backward of one minibatch: leaves the UNSCALED folded gradients in ``net.dwp`` and the loss
        kernel's per-block partial sums in ``net.part_scalars``.  ``seq`` (recurrent nets): the batch is L x m_pad rows in
        the GRU layout (nets.build_seq), idx = seq['idx'].  ``logp_out`` [m, act_w]: also emit log pi(a|o) under the
        pre-step parameters (by batch position).  ``factor`` None = 1 (MAPPO)."""
        net = self.actor
        s = stream()
        if net.fused_update_ok(idx, seq):  # fused forward + loss (csrc/update.hip), then the layer backward (hybrid) or harl_update_bwd
            fa = net.fused_args(obs, m)
            self._await_factor()
            call("harl_update_fwd_actor", *fa, ptr(net.log_std()), net.std_x_coef, net.std_y_coef, int(net.discrete),
                 net.act_dim, ptr(actions), ptr(avail), ptr(old_logp), ptr(adv), ptr(adv_moments), ptr(factor),
                 ptr(active), float(self.clip_param), float(self.entropy_coef), int(self.action_aggregation == "mean"),
                 self._surrogate_mode, ptr(logp_out), ptr(net.dz[0]), ptr(net.part_scalars),
                 ptr(net.part[net._part_offs[-1]:]), net.n_wg, *net.hybrid_outputs(), s, tag="update_fwd")
            net.backward_after_fused(obs, m)
            return net.n_wg
        if net.fused_last_ok(idx, seq):  # deeper networks: the last hidden layer runs inside the loss launch
            L = len(net.hidden_sizes)
            net.forward_trunk(obs, idx, m, upto=L - 1)
            (Wl, bl), (Wh, bh) = net._packs[L - 1], net._packs[-1]
            self._await_factor()
            call("harl_update_last_actor", ptr(net.xh[L - 2]), m, net.hidden_sizes[-1], ptr(Wl), ptr(bl), ptr(Wh), ptr(bh),
                 ptr(net.log_std()), net.std_x_coef, net.std_y_coef, int(net.discrete), net.act_dim, ptr(idx), ptr(actions),
                 ptr(avail), ptr(old_logp), ptr(adv), ptr(adv_moments), ptr(factor), ptr(active), float(self.clip_param),
                 float(self.entropy_coef), int(self.action_aggregation == "mean"), self._surrogate_mode, ptr(logp_out),
                 ptr(net.dz[0]), ptr(net.part_scalars), ptr(net.part[net._part_offs[-1]:]), net.n_wg, s, tag="update_last")
            net.backward_trunk(obs, idx, m, head_dw_done=True)
            return net.n_wg
        net.forward_trunk(obs, idx, m, seq=seq)
        Wp, bp = net._packs[-1]
        fx, fmask, frstd, fh = net.feat()
        mv, mp = (seq["m"], seq["m_pad"]) if seq is not None else (0, 0)
        self._await_factor()
        if net.md:  # MultiDiscrete (csrc/multihead.hip): logits GEMM, per-sample loss -> d(logits) in place, layer-kernel backward
            net.md_logits(m)
            nblk = _lib.load().harl_head_blocks(m)
            lay = net.md_layout()
            call("harl_md_head_loss", lay[0], *lay, m, ptr(idx), ptr(actions), ptr(old_logp), old_logp.shape[1], ptr(adv),
                 ptr(adv_moments), ptr(factor), ptr(active), ptr(self._md_ent_scale(idx, m, active, seq)),
                 float(self.clip_param), float(self.entropy_coef), int(self.action_aggregation == "mean"),
                 self._surrogate_mode, mv, mp, ptr(logp_out), ptr(net.part_scalars), nblk, s, tag="md_head_loss")
            net.backward_trunk(obs, idx, m, seq=seq)
            return nblk
        call("harl_actor_head_loss", ptr(fx), ptr(fmask), ptr(frstd), m, fh,
             ptr(Wp), ptr(bp), ptr(net.log_std()), net.std_x_coef, net.std_y_coef, int(net.discrete), net.act_dim,
             ptr(idx), ptr(actions), ptr(avail), ptr(old_logp), ptr(adv), ptr(adv_moments), ptr(factor), ptr(active),
             float(self.clip_param), float(self.entropy_coef), int(self.action_aggregation == "mean"), self._surrogate_mode,
             mv, mp, ptr(logp_out), ptr(net.dz[0]), ptr(net.dhead), ptr(net.part_scalars),
             None if net.wide_head else ptr(net.part[net._part_offs[-1]:]), net.n_wg, s,
             tag="actor_head_loss")  # head dW fused into this launch (heads of 33..64 actions: separate dW pass)
        net.backward_trunk(obs, idx, m, seq=seq, head_dw_done=not net.wide_head)
        return net.n_wg if not net.wide_head else _lib.load().harl_head_blocks(m)  # rows of part_scalars

    def _md_ent_scale(self, idx, m, active, seq) -> Optional[torch.Tensor]:
        """MultiDiscrete: device scalar sum(active) / rows of the (global) minibatch.  The reference's entropy bonus there is
        (1/rows) sum_rows sum_heads H while the surrogate is divided by sum(active) (act.py:126-139, happo.py:77-81); the
        optimiser kernel applies ONE scale 1 / sum(active) to the whole gradient, so the entropy part carries this ratio.
        None (= 1) without active masks."""
        if active is None:
            return None
        if getattr(self, "_md_ent_override", None) is not None:  # parameter sharing: the ratio of the concatenated batch
            return self._md_ent_override
        if seq is not None:      # L x m_pad rows, padding rows addressed through idx but not counted
            j = torch.arange(m, device=self.device)
            live = (j % seq["m_pad"]) < seq["m"]
            rows = idx[live] if idx is not None else j[live]
            st = torch.stack([active[rows].sum(), torch.tensor(float(rows.numel()), device=self.device)])
        else:
            a = active if idx is None else active[idx]
            st = torch.stack([a.sum(), torch.tensor(float(m), device=self.device)])
        self.comm.all_reduce_sum(st)
        return (st[0] / st[1]).reshape(1).contiguous()

    def _optimizer_step(self, nblk: Optional[int]):
        """[data-parallel all-reduce] + fused scalar reduce / unfold / grad-norm / clip / Adam / re-fold.  ``nblk`` None:
        ``net.scalars`` already holds the summed loss scalars (several minibatch segments were accumulated)."""
        net = self.actor
        net._ensure_ws(1)
        sc = net.scalars
        if self._state_tap is not None:  # (the full-size checks re-run every update of the oracle from exactly this state)
            o = self.actor_optimizer
            self._state_tap.append((net.flat_param.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), o.step_count))
        if nblk == 0:  # this rank holds no row of the (global) minibatch: contribute zeros to the all-reduce
            net.dwp.zero_()
            sc.zero_()
            nblk = None
        ps_kw = dict(part_scalars=net.part_scalars, n_scalar_blocks=nblk) if nblk is not None else {}
        if self.comm.enabled:  # ONE collective per optimiser step: [folded gradients | loss scalars] (dist.py)
            hilo = net.dwp_msg[net.total_dwp:]
            if nblk is not None:  # partial rows -> fp64 sums -> fixed-grid fp32 pieces behind the gradients, one launch
                call("harl_reduce_pack_scalars", ptr(net.part_scalars), nblk, ptr(sc), ptr(hilo), stream())
            else:
                call("harl_pack_scalars_hilo", ptr(sc), ptr(hilo), stream())
            self.comm.all_reduce_message(net.dwp_msg)
            ps_kw = dict(scalars_hilo=hilo)
        # loss = sum / sum(active) (happo.py:77-85): gradients are linear in 1/sum(active), applied inside the kernel
        ls_off = -1 if net.discrete else net.offsets["act.action_out.log_std"][0]
        self.actor_optimizer.step(0, 0.0, self.use_max_grad_norm, self.max_grad_norm, self._info, ls_off, net.act_dim,
                                  **ps_kw)
        if self._trace is not None:  # test hook: per-update statistics = differences of these snapshots (no host sync)
            self._trace.append(self._info.clone())
        if self._grad_tap is not None:  # test hook: scaled, pre-clip gradient of this update (host sync)
            self._grad_tap(net.flat_grad * float(1.0 / sc[1].item()), sc.clone())

    def _update_core(self, obs, idx, m, actions, avail, old_logp, adv, adv_moments, factor, active, seq=None,
                     logp_out=None):
        nblk = self._forward_backward(obs, idx, m, actions, avail, old_logp, adv, adv_moments, factor, active, seq=seq,
                                      logp_out=logp_out)
        self._optimizer_step(nblk)

    def update(self, sample):
        """API-compatible single update on an already-gathered minibatch (tuple order of happo.py:37-48).
        Returns (policy_loss, dist_entropy, actor_grad_norm, mean importance weight) as 0-d device tensors."""
        (obs, _rnn, actions, _masks, active, old_logp, adv, avail, factor) = sample
        dev = self.device
        obs = _as_dev(obs, dev)
        m = obs.shape[0]
        before = self._info.clone()
        self.actor.fold()
        if self.actor.recurrent:  # gathered [L*m, .] l-major minibatch + rnn_states [m, 1, H] (recurrent generators)
            HS = self.actor.hidden_sizes[-1] * self.actor.recurrent_n
            nseq = _as_dev(_rnn, dev).shape[0]
            seq = build_seq(dev, m // nseq, nseq, HS, h0=_as_dev(_rnn, dev).reshape(nseq, HS), masks_src=_as_dev(_masks, dev))
            idx = seq["idx"]
            self._update_core(obs.reshape(m, -1), idx, seq["L"] * seq["m_pad"], _as_dev(actions, dev).reshape(m, -1),
                              None if avail is None else _as_dev(avail, dev).reshape(m, -1),
                              _as_dev(old_logp, dev).reshape(m, -1), _as_dev(adv, dev).reshape(m), None,
                              None if factor is None else _as_dev(factor, dev).reshape(m),
                              _as_dev(active, dev).reshape(m) if self.use_policy_active_masks else None, seq=seq)
            d = self._info - before
            return d[0], d[1], d[2], d[3]
        self._update_core(obs.reshape(m, -1), None, m, _as_dev(actions, dev).reshape(m, -1),
                          None if avail is None else _as_dev(avail, dev).reshape(m, -1),
                          _as_dev(old_logp, dev).reshape(m, -1), _as_dev(adv, dev).reshape(m), None,
                          None if factor is None else _as_dev(factor, dev).reshape(m),
                          _as_dev(active, dev).reshape(m) if self.use_policy_active_masks else None)
        d = self._info - before
        return d[0], d[1], d[2], d[3]

    _INFO_KEYS = ("policy_loss", "dist_entropy", "actor_grad_norm", "ratio")

    def masked_moments(self, actor_buffer: OnPolicyActorBuffer, advantages, out: torch.Tensor) -> None:
        """fp64 {sum, sumsq, count} of ``advantages`` over this agent's active entries (happo.py:119-127) -> out[3]."""
        T, N = actor_buffer.actions.shape[:2]
        adv = _as_dev(advantages, self.device).reshape(T * N)
        call("harl_masked_moments", ptr(adv), ptr(actor_buffer.flat("active_masks")), T * N, ptr(out), _lib.scratch("mm"), stream())

    def rng_footprint(self, actor_buffer: OnPolicyActorBuffer) -> list:
        """Sizes of the ``torch.randperm`` draws ONE train() of this actor takes from the global CPU generator, in order (one
        per epoch: on_policy_actor_buffer.py:131 / :196 / :241 over the GLOBAL buffer).  The runner fast-forwards the
        generator over them to start the critic's update -- whose samplers draw AFTER every actor's in the reference --
        next to the actors' (runner.train, round 6)."""
        T, N = actor_buffer.actions.shape[:2]
        n_g = self.shard[0] if self.shard else N
        if self.use_recurrent_policy:
            size = (T * n_g) // self.data_chunk_length
        elif self.use_naive_recurrent_policy:
            size = n_g
        else:
            size = T * n_g
        return [size] * self.ppo_epoch

    def fuses_old_logp(self) -> bool:
        """True when train()'s first forward sees every buffer row, in order, under the pre-update parameters -- i.e. it
        computes exactly what the runner's pre-update log-prob pass computes (on_policy_ha_runner.py:66-83)."""
        return self.actor_num_mini_batch == 1 and not (self.use_recurrent_policy or self.use_naive_recurrent_policy)

    def train(self, actor_buffer: OnPolicyActorBuffer, advantages, state_type, _pre=None, _defer=False,
              _old_logp_out=None):
        """ppo_epoch x actor_num_mini_batch updates (happo.py:104-158).  ``advantages`` is the raw [T, N, 1]
        advantage tensor; the per-agent masked normalisation (happo.py:122-127) is folded into the loss kernel
        through the fp64 moments {sum, sumsq, count}.
        Runner-internal: ``_pre = (moments[3] device fp64, count)`` when the runner already reduced the moments of all
        agents with one read-back; ``_defer`` returns the averaged statistics as a device tensor (resolved by the
        runner in one transfer at the end of train()) and leaves deferred RNG advances pending; ``_old_logp_out``
        [T*N, act_w] receives the pre-update log-probs from the first epoch's forward (only if ``fuses_old_logp()``)."""
        dev = self.device
        buf = actor_buffer
        T, N = buf.actions.shape[:2]
        B = T * N
        if _pre is None:  # called directly (the runner invalidates once, before its own pre-update log-prob pass)
            self.actor.invalidate_caches()
        train_info = {k: 0.0 for k in self._INFO_KEYS}
        adv = _as_dev(advantages, dev).reshape(B).contiguous()
        active = buf.flat("active_masks").reshape(B)
        if _pre is None:
            moments = torch.zeros(3, dtype=torch.float64, device=dev)
            self.masked_moments(buf, adv, moments)
            self.comm.all_reduce_sum(moments)
            count = float(moments[2].item())
        else:
            moments, count = _pre
        if count == 0.0:  # np.all(active_masks[:-1] == 0) early-out (happo.py:119-120)
            return train_info
        if state_type != "EP":  # FP: the runner already normalised over all agents (on_policy_ha_runner.py:36-45)
            moments = None
        self._info.zero_()
        buf.__dict__.pop("_seq_cache", None)  # the recurrent samplers' per-update table never outlives one train() (buffers._recurrent_seqs)
        self.actor.fold()
        obs = buf.flat("obs")
        actions = buf.flat("actions")
        avail = None if buf.available_actions is None else buf.flat("available_actions")
        old_logp = buf.flat("action_log_probs")
        factor = None if buf.factor is None else buf.factor.reshape(B)  # None: MAPPO (no sequential-update factor)
        n_global = self.shard[0] * T if self.shard else B
        for epoch in range(self.ppo_epoch):
            if self.use_recurrent_policy or self.use_naive_recurrent_policy:
                for seq in buf.recurrent_batches(self.actor_num_mini_batch, self.data_chunk_length,
                                                 naive=not self.use_recurrent_policy, shard=self.shard):
                    if seq.get("empty"):
                        self._optimizer_step(0)
                        continue
                    self._update_core(obs, seq["idx"], seq["L"] * seq["m_pad"], actions, avail, old_logp, adv, moments,
                                      factor, active if self.use_policy_active_masks else None, seq=seq)
                continue
            if self.actor_num_mini_batch == 1:
                # the single "minibatch" is the whole (local) buffer: the sums do not depend on the order, so only the
                # generator state is replayed (bit-exact RNG stream), not the 819200-element shuffle itself
                consume_randperm(n_global)
                self._update_core(obs, None, B, actions, avail, old_logp, adv, moments, factor,
                                  active if self.use_policy_active_masks else None,
                                  logp_out=_old_logp_out if epoch == 0 else None)
                continue
            sampler = minibatch_indices(n_global, self.actor_num_mini_batch, dev)  # CPU RNG draw, bit-exact with the reference
            for ind in sampler:
                if self.shard:
                    ind = local_minibatch_rows(ind, self.shard[0], self.shard[1], self.shard[2])
                    if ind.numel() == 0:
                        self._optimizer_step(0)
                        continue
                self._update_core(obs, ind.to(dev), ind.numel(), actions, avail, old_logp, adv, moments, factor,
                                  active if self.use_policy_active_masks else None)
        n_upd = self.ppo_epoch * self.actor_num_mini_batch
        if _defer:
            return self._info / n_upd
        rng_sync()
        vals = (self._info / n_upd).cpu().tolist()  # the single read-back of this agent's update
        for k, v in zip(self._INFO_KEYS, vals):
            train_info[k] = v
        return train_info


class HAA2C(HAPPO):
    """HAPPO without ratio clipping, ``a2c_epoch`` epochs (reference: harl/algorithms/actors/haa2c.py:10-153)."""

    def __init__(self, args, obs_space, act_space, device=torch.device("cuda:0")):
        a = dict(args)
        a.setdefault("clip_param", 0.0)   # unused by the unclipped surrogate
        a.setdefault("ppo_epoch", a["a2c_epoch"])
        super().__init__(a, obs_space, act_space, device)
        self.args = args
        self.a2c_epoch = self.ppo_epoch = args["a2c_epoch"]
        self._surrogate_mode = 2
