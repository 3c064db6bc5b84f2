"""V critic on MI355X (reference: harl/algorithms/critics/v_critic.py:14-208)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib
from ._lib import PS_STRIDE, call, ptr, stream
from .buffers import OnPolicyCriticBufferEP, consume_randperm, minibatch_indices, rng_sync
from .dist import Comm, local_minibatch_rows
from .nets import FusedAdam, VNet, build_seq
from .valuenorm import ValueNorm, _as_dev


class VCritic:
    def __init__(self, args, cent_obs_space, device=torch.device("cuda:0")):
        self.args = args
        self.device = torch.device(device)
        _lib.require_gpu(self.device)
        self.tpdv = dict(dtype=torch.float32, device=self.device)
        self.clip_param = args["clip_param"]
        self.critic_epoch = args["critic_epoch"]
        self.critic_num_mini_batch = args["critic_num_mini_batch"]
        self.data_chunk_length = args["data_chunk_length"]
        self.value_loss_coef = args["value_loss_coef"]
        self.max_grad_norm = args["max_grad_norm"]
        self.huber_delta = args["huber_delta"]
        self.use_recurrent_policy = args["use_recurrent_policy"]
        self.use_naive_recurrent_policy = args["use_naive_recurrent_policy"]
        self.use_max_grad_norm = args["use_max_grad_norm"]
        self.use_clipped_value_loss = args["use_clipped_value_loss"]
        self.use_huber_loss = args["use_huber_loss"]
        self.use_policy_active_masks = args["use_policy_active_masks"]
        self.critic_lr = args["critic_lr"]
        self.opti_eps = args["opti_eps"]
        self.weight_decay = args["weight_decay"]
        self.share_obs_space = cent_obs_space
        self.critic = VNet(args, cent_obs_space, self.device)
        self.critic_optimizer = FusedAdam(self.critic, self.critic_lr, self.opti_eps, self.weight_decay)
        self.comm = Comm()
        self.shard = None
        self._info = torch.zeros(2, dtype=torch.float64, device=self.device)  # fp64 sums of the per-update fp32 value_loss, critic_grad_norm (v_critic.py:186-187)
        self._grad_tap = None
        self._trace = None  # test hook: snapshots of the running statistics after every optimiser step
        self._state_tap = None  # test hook: (parameters, exp_avg, exp_avg_sq, step) BEFORE every optimiser step

    def lr_decay(self, episode, episodes):
        lr = self.critic_lr - (self.critic_lr * ((episode - 1) / float(episodes)))
        for g in self.critic_optimizer.param_groups:
            g["lr"] = lr

    def get_values(self, cent_obs, rnn_states_critic, masks):
        """values [B, 1] (device) = VNet(cent_obs)  (v_critic.py:62-73, v_net.py:48-67); recurrent nets also return the
        next hidden state [B, 1, H] (single GRU step when B == rnn_states.shape[0], rnn.py:24-33)."""
        x = _as_dev(cent_obs, self.device)
        x = x.reshape(x.shape[0], -1)
        M = x.shape[0]
        net = self.critic
        net.fold()
        Wp, bp = net._packs[-1]
        out = torch.empty(M, 1, **self.tpdv)
        if net.fused_update_ok(None, train=False):
            call("harl_update_values", *net.fused_args(x, M), ptr(out), stream(), tag="update_values")
            return out, (None if rnn_states_critic is None else _as_dev(rnn_states_critic, self.device))
        if not net.recurrent:
            net.forward_trunk(x, None, M, for_backward=False)
            call("harl_critic_head_values", ptr(net.xh[-1]), M, net.hidden_sizes[-1], ptr(Wp), ptr(bp), ptr(out), stream())
            return out, (None if rnn_states_critic is None else _as_dev(rnn_states_critic, self.device))
        H = net.hidden_sizes[-1]
        h0 = _as_dev(rnn_states_critic, self.device)
        m = h0.shape[0]
        HS = H * net.recurrent_n
        seq = build_seq(self.device, M // m, m, HS, h0=h0.reshape(m, HS), masks_src=_as_dev(masks, self.device),
                        want_h_last=True)
        Mp = seq["L"] * seq["m_pad"]
        net.forward_trunk(x, seq["idx"], Mp, for_backward=False, seq=seq)
        fx, _, _, fh = net.feat()
        outp = out if seq["idx"] is None else torch.empty(Mp, 1, **self.tpdv)
        call("harl_critic_head_values", ptr(fx), Mp, fh, ptr(Wp), ptr(bp), ptr(outp), stream())
        if seq["idx"] is not None:
            out.copy_(outp.reshape(seq["L"], seq["m_pad"], 1)[:, :m].reshape(M, 1))
        return out, seq["h_last"][:m].reshape(m, net.recurrent_n, H).clone()

    def _update_core(self, share_obs, idx, m, m_global, value_preds, returns, vn: Optional[ValueNorm], seq=None):
        """One optimiser step on rows idx[0..m) (m = 0: this rank holds none of the global minibatch's m_global rows and
        only takes part in the collectives)."""
        net = self.critic
        s = stream()
        if vn is not None:  # ValueNorm.update runs on this minibatch's returns BEFORE the targets are normalised
            vidx = seq["valid_idx"] if seq is not None else idx
            vn.update(returns, vidx, count=m_global, reduce_fn=self.comm.all_reduce_sum if self.comm.enabled else None,
                      local_count=(seq["L"] * seq["m"] if seq is not None else m))
        net._ensure_ws(max(m, 1))
        sc = net.scalars
        self._forward_backward(net, share_obs, idx, m, value_preds, returns, vn, seq, s)
        nblk = net.n_wg if m > 0 else 0  # rows of part_scalars
        ps_kw = dict(part_scalars=net.part_scalars, n_scalar_blocks=nblk)  # reduced inside the optimiser launch
        if self.comm.enabled:  # ONE collective per optimiser step: [folded gradients | loss scalars] (dist.py)
            if m <= 0:
                net.dwp.zero_()
            hilo = net.dwp_msg[net.total_dwp:]
            call("harl_reduce_pack_scalars", ptr(net.part_scalars), nblk, ptr(sc), ptr(hilo), s)  # (nblk = 0: zeros)
            self.comm.all_reduce_message(net.dwp_msg)
            ps_kw = dict(scalars_hilo=hilo)
        # loss = mean over the (global) minibatch, times value_loss_coef before backward (v_critic.py:112,146)
        scale = float(self.value_loss_coef) / float(m_global)
        if self._state_tap is not None:
            o = self.critic_optimizer
            self._state_tap.append((net.flat_param.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), o.step_count))
        self.critic_optimizer.step(1, scale, self.use_max_grad_norm, self.max_grad_norm, self._info, **ps_kw)
        if self._trace is not None:
            self._trace.append(self._info.clone())
        if self._grad_tap is not None:
            self._grad_tap(net.flat_grad * scale, sc.clone())

    def _forward_backward(self, net, share_obs, idx, m, value_preds, returns, vn, seq, s) -> None:
        """Forward, value loss and backward of one minibatch: the unscaled folded gradients and the loss kernel's partial sums."""
        if m > 0 and net.fused_update_ok(idx, seq):  # fused forward + loss (csrc/update.hip), then the layer backward (hybrid) or harl_update_bwd
            call("harl_update_fwd_critic", *net.fused_args(share_obs, m), ptr(value_preds), ptr(returns),
                 ptr(vn.stats) if vn is not None else None, float(self.clip_param), int(self.use_clipped_value_loss),
                 int(self.use_huber_loss), float(self.huber_delta), ptr(net.dz[0]), ptr(net.part_scalars),
                 ptr(net.part[net._part_offs[-1]:]), net.n_wg, *net.hybrid_outputs(), s, tag="update_fwd_critic")
            net.backward_after_fused(share_obs, m)
        elif m > 0 and net.fused_last_ok(idx, seq):  # deeper networks: the last hidden layer runs inside the loss launch
            L = len(net.hidden_sizes)
            net.forward_trunk(share_obs, idx, m, upto=L - 1)
            (Wl, bl), (Wh, bh) = net._packs[L - 1], net._packs[-1]
            call("harl_update_last_critic", ptr(net.xh[L - 2]), m, net.hidden_sizes[-1], ptr(Wl), ptr(bl), ptr(Wh), ptr(bh),
                 ptr(idx), ptr(value_preds), ptr(returns), ptr(vn.stats) if vn is not None else None, float(self.clip_param),
                 int(self.use_clipped_value_loss), int(self.use_huber_loss), float(self.huber_delta), ptr(net.dz[0]),
                 ptr(net.part_scalars), ptr(net.part[net._part_offs[-1]:]), net.n_wg, s, tag="update_last_critic")
            net.backward_trunk(share_obs, idx, m, head_dw_done=True)
        elif m > 0:
            net.forward_trunk(share_obs, idx, m, seq=seq)
            Wp, bp = net._packs[-1]
            fx, fmask, frstd, fh = net.feat()
            mv, mp = (seq["m"], seq["m_pad"]) if seq is not None else (0, 0)
            call("harl_critic_head_loss", ptr(fx), ptr(fmask), ptr(frstd), m, fh,
                 ptr(Wp), ptr(bp), ptr(idx), ptr(value_preds), ptr(returns), ptr(vn.stats) if vn is not None else None,
                 float(self.clip_param), int(self.use_clipped_value_loss), int(self.use_huber_loss), float(self.huber_delta),
                 mv, mp, ptr(net.dz[0]), ptr(net.dhead), ptr(net.part_scalars), ptr(net.part[net._part_offs[-1]:]),
                 net.n_wg, s, tag="critic_head_loss")  # head weight gradient fused into this launch
            net.backward_trunk(share_obs, idx, m, seq=seq, head_dw_done=True)

    def update(self, sample, value_normalizer=None):
        """API-compatible single update on a gathered minibatch (v_critic.py:116-157)."""
        share_obs, _rnn, value_preds, returns, _masks = sample
        dev = self.device
        x = _as_dev(share_obs, dev)
        m = x.shape[0]
        before = self._info.clone()
        self.critic.fold()
        if self.critic.recurrent:
            HS = self.critic.hidden_sizes[-1] * self.critic.recurrent_n
            h0 = _as_dev(_rnn, dev)
            seq = build_seq(dev, m // h0.shape[0], h0.shape[0], HS, h0=h0.reshape(-1, HS), masks_src=_as_dev(_masks, dev))
            self._update_core(x.reshape(m, -1), seq["idx"], seq["L"] * seq["m_pad"], m, _as_dev(value_preds, dev).reshape(m),
                              _as_dev(returns, dev).reshape(m), value_normalizer, seq=seq)
            d = self._info - before
            return d[0], d[1]
        self._update_core(x.reshape(m, -1), None, m, m, _as_dev(value_preds, dev).reshape(m),
                          _as_dev(returns, dev).reshape(m), value_normalizer)
        d = self._info - before
        return d[0], d[1]

    def train(self, critic_buffer: OnPolicyCriticBufferEP, value_normalizer: Optional[ValueNorm] = None, _defer=False):
        """critic_epoch x critic_num_mini_batch updates (v_critic.py:159-200).  ``_defer`` (runner-internal): return the
        averaged statistics as a device tensor and leave deferred RNG advances pending."""
        for _ in self._train_epochs(critic_buffer, value_normalizer):
            pass
        return self._train_result(_defer)

    def _train_result(self, _defer: bool):
        n_upd = self.critic_epoch * self.critic_num_mini_batch
        if _defer:
            return self._info / n_upd
        rng_sync()
        vals = (self._info / n_upd).cpu().tolist()
        return {"value_loss": vals[0], "critic_grad_norm": vals[1]}

    def _train_epochs(self, critic_buffer: OnPolicyCriticBufferEP, value_normalizer: Optional[ValueNorm] = None):
        """The body of train() as a generator that yields after every epoch: the runner hands the critic's epochs out one by
        one between the actors' updates (runner.train, round 6), each on the critic's stream behind the event of the agent it
        follows.  Draws from the global CPU generator happen inside the epoch they belong to, exactly as in train()."""
        buf = critic_buffer
        T, N = buf.rewards.shape[:2]
        A = getattr(buf, "num_agents", None)  # FP buffers carry an agent axis: rows = (t*N + n)*A + a
        B = T * N * (A or 1)
        dev = self.device
        self._info.zero_()
        self.critic.invalidate_caches()
        buf.__dict__.pop("_seq_cache", None)  # the recurrent samplers' per-update table never outlives one train() (buffers._recurrent_seqs)
        self.critic.fold()
        share_obs = buf.flat("share_obs")
        value_preds = buf.flat("value_preds").reshape(B)
        returns = buf.flat("returns").reshape(B)
        n_global = self.shard[0] * T * (A or 1) if self.shard else B
        for _ in range(self.critic_epoch):
            if self.use_recurrent_policy or self.use_naive_recurrent_policy:
                for seq in buf.recurrent_batches(self.critic_num_mini_batch, self.data_chunk_length,
                                                 naive=not self.use_recurrent_policy, shard=self.shard):
                    if seq.get("empty"):
                        self._update_core(share_obs, None, 0, seq["L"] * seq["m_global"], value_preds, returns,
                                          value_normalizer)
                        continue
                    self._update_core(share_obs, seq["idx"], seq["L"] * seq["m_pad"], seq["L"] * seq["m_global"],
                                      value_preds, returns, value_normalizer, seq=seq)
                yield
                continue
            if self.critic_num_mini_batch == 1:
                consume_randperm(n_global)  # replay the generator state only (see HAPPO.train)
                self._update_core(share_obs, None, B, n_global, value_preds, returns, value_normalizer)
                yield
                continue
            sampler = minibatch_indices(n_global, self.critic_num_mini_batch, dev)
            for ind in sampler:
                m_global = ind.numel()
                if self.shard:
                    ind = local_minibatch_rows(ind, self.shard[0], self.shard[1], self.shard[2], agents=A or 1)
                self._update_core(share_obs, ind.to(dev), ind.numel(), m_global, value_preds, returns, value_normalizer)
            yield

    def prep_training(self):
        self.critic.train()
        self.critic.invalidate_caches()

    def prep_rollout(self):
        # phase boundary = cache boundary: the folded weights and the normalised-input image are re-derived on first use of
        # every rollout / training phase, so a parameter write that torch's version counter cannot see (``p.data.copy_()``,
        # a raw-pointer kernel, load through DLPack) is picked up at the next prep_rollout() / prep_training() / train()
        self.critic.eval()
        self.critic.invalidate_caches()
