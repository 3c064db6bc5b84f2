"""HATRPO on MI355X (reference: harl/algorithms/actors/hatrpo.py:18-247, harl/utils/trpo_util.py:5-158).

update() = surrogate gradient -> 10 conjugate-gradient steps on Fisher-vector products -> step size from the KL
threshold -> backtracking line search on (KL, surrogate improvement).  The Fisher-vector product is evaluated as
J^T M (J v): a forward-mode tangent pass through the MLP, the KL Hessian w.r.t. the distribution parameters, and the
same backward kernels HAPPO uses -- exact at theta_new == theta_old, where the reference's double backward computes
the same matrix.  Vectors of parameter size (P ~ 20-70 k) are combined with torch device ops; the control decisions
of CG / line search are host-side like the reference's.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream
from .buffers import OnPolicyActorBuffer, consume_randperm, rng_sync
from .happo import OnPolicyBase
from .nets import build_seq, consume_policy_init_rng, seq_compact
from .valuenorm import _as_dev


class HATRPO(OnPolicyBase):
    def __init__(self, args, obs_space, act_space, device=torch.device("cuda:0")):
        assert act_space.__class__.__name__ != "MultiDiscrete", \
            "only continuous and discrete action space is supported by HATRPO."
        super().__init__(args, obs_space, act_space, device)
        self.kl_threshold = args["kl_threshold"]
        self.ls_step = args["ls_step"]
        self.accept_ratio = args["accept_ratio"]
        self.backtrack_coeff = args["backtrack_coeff"]
        self._tangent_ws = None
        self._grad_tap = None
        self._cg_tap = None  # test hook: called with (k, x_k) after CG steps 1, 5 and 10
        self._moments = None
        self._vec_ws = None  # P-sized vectors + the update's device record (see _update_core)
        self._trace = None  # test hook: list receiving one dict per update (accept decision, backtracks, the five statistics)

    # ---- surrogate  sum_s ratio*f*adv*active / sum(active)  (hatrpo.py:77-90), optionally with its gradient ---------
    def _surrogate(self, obs, m, actions, avail, old_logp, adv, adv_moments, factor, active, want_grad: bool, seq=None,
                   raw: bool = False):
        """Returns (scalars, gradient).  ``raw`` (the update itself): the UNSCALED sums in place -- ``net.scalars`` and
        ``net.flat_grad`` with its log_std block still unset; harl_trpo_begin / harl_trpo_ls_test divide by sum(active) and take
        the log_std gradient from the scalars.  Default (tests, diagnostics): copies, the gradient scaled by 1 / sum(active).
        ``seq`` (GRU policies): the batch is L x m_pad rows in the recurrent layout (nets.build_seq), m = L * m_pad and the
        row arrays are the flat buffers gathered through seq['idx'] inside the kernels."""
        net = self.actor
        idx = None if seq is None else seq["idx"]
        net.forward_trunk(obs, idx, m, for_backward=True, seq=seq)
        Wp, bp = net._packs[-1]
        s = stream()
        fx, fmask, frstd, fh = net.feat()
        mv, mp = (seq["m"], seq["m_pad"]) if seq is not None else (0, 0)
        call("harl_actor_head_loss", ptr(fx), ptr(fmask), ptr(frstd), m, fh,
             ptr(Wp), ptr(bp), ptr(net.log_std()), net.std_x_coef, net.std_y_coef, int(net.discrete), net.act_dim,
             ptr(idx), ptr(actions), ptr(avail), ptr(old_logp), ptr(adv), ptr(adv_moments), ptr(factor), ptr(active),
             0.0, 0.0, int(self.action_aggregation == "mean"), 1, mv, mp, None, ptr(net.dz[0]), ptr(net.dhead), ptr(net.part_scalars), None, 0, s)
        call("harl_reduce_scalars_set", ptr(net.part_scalars), _lib.load().harl_head_blocks(m), ptr(net.scalars), s)
        grad = None
        if want_grad:
            net.backward_trunk(obs, idx, m, seq=seq)
            net.unfold_grads()
            grad = net.flat_grad
        sc = net.scalars
        if self.comm.enabled:  # in place: both are recomputed by the next pass
            self.comm.all_reduce_sum(sc)
            if grad is not None:
                self.comm.all_reduce_sum(grad)
        if raw:
            return sc, grad
        sc = sc.clone()
        if grad is not None:
            if not net.discrete:  # (the loss kernel leaves d loss / d log_std among its scalar sums)
                net.gview("act.action_out.log_std").copy_(sc[8:8 + net.act_dim])
            grad = grad * (1.0 / sc[1]).to(torch.float32)
        return sc, grad

    # ---- Fisher-vector product (trpo_util.py:132-158):  F v + 0.1 v -----------------------------------------------
    def _fvp(self, obs, m, m_global, avail, vec: torch.Tensor, seq=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``avail`` is indexed by batch position (already gathered for recurrent batches)."""
        net = self.actor
        s = stream()
        L = len(net.hidden_sizes)
        hs = net.hidden_sizes
        idx = None if seq is None else seq["idx"]
        if self._tangent_ws is None or self._tangent_ws["rows"] < m:
            mp = ((m + 31) // 32) * 32
            self._tangent_ws = dict(rows=m, xd=[torch.empty(mp * h, **self.tpdv) for h in hs],
                                    pack_d=torch.empty_like(net.pack_arena))
            if net.recurrent:
                H = hs[-1]
                self._tangent_ws["gates"] = [torch.empty(mp * H, **self.tpdv) for _ in range(4)]  # g_r, g_z, g_nx, g_nh
                self._tangent_ws["ydot"] = torch.empty(mp * H, **self.tpdv)
        ws = self._tangent_ws
        vec = vec.contiguous()
        fp, pd = net.flat_param, ws["pack_d"]
        # tangent of every folded block, in table order, into an arena laid out like net.pack_arena
        if net.table is not None and os.environ.get("HARL_UNFOLD_TABLE", "1") != "0":  # one launch for all entries
            call("harl_fold_tangent_table", ptr(fp), ptr(vec), ptr(pd), ptr(net.table), net.n_entries,
                 sum(r[4] for r in net._table_rows), s)
        else:
            for (wo, bo, go, beo, o, k), (pw, pb, _, _) in zip(net._entries(), net._pack_slots):
                call("harl_fold_linear_tangent", ptr(fp[wo:]), ptr(fp[go:]) if go >= 0 else None,
                     ptr(fp[beo:]) if beo >= 0 else None, ptr(vec[wo:]), ptr(vec[bo:]), ptr(vec[go:]) if go >= 0 else None,
                     ptr(vec[beo:]) if beo >= 0 else None, ptr(pd[pw:]), ptr(pd[pb:]), o, k, s)
        packs_d = [(pd[pw:pw + o * k], pd[pb:pb + o]) for (pw, pb, o, k) in net._pack_slots]
        Wpd, bpd = packs_d[0]
        if net.act_id:
            # activation other than ReLU (round 4): the tangent of [Linear, act, LayerNorm] composed like the forward pass -- raw
            # GEMMs for the two halves of the pre-activation's tangent, one element-wise launch for act' and the LayerNorm
            # tangent (harl_act_ln_tangent).  x0n, x_hat_l, mean(a_l), rstd_l of the same rows are still in the workspace
            # from the forward pass of _surrogate().
            if ws.get("zd") is None or ws["zd"][0].numel() < ws["xd"][0].numel() // hs[0] * max(hs):
                rows = ws["xd"][0].numel() // hs[0]
                ws["zd"] = [torch.empty(rows * max(hs), **self.tpdv) for _ in range(2)]
                ws["zero_bias"] = torch.zeros(max(hs), **self.tpdv)
            zd1, zd2 = ws["zd"]
            call("harl_mlp_linear_wide", ptr(net.x0n), m, net.kp0, ptr(Wpd), net.in_dim, ptr(bpd), hs[0], ptr(net.w1img),
                 ptr(zd1), s, tag="linear_wide")  # the inputs carry no tangent: z_dot = W'_dot x0n + b'_dot
            call("harl_act_ln_tangent", ptr(zd1), None, ptr(net.xh[0]), ptr(net.amean[0]), ptr(net.rstd[0]), m, hs[0],
                 net.act_id, ptr(ws["xd"][0]), s, tag="act_ln_tangent")
            for l in range(1, L):
                Wp, _ = net._packs[l]
                Wpd, bpd = packs_d[l]
                call("harl_mlp_linear", ptr(ws["xd"][l - 1]), m, hs[l - 1], hs[l], ptr(Wp), ptr(ws["zero_bias"]), ptr(zd1), s,
                     tag="linear")
                call("harl_mlp_linear", ptr(net.xh[l - 1]), m, hs[l - 1], hs[l], ptr(Wpd), ptr(bpd), ptr(zd2), s, tag="linear")
                call("harl_act_ln_tangent", ptr(zd1), ptr(zd2), ptr(net.xh[l]), ptr(net.amean[l]), ptr(net.rstd[l]), m, hs[l],
                     net.act_id, ptr(ws["xd"][l]), s, tag="act_ln_tangent")
        elif net.panel:  # width 256 (csrc/panel.hip, round 4): the same K-panel walk with the LayerNorm Jacobian as epilogue
            call("harl_mlp_panel_tangent", None, ptr(net.x0n), m, net.kp0, None, ptr(Wpd), net.in_dim, ptr(bpd), ptr(net.xh[0]),
                 ptr(net.rmask[0]), ptr(net.rstd[0]), ptr(ws["xd"][0]), s, tag="tangent_panel")
            for l in range(1, L):
                Wp, _ = net._packs[l]
                Wpd, bpd = packs_d[l]
                call("harl_mlp_panel_tangent", ptr(ws["xd"][l - 1]), ptr(net.xh[l - 1]), m, 256, ptr(Wp), ptr(Wpd), 256, ptr(bpd),
                     ptr(net.xh[l]), ptr(net.rmask[l]), ptr(net.rstd[l]), ptr(ws["xd"][l]), s, tag="tangent_panel")
        elif net.wide:  # x0n of the same rows is still there from the forward pass of _surrogate()
            call("harl_mlp_tangent_wide", ptr(net.x0n), m, net.kp0, ptr(Wpd), net.in_dim, ptr(bpd), hs[0], ptr(net.w1img),
                 ptr(net.xh[0]), ptr(net.rmask[0]), ptr(net.rstd[0]), ptr(ws["xd"][0]), s, tag="tangent_wide")
        else:
            call("harl_mlp_tangent_input", ptr(obs), obs.shape[1], ptr(idx), m, net.in_dim, ptr(Wpd), ptr(bpd),
                 int(net.use_feature_normalization), hs[0], ptr(net.xh[0]), ptr(net.rmask[0]), ptr(net.rstd[0]),
                 ptr(ws["xd"][0]), s)
        one_launch = os.environ.get("HARL_TANGENT_ONE_LAUNCH", "1") != "0"
        for l in range(1, L if not (net.act_id or net.panel) else 1):
            Wp, _ = net._packs[l]
            Wpd, bpd = packs_d[l]
            if one_launch:  # [W' | W'_dot] [x_dot ; x_hat] as ONE K = 2 H GEMM, weight images streamed from L2 (csrc/wide.hip)
                if ws.get("timg") is None or ws["timg"].numel() < 3 * hs[l] * hs[l - 1]:
                    ws["timg"] = torch.empty(3 * max(hs) * max(hs), **self.tpdv)  # 3 terms x HO x 2 HI bf16
                call("harl_mlp_tangent_hidden2", ptr(ws["xd"][l - 1]), ptr(net.xh[l - 1]), m, hs[l - 1], hs[l], ptr(Wp), ptr(Wpd),
                     ptr(bpd), ptr(ws["timg"]), ptr(net.xh[l]), ptr(net.rmask[l]), ptr(net.rstd[l]), ptr(ws["xd"][l]), s,
                     tag="tangent_hidden")
            else:
                call("harl_mlp_tangent_hidden", ptr(ws["xd"][l - 1]), ptr(net.xh[l - 1]), m, hs[l - 1], hs[l], ptr(Wp), ptr(Wpd),
                     ptr(bpd), ptr(net.xh[l]), ptr(net.rmask[l]), ptr(net.rstd[l]), ptr(ws["xd"][l]), s, tag="tangent_hidden")
        fx, fmask, frstd, fh = net.feat()
        xLdot = ws["xd"][-1]
        mv, mp_ = 0, 0
        if net.recurrent and net.gru_wide:  # 128-wide / stacked GRUs: the composed tangent (gru_wide.tangent, round 4)
            from . import gru_wide
            xLdot = gru_wide.tangent(net, seq, ws["xd"][-1], pd, ws)
            mv, mp_ = seq["m"], seq["m_pad"]
        elif net.recurrent:  # tangent through the recurrence (csrc/gru.hip): parallel gate pre-pass + sequential kernel
            H, gp, sv = hs[-1], net.gru_pack, net.rnn_saved
            b0, n3 = net._gru_pack_base, 3 * H * H
            Wihd, bihd = pd[b0:b0 + n3], pd[b0 + n3:b0 + n3 + 3 * H]
            Whhd, bhhd = pd[b0 + n3 + 3 * H:b0 + 2 * n3 + 3 * H], pd[b0 + 2 * n3 + 3 * H:b0 + 2 * n3 + 6 * H]
            g_r, g_z, g_nx, g_nh = ws["gates"]
            n_slabs = m // 32
            call("harl_gru_gates", ptr(ws["xd"][-1]), ptr(gp["Wih"]), H, n_slabs, ptr(g_r), ptr(g_z), ptr(g_nx), 0, s)
            call("harl_gru_gates", ptr(net.xh[-1]), ptr(Wihd), H, n_slabs, ptr(g_r), ptr(g_z), ptr(g_nx), 7, s)
            call("harl_gru_gates", ptr(sv[0]), ptr(Whhd), H, n_slabs, ptr(g_r), ptr(g_z), ptr(g_nh), 3, s)
            call("harl_gru_tangent", ptr(g_r), ptr(g_z), ptr(g_nx), ptr(g_nh), ptr(seq["mask_rows"]), ptr(gp["Whh"]),
                 ptr(bihd), ptr(bhhd), ptr(sv[0]), ptr(sv[1]), ptr(sv[2]), ptr(sv[3]), ptr(sv[4]), ptr(net.rnn_y),
                 ptr(net.rnn_rstd), H, seq["L"], seq["m_pad"], ptr(ws["ydot"]), s, tag="gru_tangent")
            xLdot = ws["ydot"]
            mv, mp_ = seq["m"], seq["m_pad"]
        Whp, bhp = net._packs[-1]
        Whpd, bhpd = packs_d[-1]
        call("harl_actor_head_fvp", ptr(fx), ptr(xLdot), ptr(fmask), ptr(frstd), m, fh,
             ptr(Whp), ptr(bhp), ptr(Whpd), ptr(bhpd), ptr(net.log_std()), net.std_x_coef, net.std_y_coef,
             int(net.discrete), net.act_dim, ptr(avail), mv, mp_, ptr(net.dz[0]), ptr(net.dhead), s)
        net.backward_trunk(obs, idx, m, seq=seq)
        net.unfold_grads()
        g = net.flat_grad
        if self.comm.enabled:
            g = g.clone()
            if not net.discrete:  # the log_std block is analytic (below), not part of the all-reduced sum
                off0 = net.offsets["act.action_out.log_std"][0]
                g[off0:off0 + net.act_dim] = 0.0
            self.comm.all_reduce_sum(g)
        # kl.mean() over the (global) batch, the analytic log_std block (d2 KL / d sigma^2 = 2 / sigma^2 per sample, sigma =
        # sigmoid(ls / xc) yc) and the 0.1 damping in ONE launch (csrc/elementwise.hip)
        if out is None:
            out = torch.empty_like(vec)
        ls_off = -1 if net.discrete else net.offsets["act.action_out.log_std"][0]
        call("harl_trpo_fvp_finish", ptr(g), ptr(vec), ptr(net.log_std()), ptr(out), out.numel(), float(m_global), 0.1,
             int(ls_off), net.act_dim, net.std_x_coef, net.std_y_coef, s)
        return out

    def _head_outputs(self, obs, m, actions, avail, reuse_trunk=False, seq=None, avail_rows=None) -> torch.Tensor:
        """Distribution parameters at the current weights: Gaussian mean / normalised logits, [rows, act_dim].
        ``reuse_trunk``: a _surrogate() call under the same weights has just left x_hat_L in the workspace.
        Recurrent batches: always reuse (the caller has just run _surrogate on the same layout); ``avail_rows`` is the
        availability mask by batch position; padding sequences are compacted away (rows = L * m)."""
        net = self.actor
        if seq is None:
            out = torch.empty(m, net.act_dim, **self.tpdv)
            self._logp_pass(obs, actions, avail, m, None, head_out=out, reuse_trunk=reuse_trunk)
            return out
        fx, _, _, fh = net.feat()
        Wp, bp = net._packs[-1]
        ho = torch.empty(m, net.act_dim, **self.tpdv)
        call("harl_actor_head_logp", ptr(fx), m, fh, ptr(Wp), ptr(bp), ptr(net.log_std()), net.std_x_coef, net.std_y_coef,
             int(net.discrete), net.act_dim, None, ptr(avail_rows), None, None, None, 0, ptr(ho), seq["m"], seq["m_pad"],
             stream(), tag="actor_head_logp")
        return seq_compact(ho, seq) if seq["m_pad"] != seq["m"] else ho

    def _kl_sum(self, head_old, ls_old, head_new, m, acc: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Sum over the (global) batch of KL(old || new), fp64 device scalar [1] (``acc``: accumulated into; the line search's
        accumulator is left at zero by harl_trpo_ls_test)."""
        net = self.actor
        if acc is None:
            acc = torch.zeros(1, dtype=torch.float64, device=self.device)
        call("harl_trpo_kl_sum", ptr(head_old), ptr(head_new), ptr(ls_old), ptr(net.log_std()), net.std_x_coef,
             net.std_y_coef, m, net.act_dim, int(net.discrete), ptr(acc), stream())
        if self.comm.enabled:
            self.comm.all_reduce_sum(acc)
        return acc

    def _kl_mean(self, head_old, ls_old, head_new, m, m_global) -> float:
        return float(self._kl_sum(head_old, ls_old, head_new, m).item()) / float(m_global)

    def _update_core(self, obs, m, m_global, actions, avail, old_logp, adv, adv_moments, factor, active, seq=None):
        """``seq``: recurrent batch layout (m = L * m_pad rows, m_global = global number of real rows)."""
        net = self.actor
        net.fold()
        avail_rows, kl_rows = avail, m
        if seq is not None:
            kl_rows = seq["L"] * seq["m"]
            if avail is not None and seq["idx"] is not None:
                avail_rows = avail[seq["idx"]].contiguous()  # by batch position, for the FVP / head-output kernels
        # P-sized vectors and the update's 16-double record: allocated once per actor (harl_trpo_* keep every scalar decision on
        # the device; no torch arithmetic, no rocBLAS between the first and the last launch of an update)
        n = net.flat_param.numel()
        w = self._vec_ws
        if w is None or w["n"] != n:
            f = lambda: torch.empty(n, **self.tpdv)  # noqa: E731
            w = self._vec_ws = dict(n=n, g=f(), x=f(), r=f(), p=f(), fp=f(), theta=f(), full_step=f(),
                                    cg_state=torch.zeros(2, **self.tpdv),
                                    st=torch.zeros(16, dtype=torch.float64, device=self.device),
                                    kl=torch.zeros(1, dtype=torch.float64, device=self.device),
                                    st_host=torch.zeros(16, dtype=torch.float64).pin_memory())
        g, x, r, p, st = w["g"], w["x"], w["r"], w["p"], w["st"]
        s = stream()
        sc, graw = self._surrogate(obs, m, actions, avail, old_logp, adv, adv_moments, factor, active, want_grad=True, seq=seq,
                                   raw=True)
        # g = grad / sum(active) (log_std block from the loss kernel's scalar sums), x = 0, r = p = g, r.r
        # (hatrpo.py:92-95, trpo_util.py:101-105)
        ls_off = -1 if net.discrete else net.offsets["act.action_out.log_std"][0]
        call("harl_trpo_begin", ptr(graw), ptr(sc), int(ls_off), net.act_dim, ptr(g), ptr(x), ptr(r), ptr(p), n,
             ptr(w["cg_state"]), ptr(st), _lib.scratch("cg"), s)
        # conjugate gradient, 10 steps, residual tolerance 1e-10 (trpo_util.py:96-129)
        # The reference leaves the loop once rdotr < 1e-10; here that test stays on the device (a `done` flag freezes x, r
        # and p from then on -- the same iterates, no host round trip per iteration, at the price of idle FVPs after a break).
        for it in range(10):
            avp = self._fvp(obs, m, m_global, avail_rows, p, seq=seq, out=w["fp"])
            call("harl_trpo_cg_step", ptr(x), ptr(r), ptr(p), ptr(avp), n, ptr(w["cg_state"]), _lib.scratch("cg"), s)
            if self._cg_tap is not None and it + 1 in (1, 5, 10):
                self._cg_tap(it + 1, x.clone())
        fv = self._fvp(obs, m, m_global, avail_rows, x, seq=seq, out=w["fp"])
        # shs, step size, full_step, the snapshot of theta_old, expected improvement (hatrpo.py:123-133)
        call("harl_trpo_step", ptr(x), ptr(fv), ptr(g), ptr(net.flat_param), ptr(w["theta"]), ptr(w["full_step"]), n,
             float(self.kl_threshold), ptr(st), _lib.scratch("cg"), s)
        # "old actor" snapshot (hatrpo.py:127-130): distribution parameters at theta_old + the RNG draws its construction costs
        head_old = self._head_outputs(obs, m, actions, avail, reuse_trunk=True, seq=seq, avail_rows=avail_rows)  # FVPs do not touch x_hat_l / y
        ls_old = None if net.discrete else net.log_std().clone()
        consume_policy_init_rng(self.args, self.obs_space, self.act_space)
        if self._grad_tap is not None:
            torch.cuda.synchronize()
            self._grad_tap(g.clone(), x.clone(), float(st[2].item()))
        rec = None
        for _ in range(self.ls_step):
            call("harl_trpo_ls_candidate", ptr(w["theta"]), ptr(w["full_step"]), ptr(st), ptr(net.flat_param), n, s)
            net._fold_version = None  # (the parameters changed behind torch's version counter)
            net.fold()
            sc_new, _ = self._surrogate(obs, m, actions, avail, old_logp, adv, adv_moments, factor, active, want_grad=False,
                                        seq=seq, raw=True)
            head_new = self._head_outputs(obs, m, actions, avail, reuse_trunk=True, seq=seq, avail_rows=avail_rows)
            self._kl_sum(head_old, ls_old, head_new, kl_rows, acc=w["kl"])
            # kl, improvement, the accept test and the backtrack bookkeeping (hatrpo.py:171-185) on the device ...
            call("harl_trpo_ls_test", ptr(sc_new), ptr(w["kl"]), float(m_global), float(self.kl_threshold), float(self.accept_ratio),
                 float(self.backtrack_coeff), ptr(st), s)
            # ... and ONE read-back per line-search step: the record
            w["st_host"].copy_(st, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            rec = w["st_host"].tolist()
            if rec[5] != 0.0:
                break
        flag = rec[5] != 0.0
        if not flag:
            net.flat_param.copy_(w["theta"])
            net._fold_version = None
            net.fold()
            print("policy update does not impove the surrogate")
        kl, loss_improve, expected_improve, dist_entropy, ratio = rec[7], rec[8], rec[3], rec[10], rec[11]
        if self._trace is not None:
            self._trace.append(dict(accepted=bool(flag), backtracks=int(rec[6]), fraction=rec[4], kl=kl, loss=rec[0],
                                    loss_improve=loss_improve, expected_improve=expected_improve, dist_entropy=dist_entropy,
                                    ratio=ratio, step_size=rec[2], shs=rec[1]))
        return kl, loss_improve, expected_improve, dist_entropy, ratio

    def update(self, sample):
        """API-compatible update on an already-gathered sample (tuple order of hatrpo.py:50-60)."""
        (obs, _rnn, actions, _masks, active, old_logp, adv, avail, factor) = sample
        dev = self.device
        obs = _as_dev(obs, dev)
        m = obs.shape[0]
        if self.actor.recurrent:  # gathered [L*m, .] l-major sample + rnn_states [m, 1, H] (recurrent generators)
            HS = self.actor.hidden_sizes[-1] * self.actor.recurrent_n
            nseq = _as_dev(_rnn, dev).shape[0]
            seq = build_seq(dev, m // nseq, nseq, HS, h0=_as_dev(_rnn, dev).reshape(nseq, HS), masks_src=_as_dev(_masks, dev))
            return self._update_core(obs.reshape(m, -1), seq["L"] * seq["m_pad"], m, _as_dev(actions, dev).reshape(m, -1),
                                     None if avail is None else _as_dev(avail, dev).reshape(m, -1),
                                     _as_dev(old_logp, dev).reshape(m, -1), _as_dev(adv, dev).reshape(m), None,
                                     _as_dev(factor, dev).reshape(m),
                                     _as_dev(active, dev).reshape(m) if self.use_policy_active_masks else None, seq=seq)
        return self._update_core(obs.reshape(m, -1), m, m, _as_dev(actions, dev).reshape(m, -1),
                                 None if avail is None else _as_dev(avail, dev).reshape(m, -1),
                                 _as_dev(old_logp, dev).reshape(m, -1), _as_dev(adv, dev).reshape(m), None,
                                 _as_dev(factor, dev).reshape(m),
                                 _as_dev(active, dev).reshape(m) if self.use_policy_active_masks else None)

    def train(self, actor_buffer: OnPolicyActorBuffer, advantages, state_type):
        """One full-batch update (hatrpo.py:196-247)."""
        dev = self.device
        buf = actor_buffer
        T, N = buf.actions.shape[:2]
        B = T * N
        info = {"kl": 0.0, "dist_entropy": 0.0, "loss_improve": 0.0, "expected_improve": 0.0, "ratio": 0.0}
        self.actor.invalidate_caches()
        adv = _as_dev(advantages, dev).reshape(B).contiguous()
        active = buf.flat("active_masks").reshape(B)
        if self._moments is None:
            self._moments = torch.zeros(3, dtype=torch.float64, device=dev)
        moments = self._moments
        call("harl_zero_bytes", ptr(moments), 24, stream())
        call("harl_masked_moments", ptr(adv), ptr(active), B, ptr(moments), _lib.scratch("mm"), stream())
        self.comm.all_reduce_sum(moments)
        if float(moments[2].item()) == 0.0:
            return info
        if state_type != "EP":  # FP: advantages arrive normalised over all agents (on_policy_ha_runner.py:36-45)
            moments = None
        n_global = self.shard[0] * T if self.shard else B
        buf.__dict__.pop("_seq_cache", None)  # (buffers._recurrent_seqs: one table per update)
        if self.use_recurrent_policy or self.use_naive_recurrent_policy:
            # recurrent_generator_actor(advantages, 1, L) / naive_recurrent_generator_actor(advantages, 1): ONE sample that
            # holds every chunk (hatrpo.py:222-231)
            for seq in buf.recurrent_batches(1, self.data_chunk_length, naive=not self.use_recurrent_policy, shard=self.shard):
                if seq.get("empty"):
                    raise NotImplementedError("HATRPO: a rank without any sequence of the (single) batch")
                kl, li, ei, ent, ratio = self._update_core(
                    buf.flat("obs"), seq["L"] * seq["m_pad"], seq["L"] * seq["m_global"], buf.flat("actions"),
                    None if buf.available_actions is None else buf.flat("available_actions"), buf.flat("action_log_probs"),
                    adv, moments, buf.factor.reshape(B), active if self.use_policy_active_masks else None, seq=seq)
                info.update(kl=kl, dist_entropy=ent, loss_improve=li, expected_improve=ei, ratio=ratio)
            rng_sync()
            return info
        consume_randperm(n_global)  # feed_forward_generator_actor(advantages, 1): one draw, whole buffer
        kl, li, ei, ent, ratio = self._update_core(
            buf.flat("obs"), B, n_global, buf.flat("actions"),
            None if buf.available_actions is None else buf.flat("available_actions"), buf.flat("action_log_probs"), adv,
            moments, buf.factor.reshape(B), active if self.use_policy_active_masks else None)
        info.update(kl=kl, dist_entropy=ent, loss_improve=li, expected_improve=ei, ratio=ratio)
        rng_sync()
        return info
