"""A/B of the fused optimiser-step kernels (csrc/update.hip) against the layer-by-layer path on the GPU box:
parity vs the oracle on small shapes, old-vs-new gradient agreement and per-kernel timings at BASELINE size.

    python tools/fused_ab.py [--rows 819200] [--reps 5]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from harl_amd import _lib  # noqa: E402
from tests import gpu_checks as G  # noqa: E402
from harl_amd.synthetic import Shapes, make_buffers  # noqa: E402


def oracle_parity():
    out = {}
    for spec in (G.FWD_SHAPES[0], G.FWD_SHAPES[1]):
        for fused in ("1", "0"):
            os.environ["HARL_FUSED_UPDATE"] = fused
            r = G.check_gradients(spec)
            out[f"{spec['name']}|fused={fused}"] = {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in r.items()}
            f = G.check_forward(spec)
            out[f"{spec['name']}|fused={fused}|fwd"] = {k: f"{v:.2e}" for k, v in f.items()}
    os.environ["HARL_FUSED_UPDATE"] = "1"
    return out


def ab(rows: int, reps: int, discrete: bool = False):
    sh = Shapes(T=rows, N=1, A=1, obs_dim=18, share_obs_dim=54, act_dim=5, discrete=discrete, hidden_sizes=[128, 128])
    d = make_buffers(sh, 3)
    actor, _, _ = G._mk_actor(sh, 1)
    critic, _, _ = G._mk_critic(sh, 2)
    dev = G.dev
    obs = dev(d.obs[0][:-1].reshape(rows, -1))
    act = dev(d.actions[0].reshape(rows, -1))
    rng = np.random.default_rng(0)
    adv = dev(rng.standard_normal(rows).astype(np.float32))
    factor = dev((1 + 0.1 * rng.standard_normal(rows)).astype(np.float32))
    active = dev(np.ones(rows, dtype=np.float32))
    actor.actor.fold()
    lp0 = torch.empty(rows, actor.actor.act_w, device=G.DEV)
    os.environ["HARL_FUSED_UPDATE"] = "0"
    actor._logp_pass(obs, act, None, rows, lp0)
    old_logp = (lp0 + 0.1 * torch.randn_like(lp0)).contiguous()
    so = dev(d.share_obs[:-1].reshape(rows, -1))
    vp = dev(rng.standard_normal(rows).astype(np.float32))
    ret = dev((vp.cpu().numpy() + rng.standard_normal(rows)).astype(np.float32))
    res = {}
    grads = {}
    for fused in ("0", "1", "1"):
        os.environ["HARL_FUSED_UPDATE"] = fused
        actor.actor.invalidate_caches()
        critic.critic.invalidate_caches()
        lp = torch.zeros(rows, actor.actor.act_w, device=G.DEV)
        for _ in range(2):  # warm-up
            actor._forward_backward(obs, None, rows, act, None, old_logp, adv, None, factor, active, logp_out=lp)
        torch.cuda.synchronize()
        _lib.enable_kernel_timing(True)
        for _ in range(reps):
            nblk = actor._forward_backward(obs, None, rows, act, None, old_logp, adv, None, factor, active, logp_out=lp)
            lp2 = torch.empty(rows, actor.actor.act_w, device=G.DEV)
            fac = factor.clone()
            actor._logp_pass(obs, act, None, rows, lp2, old_logp=old_logp, factor=fac)
        t = _lib.collect_kernel_timing()
        _lib.enable_kernel_timing(False)
        sc = torch.zeros(_lib.PS_STRIDE, dtype=torch.float64, device=G.DEV)
        _lib.call("harl_reduce_scalars", _lib.ptr(actor.actor.part_scalars), nblk, _lib.ptr(sc), _lib.stream())
        key = f"actor|fused={fused}" + ("|again" if f"actor|fused={fused}" in grads else "")
        grads[key] = (actor.actor.dwp.clone(), sc.clone(), lp.clone(), lp2.clone(), fac.clone())
        res[key] = {k: round(v["avg_ms"], 4) for k, v in t.items()}
        # critic
        net = critic.critic
        net.fold()
        _lib.enable_kernel_timing(True)
        for _ in range(reps):
            net._ensure_ws(rows)
            s = _lib.stream()
            if net.fused_update_ok(None):
                _lib.call("harl_update_fwd_critic", *net.fused_args(so, rows), _lib.ptr(vp), _lib.ptr(ret), None, 0.2, 1, 1, 10.0,
                          _lib.ptr(net.dz[0]), _lib.ptr(net.part_scalars), _lib.ptr(net.part[net._part_offs[-1]:]), net.n_wg,
                          *net.hybrid_outputs(), s, tag="update_fwd_critic")
                net.backward_after_fused(so, rows)
            else:
                net.forward_trunk(so, None, rows)
                Wp, bp = net._packs[-1]
                fx, fm, fr, fh = net.feat()
                _lib.call("harl_critic_head_loss", _lib.ptr(fx), _lib.ptr(fm), _lib.ptr(fr), rows, fh, _lib.ptr(Wp), _lib.ptr(bp),
                          None, _lib.ptr(vp), _lib.ptr(ret), None, 0.2, 1, 1, 10.0, 0, 0, _lib.ptr(net.dz[0]),
                          _lib.ptr(net.dhead), _lib.ptr(net.part_scalars), _lib.ptr(net.part[net._part_offs[-1]:]),
                          net.n_wg, s, tag="critic_head_loss")
                net.backward_trunk(so, None, rows, head_dw_done=True)
        t = _lib.collect_kernel_timing()
        _lib.enable_kernel_timing(False)
        sc = torch.zeros(_lib.PS_STRIDE, dtype=torch.float64, device=G.DEV)
        _lib.call("harl_reduce_scalars", _lib.ptr(net.part_scalars), net.n_wg, _lib.ptr(sc), _lib.stream())
        ckey = f"critic|fused={fused}" + ("|again" if f"critic|fused={fused}" in grads else "")
        grads[ckey] = (net.dwp.clone(), sc.clone())
        res[ckey] = {k: round(v["avg_ms"], 4) for k, v in t.items()}

    def vrel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    for who in ("actor", "critic"):
        a, b, c = grads[f"{who}|fused=0"], grads[f"{who}|fused=1"], grads[f"{who}|fused=1|again"]
        res[f"{who}|dwp_vec_rel(new vs old)"] = f"{vrel(b[0], a[0]):.2e}"
        res[f"{who}|scalars_rel"] = f"{float(((b[1] - a[1]).abs() / a[1].abs().clamp_min(1e-30)).max()):.2e}"
        res[f"{who}|rerun_bitwise_equal"] = bool(torch.equal(b[0], c[0]))
        res[f"{who}|scalars_old"] = [f"{x:.9g}" for x in a[1][:14].tolist()]
        res[f"{who}|scalars_new"] = [f"{x:.9g}" for x in b[1][:14].tolist()]
        net = actor.actor if who == "actor" else critic.critic
        offs = list(net._dwp_offs) + [net.total_dwp]
        res[f"{who}|dwp_rel_per_entry"] = [f"{vrel(b[0][offs[k]:offs[k + 1]], a[0][offs[k]:offs[k + 1]]):.2e}" for k in range(len(offs) - 1)]
        if who == "actor":
            res["actor|logp_first_epoch_vec_rel"] = f"{vrel(b[2], a[2]):.2e}"
            res["actor|logp_pass_vec_rel"] = f"{vrel(b[3], a[3]):.2e}"
            res["actor|factor_vec_rel"] = f"{vrel(b[4], a[4]):.2e}"
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=819200)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--skip-oracle", action="store_true")
    ap.add_argument("--only-full", action="store_true")
    ap.add_argument("--no-mid", action="store_true")
    a = ap.parse_args()
    out = {}
    if not a.skip_oracle:
        out["oracle_parity"] = oracle_parity()
    if not a.only_full:
        out["ab_small_ragged"] = ab(32 * 37 + 5, 2)
        out["ab_disc_small"] = ab(4099, 2, discrete=True)
    if not a.no_mid:
        out["ab_mid"] = ab(32 * 8 * 256 * 2 + 7 * 32 + 3, 2)
    out["ab_full"] = ab(a.rows, a.reps)
    print(json.dumps(out, indent=1))
