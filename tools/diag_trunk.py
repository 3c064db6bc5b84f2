"""Where do the one-launch trunk and the layer launches part ways?  Every intermediate image of one training forward + backward,
compared bit for bit (HARL_TRUNK_FUSED=0 / 1 on the same weights and rows)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as G  # noqa: E402


def run(mode, spec):
    os.environ["HARL_TRUNK_FUSED"] = mode
    L, m = spec["L"], spec["m"]
    M = L * m
    rnn = spec["rnn"]
    over = dict(use_recurrent_policy=True) if rnn else {}
    sh = G.Shapes(T=L, N=m, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                  discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"])
    d = G.make_buffers(sh, 61, inactive_p=0.2, unavailable_p=0.25 if sh.discrete else 0.0, rnn=rnn)
    rng = np.random.default_rng(8)
    obs = d.obs[0][:-1].reshape(M, -1)
    masks = d.masks[0][:-1].reshape(M, 1)
    h0 = d.rnn["actor"][0][0] if rnn else None
    act = d.actions[0].reshape(M, -1)
    avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
    active = d.active_masks[0][:-1].reshape(M, 1)
    adv = rng.standard_normal((M, 1)).astype(np.float32)
    factor = (1 + 0.2 * rng.standard_normal((M, 1))).astype(np.float32)
    actor, _, _ = G._mk_actor(sh, 17, **over)
    net = actor.actor
    lp, _, _ = actor.evaluate_actions(obs, h0, act, masks, avail, None)
    out = dict(logp=lp.clone())
    if rnn:
        out["gi_fwdonly"] = net.rnn_gi.clone()
        out["rnn_y_fwdonly"] = net.rnn_y.clone()
    old_logp = (lp.cpu().numpy() + 0.15).astype(np.float32)
    taps = []
    actor._grad_tap = lambda gr, sc: taps.append(gr.clone())
    actor.update((obs, h0, act, masks, active, old_logp, adv, avail, factor))
    torch.cuda.synchronize()
    for l in range(len(net.xh)):
        out[f"xh{l}"] = net.xh[l].clone()
        out[f"rstd{l}"] = net.rstd[l].clone()
        out[f"mask{l}"] = net.rmask[l].clone()
    out["x0n"] = net.x0n.clone()
    if rnn:
        out["gi"] = net.rnn_gi.clone()
        out["rnn_y"] = net.rnn_y.clone()
        for k, t in enumerate(net.rnn_dgate):
            out[f"dgate{k}"] = t.clone()
    for k, t in enumerate(net.dz):
        out[f"dzbuf{k}"] = t.clone()
    out["part"] = net.part.clone()
    out["dwp"] = net.dwp.clone()
    out["grad"] = taps[0]
    return out, net


for spec in G.TRUNK_SPECS[:1]:
    a, na = run("0", spec)
    b, nb = run("1", spec)
    print(spec["name"], "rows", spec["L"] * spec["m"], "part offs", na._part_offs, "n_wg", na.n_wg)
    for k in a:
        if k not in b or a[k].shape != b[k].shape:
            print(f"  {k}: shape {tuple(a[k].shape)} vs {tuple(b[k].shape) if k in b else None}")
            continue
        ne = (a[k] != b[k])
        if a[k].dtype.is_floating_point:
            ne &= ~(torch.isnan(a[k]) & torch.isnan(b[k]))
        n = int(ne.sum())
        msg = ""
        if n and a[k].dtype.is_floating_point:
            den = float(a[k].abs().max())
            msg = f" max|d|/max|a| {float((a[k].double() - b[k].double()).abs().max()) / max(den, 1e-30):.3e} first {int(ne.reshape(-1).nonzero()[0])}"
        print(f"  {k}: {n} of {a[k].numel()} differ{msg}")
