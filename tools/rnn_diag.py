"""Per-tensor gradient / post-Adam parameter differences for one recurrent HAPPO update (diagnostic)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as G
from tests.gpu_checks import O, Shapes, make_buffers, actor_param_shapes, dev

spec = G.RNN_SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
L, m = spec["L"], spec["m"]; M = L * m
sh = Shapes(T=L, N=m, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
            discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"])
d = make_buffers(sh, 61, inactive_p=0.2, unavailable_p=0.25 if sh.discrete else 0.0, rnn=True)
actor, sd, args = G._mk_actor(sh, 17, use_recurrent_policy=True)
cfg = O.PathConfig.from_reference_dicts({}, args, args)
rng = np.random.default_rng(8)
obs = d.obs[0][:-1].reshape(M, -1); masks = d.masks[0][:-1].reshape(M, 1); h0 = d.rnn["actor"][0][0]
act = d.actions[0].reshape(M, -1)
avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
active = d.active_masks[0][:-1].reshape(M, 1)
oracle = O.OracleHAPPO({k: torch.from_numpy(v) for k, v in sd.items()}, cfg)
with torch.no_grad():
    lp, _, _ = oracle.evaluate_actions(obs, act, avail, None, h0, masks)
old_logp = (lp.numpy() + 0.15 * rng.standard_normal(lp.shape)).astype(np.float32)
adv = rng.standard_normal((M, 1)).astype(np.float32)
factor = (1 + 0.2 * rng.standard_normal((M, 1))).astype(np.float32)
p0 = oracle.net.flat().copy()
pl, ent, gn, imp, g = oracle.update((obs, act, active, old_logp, adv, avail, factor, h0, masks), keep_grad=True)
taps = []
actor._grad_tap = lambda gr, sc: taps.append((gr.clone(), sc))
actor.update((obs, h0, act, masks, active, old_logp, adv, avail, factor))
torch.cuda.synchronize()
gg = taps[0][0].cpu().numpy()
p1o, p1g = oracle.net.flat(), actor.actor.flat_param.cpu().numpy()
off = 0
print(f"{'tensor':40s} {'|g|max':>10s} {'g abs err':>10s} {'g rel':>9s} {'dp abs':>10s} {'n(|g|<1e-5)':>12s}")
for name, shp in actor_param_shapes(sh, True, True):
    n = int(np.prod(shp)); s = slice(off, off + n)
    ge = np.max(np.abs(gg[s] - g[s])); gm = np.max(np.abs(g[s]))
    print(f"{name:40s} {gm:10.3e} {ge:10.3e} {ge/max(gm,1e-30):9.2e} {np.max(np.abs(p1o[s]-p1g[s])):10.3e} {int(np.sum(np.abs(g[s])<1e-5)):12d}")
    off += n
