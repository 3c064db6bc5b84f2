"""Run compute() + train() at BASELINE-like sizes for the other configs of BASELINE.json (not bench lines): checks that
the large shapes work and prints ms per update.  python tools/scale_smoke.py [name ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from harl_amd.runner import OnPolicyHARunner
from harl_amd.synthetic import Shapes, make_buffers

class Box:
    def __init__(self, shape): self.shape = shape
class Discrete:
    def __init__(self, n): self.n = n

CASES = {
    "halfcheetah6x1_happo": dict(algo="happo", T=200, N=4096, A=6, obs=23, sobs=17, act=1, disc=False, hidden=[128, 128, 128]),
    "smac3s5z_gru_happo": dict(algo="happo", T=160, N=512, A=8, obs=128, sobs=216, act=14, disc=True, hidden=[64, 64, 64],
                               rnn=True, L=10),
    "humanoid17x1_hatrpo": dict(algo="hatrpo", T=200, N=1024, A=17, obs=393, sobs=376, act=1, disc=False, hidden=[128, 128, 128]),
    "mpe_disc_mb4": dict(algo="happo", T=200, N=4096, A=3, obs=18, sobs=54, act=5, disc=True, hidden=[128, 128], mb=4),
}

def run(name, c):
    dev = torch.device("cuda:0")
    args = bench.algo_args(c["N"], c["T"])
    args["model"]["hidden_sizes"] = c["hidden"]
    args["model"]["use_recurrent_policy"] = bool(c.get("rnn"))
    args["model"]["data_chunk_length"] = c.get("L", 10)
    args["algo"]["actor_num_mini_batch"] = args["algo"]["critic_num_mini_batch"] = c.get("mb", 1)
    if c["algo"] == "hatrpo":
        args["algo"].update(kl_threshold=0.01, ls_step=10, accept_ratio=0.5, backtrack_coeff=0.8)
    torch.manual_seed(1); np.random.seed(1)
    space = Discrete(c["act"]) if c["disc"] else Box((c["act"],))
    r = OnPolicyHARunner(dict(algo=c["algo"]), args, dict(state_type="EP"), obs_spaces=[Box((c["obs"],))] * c["A"],
                         share_obs_space=Box((c["sobs"],)), act_spaces=[space] * c["A"], device=dev)
    sh = Shapes(T=c["T"], N=c["N"], A=c["A"], obs_dim=c["obs"], share_obs_dim=c["sobs"], act_dim=c["act"], discrete=c["disc"],
                hidden_sizes=c["hidden"])
    d = make_buffers(sh, seed=3, rnn=bool(c.get("rnn")))
    up = lambda x: torch.from_numpy(x).to(dev)
    for a in range(c["A"]):
        b = r.actor_buffer[a]
        b.obs.copy_(up(d.obs[a])); b.actions.copy_(up(d.actions[a])); b.masks.copy_(up(d.masks[a]))
        b.active_masks.copy_(up(d.active_masks[a]))
        if c["disc"]: b.available_actions.copy_(up(d.available_actions[a]))
        if d.rnn is not None: b.rnn_states.copy_(up(d.rnn["actor"][a]))
        kw = dict(rnn_states=b.rnn_states[0], masks=b.flat("masks")) if c.get("rnn") else {}
        lp = torch.empty(c["T"] * c["N"], r.actor[a].actor.act_w, device=dev)
        r.actor[a].actor.fold()
        r.actor[a]._logp_pass(b.flat("obs"), b.flat("actions"),
                              None if b.available_actions is None else b.flat("available_actions"), c["T"] * c["N"], lp, **kw)
        b.action_log_probs.copy_((lp + 0.05 * torch.randn_like(lp)).reshape(b.action_log_probs.shape))
    cb = r.critic_buffer
    cb.share_obs.copy_(up(d.share_obs)); cb.rewards.copy_(up(d.rewards)); cb.value_preds.copy_(up(d.value_preds))
    cb.masks.copy_(up(d.critic_masks)); cb.bad_masks.copy_(up(d.bad_masks))
    if d.rnn is not None: cb.rnn_states_critic.copy_(up(d.rnn["critic"]))
    r.prep_training()
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.compute(); infos, cinfo = r.train()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    if os.environ.get("HARL_KTIMING"):
        from harl_amd import _lib
        _lib.enable_kernel_timing(True)
        r.compute(); r.train()
        kt = _lib.collect_kernel_timing(); _lib.enable_kernel_timing(False)
        for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"]):
            print(f"    {k:20s} n={v['n']:5d} avg={v['avg_ms']*1e3:9.1f} us total={v['total_ms']:8.2f} ms")
    ok = all(np.isfinite(list(i.values())).all() for i in infos) and np.isfinite(list(cinfo.values())).all()
    tr = c["T"] * c["N"]
    print(f"{name:24s} T={c['T']} N={c['N']} A={c['A']} obs={c['obs']} hidden={c['hidden']}: {min(ts)*1e3:9.1f} ms/update "
          f"= {tr/min(ts)/1e6:7.3f} M transitions/s  finite={ok}  first actor info={ {k: round(v, 5) for k, v in infos[0].items()} }", flush=True)
    del r; torch.cuda.empty_cache()

if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CASES)):
        run(n, CASES[n])
