"""Per-block breakdown of check_fused_vs_layered's folded-gradient comparison (debugging aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as G
from harl_amd.synthetic import Shapes, make_buffers
from harl_amd._lib import call, ptr, stream
from harl_amd import _lib

def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * 8 * 256 * 2 + 7 * 32 + 3
    hidden = [128, 128, 128]
    sh = Shapes(T=rows, N=1, A=1, obs_dim=23, share_obs_dim=54, act_dim=5, discrete=False, hidden_sizes=hidden)
    d = make_buffers(sh, 3)
    actor, _, _ = G._mk_actor(sh, 1)
    dev = G.dev
    obs = dev(d.obs[0][:-1].reshape(rows, -1)); act = dev(d.actions[0].reshape(rows, -1))
    rng = np.random.default_rng(0)
    adv = dev(rng.standard_normal(rows).astype(np.float32)); factor = dev((1 + 0.1 * rng.standard_normal(rows)).astype(np.float32))
    active = dev((rng.random(rows) > 0.1).astype(np.float32))
    os.environ["HARL_FUSED_UPDATE"] = "0"
    actor.actor.fold()
    lp0 = torch.empty(rows, actor.actor.act_w, device=G.DEV)
    actor._logp_pass(obs, act, None, rows, lp0)
    old_logp = (lp0 + dev(0.1 * rng.standard_normal((rows, actor.actor.act_w)).astype(np.float32))).contiguous()
    got = {}
    for tag, mode in (("old", "0"), ("new", "hybrid")):
        os.environ["HARL_FUSED_UPDATE"] = mode
        actor.actor.invalidate_caches()
        lp = torch.zeros(rows, actor.actor.act_w, device=G.DEV)
        nblk = actor._forward_backward(obs, None, rows, act, None, old_logp, adv, None, factor, active, logp_out=lp)
        sc = torch.zeros(_lib.PS_STRIDE, dtype=torch.float64, device=G.DEV)
        call("harl_reduce_scalars", ptr(actor.actor.part_scalars), nblk, ptr(sc), stream())
        got[tag] = (actor.actor.dwp.clone(), sc.clone(), lp.clone())
    torch.cuda.synchronize()
    net = actor.actor
    offs = sorted(set([0, net.dwp.numel()] + [int(o) for o in net._dwp_offs]))
    for a_, b_ in zip(offs[:-1], offs[1:]):
        o, n = got["old"][0][a_:b_], got["new"][0][a_:b_]
        print(f"block [{a_},{b_}): max|old| {float(o.abs().max()):.3e} max|new-old| {float((n-o).abs().max()):.3e} at {int((n-o).abs().argmax())}")
    print("scalars old", got["old"][1][:14].cpu().numpy())
    print("scalars new", got["new"][1][:14].cpu().numpy())
    print("logp diff", float((got["old"][2] - got["new"][2]).abs().max()))
    # head block detail: rows of dW_head
    a_, b_ = offs[-2], offs[-1]
    o, n = got["old"][0][a_:b_], got["new"][0][a_:b_]
    H = hidden[-1]
    dW_o, dW_n = o[:32 * H].reshape(32, H), n[:32 * H].reshape(32, H)
    print("head dW row max diff", (dW_n - dW_o).abs().max(1).values[:8].cpu().numpy(), "row max", dW_o.abs().max(1).values[:8].cpu().numpy())
    print("head db old", o[32 * H:32 * H + 8].cpu().numpy(), "new", n[32 * H:32 * H + 8].cpu().numpy())

main()
