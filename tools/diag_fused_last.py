"""Debugging aid: the data of tests/gpu_checks.check_fused_vs_layered (3 x 128, obs 23), layer-by-layer vs last-layer-in-loss
launch: per-row comparison of dz_L (ATL image) and of the first-epoch log-probs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as G
from harl_amd.synthetic import Shapes, make_buffers

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
    _so = d.share_obs
    vp = rng.standard_normal(rows).astype(np.float32)
    _ret = (vp + rng.standard_normal(rows)).astype(np.float32)
    os.environ["HARL_FUSED_UPDATE"] = "0"
    actor.actor.fold()
    lp0 = torch.empty(rows, actor.actor.act_w, device=G.DEV)
    actor._logp_pass(obs, act, None, rows, lp0)
    old_logp = (lp0 + dev(0.1 * rng.standard_normal((rows, actor.actor.act_w)).astype(np.float32))).contiguous()
    got = {}
    H = 128
    for tag, mode in (("old", "0"), ("new", "hybrid")):
        os.environ["HARL_FUSED_UPDATE"] = mode
        actor.actor.invalidate_caches()
        lp = torch.zeros(rows, actor.actor.act_w, device=G.DEV)
        actor._forward_backward(obs, None, rows, act, None, old_logp, adv, None, factor, active, logp_out=lp)
        torch.cuda.synchronize()
        nsl = (rows + 31) // 32
        dz = actor.actor.dz[0][:nsl * 32 * H].clone().reshape(nsl, H // 8, 64, 4)
        got[tag] = (dz, lp.clone(), actor.actor.dwp.clone())
    dzo, dzn = got["old"][0], got["new"][0]
    diff = (dzo - dzn).abs()
    per_slab = diff.reshape(diff.shape[0], -1).max(1).values
    scale = float(dzo.abs().max())
    bad = (per_slab > 1e-5 * scale).nonzero().reshape(-1)
    print("slabs", diff.shape[0], "max|dz|", scale, "slabs with |diff| > 1e-5 max:", bad.numel(), bad[:40].tolist())
    if bad.numel():
        s0 = int(bad[0])
        dl = diff[s0].reshape(H // 8, 64, 4)
        per_lane = dl.permute(1, 0, 2).reshape(64, -1).max(1).values
        print("first bad slab", s0, "lanes with diff:", (per_lane > 1e-5 * scale).nonzero().reshape(-1).tolist())
        print(" old lane vals", dzo[s0, :2, int(per_lane.argmax())].cpu().numpy(), "new", dzn[s0, :2, int(per_lane.argmax())].cpu().numpy())
        # which grid position: slab = blockIdx * 8 + wave + k * grid*8
        g = min(256, (diff.shape[0] + 7) // 8)
        print(" block", (s0 % (g * 8)) // 8, "wave", s0 % 8, "iteration", s0 // (g * 8))
        row = s0 * 32 + int(per_lane.argmax()) % 32
        lo_, ln_ = got["old"][1][row].double(), got["new"][1][row].double()
        ol = old_logp[row].double()
        print(" row", row, "active", float(active[row]), "adv", float(adv[row]), "factor", float(factor[row]))
        print("  logp old path", lo_.cpu().numpy(), "\n  logp new path", ln_.cpu().numpy(), "\n  stored old_logp", ol.cpu().numpy())
        print("  imp old path %.9f new path %.9f lp0-based %.9f" % (float(torch.exp((lo_ - ol).sum())), float(torch.exp((ln_ - ol).sum())),
                                                                float(torch.exp((lp0[row].double() - ol).sum()))))
        import struct
        f32 = lambda v: struct.unpack("f", struct.pack("f", v))[0]
        po, pn = 1.0, 1.0
        for d_ in range(5):
            po = f32(po * f32(float(torch.exp((got["old"][1][row, d_] - old_logp[row, d_])))))
            pn = f32(pn * f32(float(torch.exp((got["new"][1][row, d_] - old_logp[row, d_])))))
        print("  fp32 product of per-dim ratios: old path %.9f new path %.9f" % (po, pn))
        print("bad slab iterations histogram:", np.bincount((bad.cpu().numpy() // (g * 8))), "waves:", np.bincount(bad.cpu().numpy() % 8, minlength=8))
    print("logp max diff", float((got["old"][1] - got["new"][1]).abs().max()))
    print("dwp vec rel", float((got["old"][2] - got["new"][2]).abs().max() / got["old"][2].abs().max()))

main()
