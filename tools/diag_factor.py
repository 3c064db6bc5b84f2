"""Per-agent sequential-update factor of the HIP path vs the oracle in fp32 and fp64 (golden case)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import gpu_checks as G
from tests.helpers import GoldenCase
from oracle import harl_oracle as O
from tests.test_oracle_golden import build_oracle

name = sys.argv[1] if len(sys.argv) > 1 else "mpe_box_h128"
case = GoldenCase(name)
runs = {}
for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
    O.set_work_dtype(dt)
    try:
        torch.manual_seed(case.seed); np.random.seed(case.seed)
        cfg, actors, critic, abufs, cbuf, vn = build_oracle(case)
        torch.manual_seed(case.seed + 12345)
        cbuf.compute_returns(cbuf.value_preds[-1].copy(), vn, cfg)
        infos, cinfo, extra = O.ha_train(actors, critic, abufs, cbuf, vn, cfg)
    finally:
        O.set_work_dtype(torch.float32)
    runs[tag] = extra
order = runs["f32"]["agent_order"]
print("order", order)
torch.manual_seed(case.seed); np.random.seed(case.seed)
r = G.build_runner(case)
torch.manual_seed(case.seed + 12345)
cb = r.critic_buffer
cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
r.prep_training()
r.train()
torch.cuda.synchronize()
# the factor an agent trained with = product over the agents before it; buffer k holds the factor BEFORE agent k's update
for pos, a in enumerate(order):
    f_gpu = r.actor_buffer[a].factor.double().cpu().numpy().reshape(-1)
    f32 = (np.ones_like(f_gpu) if pos == 0 else runs["f32"]["factors"][pos - 1].reshape(-1).astype(np.float64))
    f64 = (np.ones_like(f_gpu) if pos == 0 else runs["f64"]["factors"][pos - 1].reshape(-1))
    rg, r3 = f_gpu / f64 - 1, f32 / f64 - 1
    print(f"pos {pos} agent {a}: factor range [{f64.min():.3g}, {f64.max():.3g}]  gpu/f64-1: mean {rg.mean():+.2e} rms {np.sqrt((rg**2).mean()):.2e} max {np.abs(rg).max():.2e}"
          f" | torch32/f64-1: mean {r3.mean():+.2e} rms {np.sqrt((r3**2).mean()):.2e} max {np.abs(r3).max():.2e}")

# ---- evaluation error alone: exact (fp64) factor OF THE PARAMETERS EACH PATH ENDED WITH vs the factor that path produced
from tests.test_oracle_golden import build_oracle as _bo
O.set_work_dtype(torch.float64)
cfg64, actors64, _, abufs64, _, _ = _bo(case)
T, N = case.shapes.T, case.shapes.N
def exact_ratio(a, sd_final):
    buf = abufs64[a]
    obs, act = buf.obs[:-1].reshape(T * N, -1), buf.actions.reshape(T * N, -1)
    o0 = O.OracleHAPPO({k: torch.from_numpy(v) for k, v in case.actor_sd[a].items()}, cfg64)
    o1 = O.OracleHAPPO({k: v.double().cpu() for k, v in sd_final.items()}, cfg64)
    with torch.no_grad():
        l0, _, _ = o0.evaluate_actions(obs, act)
        l1, _, _ = o1.evaluate_actions(obs, act)
    return torch.prod(torch.exp(l1 - l0), dim=-1).numpy().reshape(-1)
prev_gpu = np.ones(T * N); prev_32 = np.ones(T * N)
O.set_work_dtype(torch.float32)
torch.manual_seed(case.seed); np.random.seed(case.seed)
cfg, actors, critic, abufs, cbuf, vn = build_oracle(case)
torch.manual_seed(case.seed + 12345)
cbuf.compute_returns(cbuf.value_preds[-1].copy(), vn, cfg)
infos, cinfo, extra = O.ha_train(actors, critic, abufs, cbuf, vn, cfg)
O.set_work_dtype(torch.float64)
for pos, a in enumerate(order):
    sd_gpu = {k: v.detach().clone() for k, v in r.actor[a].actor.state_dict().items()}
    sd_32 = {k: v.detach().clone() for k, v in actors[a].net.p.items()}
    ex_gpu = exact_ratio(a, sd_gpu)
    nxt = order[pos + 1] if pos + 1 < len(order) else None
    if nxt is not None:
        f_gpu = r.actor_buffer[nxt].factor.double().cpu().numpy().reshape(-1)
        e = f_gpu / (prev_gpu * ex_gpu) - 1
        print(f"agent {a}: HIP factor vs exact ratio of ITS OWN parameters: mean {e.mean():+.2e} rms {np.sqrt((e**2).mean()):.2e} max {np.abs(e).max():.2e}")
        prev_gpu = f_gpu
        if sd_32 is not None:
            ex32 = exact_ratio(a, sd_32)
            f32 = extra["factors"][pos].reshape(-1).astype(np.float64)
            e = f32 / (prev_32 * ex32) - 1
            print(f"          torch-fp32 factor vs exact ratio of ITS OWN parameters: mean {e.mean():+.2e} rms {np.sqrt((e**2).mean()):.2e} max {np.abs(e).max():.2e}")
            prev_32 = f32
O.set_work_dtype(torch.float32)

# ---- parameter divergence per tensor for the agent trained first (no upstream factor error)
a = order[0]
O.set_work_dtype(torch.float64)
torch.manual_seed(case.seed); np.random.seed(case.seed)
cfg6, actors6, critic6, abufs6, cbuf6, vn6 = build_oracle(case)
torch.manual_seed(case.seed + 12345)
cbuf6.compute_returns(cbuf6.value_preds[-1].copy(), vn6, cfg6)
O.ha_train(actors6, critic6, abufs6, cbuf6, vn6, cfg6)
O.set_work_dtype(torch.float32)
sd_gpu = r.actor[a].actor.state_dict()
print(f"agent {a} (first in order): per tensor  |theta - theta_f64|_inf / |theta_final - theta_init|_inf")
for k, v0 in case.actor_sd[a].items():
    t64 = actors6[a].net.p[k].detach().double().numpy()
    tg = sd_gpu[k].double().cpu().numpy()
    t32 = actors[a].net.p[k].detach().double().numpy()
    upd = np.abs(t64 - v0.astype(np.float64)).max()
    rms = lambda x: float(np.sqrt(np.mean(x * x)))
    print(f"   {k:34s} update {upd:.2e}  hip {np.abs(tg - t64).max() / upd:.2e}  torch32 {np.abs(t32 - t64).max() / upd:.2e}   (rms hip {rms(tg - t64):.2e} torch32 {rms(t32 - t64):.2e}; hip-torch32 {rms(tg - t32):.2e})")
