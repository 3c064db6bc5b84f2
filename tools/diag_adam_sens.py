"""Which gradient elements drive the parameter divergence?  Golden case, the agent trained first (factor = 1): at every
optimiser step the HIP gradient and a torch-fp32 gradient, both AT THE HIP PATH'S CURRENT PARAMETERS, against the fp64
gradient there -- per tensor, plain and weighted by Adam's first-step sensitivity  d step / d g = lr * eps / (|g| + eps)^2."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import gpu_checks as G
from tests.helpers import GoldenCase
from tests.test_oracle_golden import build_oracle
from oracle import harl_oracle as O
from harl_amd.synthetic import actor_param_shapes

name = sys.argv[1] if len(sys.argv) > 1 else "mpe_box_h128"
case = GoldenCase(name)
torch.manual_seed(case.seed); np.random.seed(case.seed)
r = G.build_runner(case)
torch.manual_seed(case.seed + 12345)
cb = r.critic_buffer
cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
r.prep_training()
order_probe = torch.get_rng_state()
first = int(torch.randperm(case.shapes.A)[0]) if not case.algo["fixed_order"] else 0
torch.set_rng_state(order_probe)
act = r.actor[first]
taps = []
act._grad_tap = lambda gr, sc: taps.append((gr.double().cpu().numpy().copy(), {k: v.detach().clone().cpu() for k, v in act.actor.state_dict().items()}))
r.train()
torch.cuda.synchronize()
print("first agent", first, "updates", len(taps))
T, N = case.shapes.T, case.shapes.N
B = T * N
params = [{k: torch.from_numpy(v) for k, v in case.actor_sd[first].items()}] + [t[1] for t in taps]
adv_raw = cb.advantages.cpu().numpy()
lr, eps = case.model["lr"], case.model["opti_eps"]
for k, (gg, _) in enumerate(taps):
    res = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        O.set_work_dtype(dt)
        try:
            cfg, actors, critic, abufs, cbuf, vn = build_oracle(case)
            o = O.OracleHAPPO(params[k], cfg)
            buf = abufs[first]
            advn = O.normalize_advantages(adv_raw.astype(O._np_work()), buf.active_masks[:-1])
            f = lambda v: v.reshape(B, -1)
            sample = (f(buf.obs[:-1]), f(buf.actions), f(buf.active_masks[:-1]), f(buf.action_log_probs), f(advn), None,
                      np.ones((B, 1), dtype=O._np_work()))
            pl, ent, gn, imp, g = o.update(sample, keep_grad=True)
        finally:
            O.set_work_dtype(torch.float32)
        res[tag] = np.asarray(g, dtype=np.float64)
        res[tag + "_p"] = {kk: v.detach().double().numpy().copy() for kk, v in o.net.p.items()}
    g64, g32 = res["f64"], res["f32"]
    off = 0
    print(f"update {k}")
    for nm, shp in actor_param_shapes(case.shapes, case.model["use_feature_normalization"]):
        n = int(np.prod(shp)); s = slice(off, off + n); off += n
        w = lr * eps / (np.abs(g64[s]) + eps) ** 2
        rms = lambda x: float(np.sqrt(np.mean(x * x)))
        print(f"   {nm:32s} |g|max {np.abs(g64[s]).max():.1e} med {np.median(np.abs(g64[s])):.1e}  err rms/max|g|: hip {rms(gg[s]-g64[s])/np.abs(g64[s]).max():.1e} t32 {rms(g32[s]-g64[s])/np.abs(g64[s]).max():.1e}"
              f"   implied step err rms: hip {rms(w*(gg[s]-g64[s])):.1e} t32 {rms(w*(g32[s]-g64[s])):.1e}")
    if k == 0:
        print("   parameters after the FIRST step (Adam state empty): rms distance from the fp64 step")
        for nm in res["f64_p"]:
            rms = lambda x: float(np.sqrt(np.mean(x * x)))
            th = taps[0][1][nm].double().numpy()
            print(f"      {nm:32s} hip {rms(th - res['f64_p'][nm]):.2e}  t32 {rms(res['f32_p'][nm] - res['f64_p'][nm]):.2e}   ulp(|theta|max)/2 {np.abs(th).max() * 3e-8:.1e}")
    if k >= 1:
        break
