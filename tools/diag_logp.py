"""Where does a systematic 1e-5 in the policy loss come from?  For every agent of a golden case: the first-update policy loss
(factor = 1) evaluated in fp64 from (a) the GPU log-probs, (b) torch-fp32 log-probs, (c) fp64 log-probs."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests import gpu_checks as G
from tests.helpers import GoldenCase
from oracle import harl_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "mpe_box_h128"
case = GoldenCase(name)
torch.manual_seed(case.seed); np.random.seed(case.seed)
r = G.build_runner(case)
cb = r.critic_buffer
cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
T, N = case.shapes.T, case.shapes.N
B = T * N
adv = cb.advantages.reshape(B).double().cpu()
train, model, algo = case.reference_dicts()
cfg = O.PathConfig.from_reference_dicts(train, model, algo)
for a in range(case.shapes.A):
    buf = r.actor_buffer[a]
    act = r.actor[a]
    lp_gpu, _, _ = act.evaluate_actions(buf.flat("obs"), None, buf.flat("actions"), None,
                                        None if buf.available_actions is None else buf.flat("available_actions"), buf.flat("active_masks"))
    lp_gpu = lp_gpu.double().cpu()
    obs = buf.flat("obs").cpu().numpy(); ac = buf.flat("actions").cpu().numpy()
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        O.set_work_dtype(dt)
        try:
            o = O.OracleHAPPO({k: torch.from_numpy(v) for k, v in case.actor_sd[a].items()}, cfg)
            with torch.no_grad():
                lp, _, _ = o.evaluate_actions(obs, ac)
        finally:
            O.set_work_dtype(torch.float32)
        out[tag] = lp.double()
    olp = buf.flat("action_log_probs").double().cpu()
    am = buf.flat("active_masks").double().cpu().reshape(B)
    advm = adv[am > 0]
    advn = ((adv - advm.mean()) / (advm.std(unbiased=False) + 1e-5)).reshape(B, 1)
    def loss(lp):
        imp = torch.prod(torch.exp(lp - olp), dim=-1, keepdim=True)
        s = torch.min(imp * advn, torch.clamp(imp, 0.8, 1.2) * advn)
        return float((-(s.reshape(B) * am)).sum() / am.sum()), imp
    l64, imp64 = loss(out["f64"]); l32, _ = loss(out["f32"]); lg, _ = loss(lp_gpu)
    d32 = (out["f32"] - out["f64"]).abs(); dg = (lp_gpu - out["f64"]).abs()
    print(f"agent {a}: loss f64 {l64:.9g}  torch-f32 logp -> rel {abs(l32 - l64) / abs(l64):.2e}   gpu logp -> rel {abs(lg - l64) / abs(l64):.2e}")
    print(f"   |logp| max {out['f64'].abs().max():.3g}  abs err: torch-f32 max {d32.max():.2e} mean {d32.mean():.2e} | gpu max {dg.max():.2e} mean {dg.mean():.2e}"
          f"  signed mean: torch {float((out['f32'] - out['f64']).mean()):.2e} gpu {float((lp_gpu - out['f64']).mean()):.2e}")
    print(f"   imp: max {imp64.max():.3g} median {imp64.median():.3g}")
