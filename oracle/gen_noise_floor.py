#!/usr/bin/env python3
"""How far do the reference's OWN fp32 results sit from exact arithmetic?  (test infrastructure, CPU only)

For every golden case the same seeded update is run once more through the oracle (oracle/harl_oracle.py, pinned to the
reference's golden vectors at 1e-6 by tests/test_oracle_golden.py) in FLOAT64 -- same buffers, same initial weights
(float32 values), same minibatch permutations -- and the per-update statistics, the train() infos and the final
parameters are written to tests/golden/noise/<case>.npz.  |golden_fp32 - fp64| / |fp64| is then the reference's own
rounding error on each figure.  Some of the logged figures are ill-conditioned (a policy loss is a masked mean of
advantage-normalised surrogates, i.e. a difference of nearly equal sums), so that error is far above 1e-5 for them
in the reference itself; tests/gpu_checks.py holds the HIP path to  max(1e-5, C x that error)  entry by entry instead
of one blanket tolerance.  A second yardstick is stored next to it: the same fp32 update with every parameter moved by ONE
ulp (random direction) at the start and again after every optimiser step, 32 independent runs -- `sens_*` = how far the
reference's own figures move under the smallest perturbation fp32 can express, injected at the rate at which two correct fp32
implementations differ (once per step).  Sequential Adam steps amplify such rounding-level differences by one to two orders
of magnitude by the end of train().

    python oracle/gen_noise_floor.py            # all cases (about a minute)
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import harl_oracle as O  # noqa: E402
from tests.helpers import (ACT_CASES, ALL_CASES, GOLDEN_DIR, MAPPO_CASES, MD_CASES, RNN128_CASES, RNN_CASES, TRPO_CASES, TRPO_RNN_CASES,  # noqa: E402
                           GoldenCase)
from tests.test_oracle_golden import build_oracle  # noqa: E402


def _perturb_one_ulp(nets, seed, gen=None) -> None:
    """Every parameter moved to a neighbouring float32 (up or down at random): the smallest change fp32 can express."""
    g = gen if gen is not None else torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in nets:
            for v in net.p.values():
                up = torch.rand(v.shape, generator=g) < 0.5
                v.copy_(torch.where(up, torch.nextafter(v, torch.full_like(v, float("inf"))),
                                    torch.nextafter(v, torch.full_like(v, float("-inf")))))


def _make_grad_hook(gen):
    """g_fp32 = g_exact + e  ->  g_exact + s * e  with a random sign per element (see the module docstring)."""
    stash = {}

    def hook(phase, obj, sample, vn):
        if O.WORK_DTYPE != torch.float32:
            return
        if phase == "pre":  # the exact gradient at the current parameters: a float64 clone of this network, same minibatch
            saved = (O.GRAD_HOOK, O.STEP_HOOK)
            O.GRAD_HOOK = O.STEP_HOOK = None
            O.set_work_dtype(torch.float64)
            try:
                clone = type(obj)({k: v.detach() for k, v in obj.net.p.items()}, obj.cfg)
                if isinstance(obj, O.OracleVCritic):
                    stash[id(obj)] = clone.update(sample, copy.deepcopy(vn), keep_grad=True)[-1]
                else:
                    stash[id(obj)] = clone.update(sample, keep_grad=True)[-1]
            finally:
                O.set_work_dtype(torch.float32)
                O.GRAD_HOOK, O.STEP_HOOK = saved
            return
        g64, off = torch.from_numpy(np.asarray(stash.pop(id(obj)), dtype=np.float64)), 0
        for p in obj.net.params():
            n = p.numel()
            e = p.grad.reshape(-1).double() - g64[off:off + n]
            off += n
            s = (torch.rand(n, generator=gen) < 0.5).double() * 2.0 - 1.0
            p.grad.add_(((s - 1.0) * e).to(p.grad.dtype).reshape(p.shape))

    return hook


def run_case(name: str, dtype=torch.float64, perturb_seed=None) -> dict:
    case = GoldenCase(name)
    torch.set_num_threads(1)
    torch.manual_seed(case.seed)
    np.random.seed(case.seed)
    O.set_work_dtype(dtype)
    try:
        cfg, actors, critic, abufs, cbuf, vn = build_oracle(case)
        if perturb_seed is not None:
            uniq = []
            for a_ in actors:
                if not any(a_ is u for u in uniq):
                    uniq.append(a_)
            _perturb_one_ulp([a_ if hasattr(a_, "p") else a_.net for a_ in uniq] + [critic.net], perturb_seed)
            # ... and again after EVERY optimiser step: another fp32 implementation of the same update differs from the
            # reference by rounding in every operation of every step, not by one kick at the start
            hook_gen = torch.Generator().manual_seed(perturb_seed + 77)
            O.STEP_HOOK = lambda net: _perturb_one_ulp([net], None, hook_gen)
            if case.algo_name != "hatrpo":  # (HATRPO's CG / line-search update has no single gradient step to perturb)
                O.GRAD_HOOK = _make_grad_hook(torch.Generator().manual_seed(perturb_seed + 991))
        torch.manual_seed(case.seed + 12345)
        cbuf.compute_returns(cbuf.value_preds[-1].copy(), vn, cfg)
        if case.algo_name == "mappo":
            infos, cinfo, extra = O.ma_train(actors, critic, abufs, cbuf, vn, cfg, share_param=case.share_param)
        else:
            infos, cinfo, extra = O.ha_train(actors, critic, abufs, cbuf, vn, cfg)
    finally:
        O.set_work_dtype(torch.float32)
        O.STEP_HOOK = None
        O.GRAD_HOOK = None
    out = {}
    if case.algo_name == "hatrpo":
        out["actor_infos"] = np.array([[float(i["kl"]), float(i["loss_improve"]), float(np.asarray(i["expected_improve"]).reshape(-1)[0]),
                                        float(i["dist_entropy"]), float(i["ratio"])] for i in infos], dtype=np.float64)
    else:
        order = [0] if getattr(case, "share_param", False) else extra["agent_order"]
        out["actor_trace"] = np.array([[a, t["policy_loss"], t["dist_entropy"], t["grad_norm"], t["ratio"]]
                                       for a in order for t in actors[a].trace], dtype=np.float64)
        out["actor_infos"] = np.array([[i["policy_loss"], i["dist_entropy"], i["actor_grad_norm"], i["ratio"]] for i in infos],
                                      dtype=np.float64)
    out["critic_trace"] = np.array([[t["value_loss"], t["grad_norm"]] for t in critic.trace], dtype=np.float64)
    out["critic_info"] = np.array([cinfo["value_loss"], cinfo["critic_grad_norm"]], dtype=np.float64)
    for a in range(case.shapes.A):
        flat = actors[a].flat().numpy() if case.algo_name == "hatrpo" else actors[a].net.flat()
        out[f"actor_final_{a}"] = np.asarray(flat, dtype=np.float64)
    out["critic_final"] = np.asarray(critic.net.flat(), dtype=np.float64)
    return out


N_PERT = 32


def main():
    os.makedirs(os.path.join(GOLDEN_DIR, "noise"), exist_ok=True)
    names = sys.argv[1:] or (ALL_CASES + TRPO_CASES + TRPO_RNN_CASES + RNN_CASES + MAPPO_CASES + MD_CASES + RNN128_CASES + ACT_CASES)
    for name in names:
        res = run_case(name)
        z = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
        # keep the fixtures small: final parameters are stored as the fp32 rounding of the fp64 run
        small = {k: (v.astype(np.float32) if k.startswith(("actor_final", "critic_final")) else v) for k, v in res.items()}
        # sensitivity of the reference's fp32 run: the same update from initial weights moved by ONE ulp (N_PERT random
        # directions); `sens_<key>` = entry-wise max |perturbed - unperturbed| / |unperturbed| (final parameters: inf-norm)
        base = run_case(name, dtype=torch.float32)
        sens = {}
        for ps in range(N_PERT):
            pr = run_case(name, dtype=torch.float32, perturb_seed=1000 + ps)
            for k, v in pr.items():
                if k.startswith(("actor_final", "critic_final")):
                    e = np.array(np.max(np.abs(v - base[k])) / (np.max(np.abs(base[k])) + 1e-30))
                else:
                    e = np.abs(v - base[k]) / (np.abs(base[k]) + 1e-12)
                    if k == "actor_trace":
                        e[:, 0] = 0.0  # agent id column
                sens[k] = np.maximum(sens[k], e) if k in sens else e
        small.update({f"sens_{k}": v for k, v in sens.items()})
        np.savez_compressed(os.path.join(GOLDEN_DIR, "noise", f"{name}.npz"), **small)
        rel = lambda a, b: float(np.max(np.abs(a - b) / (np.abs(b) + 1e-12)))  # noqa: E731
        msg = f"{name:28s} infos {rel(z['actor_infos'], res['actor_infos']):.2e}  critic {rel(z['critic_info'], res['critic_info']):.2e}"
        if "actor_trace" in res and "actor_trace" in z.files:
            msg += f"  actor_trace {rel(z['actor_trace'][:, 1:], res['actor_trace'][:, 1:]):.2e}  critic_trace {rel(z['critic_trace'], res['critic_trace']):.2e}"
            msg += f"  | sensitivity: actor_trace {float(sens['actor_trace'].max()):.2e} critic_trace {float(sens['critic_trace'].max()):.2e}"
        msg += f" infos {float(sens['actor_infos'].max()):.2e} final {max(float(v) for k, v in sens.items() if 'final' in k):.2e}"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
