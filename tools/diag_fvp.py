"""Per-tensor error of the HIP Fisher-vector product against the oracle in float64 (and the fp32 oracle's own), at the
Humanoid-17x1 shape for several batch sizes:  python tools/diag_fvp.py   (MI355X box; oracle = test infrastructure)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harl_oracle as O  # noqa: E402
from tests import gpu_checks as G  # noqa: E402
from tests.helpers import SyntheticCase  # noqa: E402
from tests.test_oracle_golden import build_oracle  # noqa: E402
from harl_amd.synthetic import Shapes  # noqa: E402


def main():
    spec = G.BASELINE_SHAPES["humanoid17"]
    for N in [int(x) for x in os.environ.get('DIAG_N', '2,10,40').split(',')]:
        shp = dict(spec["shapes"], N=N, A=1)
        case = SyntheticCase("diag", Shapes(**shp), spec["seed"], algo_name="hatrpo", overrides=spec.get("overrides"))
        masked = G._mask_relu_kinks(case, replace=os.environ.get('DIAG_REPLACE', '1') == '1')
        sh = case.shapes
        M = sh.T * sh.N
        torch.manual_seed(case.seed)
        np.random.seed(case.seed)
        r = G.build_runner(case)
        r.prep_training()
        rng = np.random.default_rng(5)
        obs = case.data.obs[0][:-1].reshape(M, -1)
        names = list(case.actor_sd[0].keys())
        sizes = [int(np.prod(case.actor_sd[0][k].shape)) for k in names]
        v = rng.standard_normal(sum(sizes)).astype(np.float32)
        only = os.environ.get("DIAG_ONLY")  # restrict the direction to the parameter tensors whose name contains this
        if only:
            off = 0
            for k, n in zip(names, sizes):
                if only not in k:
                    v[off:off + n] = 0.0
                off += n
        fv = {}
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            O.set_work_dtype(dt)
            try:
                _, actors, _, _, _, _ = build_oracle(case)
                fv[tag] = actors[0].fvp(torch.from_numpy(obs).to(dt), None, torch.from_numpy(v).to(dt)).numpy().astype(np.float64)
            finally:
                O.set_work_dtype(torch.float32)
        actor = r.actor[0]
        actor.actor.fold()
        d_obs = G.dev(obs)
        act = case.data.actions[0].reshape(M, -1)
        adv = rng.standard_normal((M, 1)).astype(np.float32)
        olp = case.data.action_log_probs[0].reshape(M, -1)
        actm = case.data.active_masks[0][:-1].reshape(M, 1)
        _, gg = actor._surrogate(d_obs, M, G.dev(act), None, G.dev(olp), G.dev(adv.reshape(M)), None,
                                 G.dev(np.ones(M, dtype=np.float32)), G.dev(actm.reshape(M)), want_grad=True)
        gg = gg.cpu().numpy().astype(np.float64)
        og = {}
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            O.set_work_dtype(dt)
            try:
                _, actors, _, _, _, _ = build_oracle(case)
                t = lambda x: torch.from_numpy(x).to(dt)  # noqa: E731
                loss, _, _ = actors[0].surrogate(t(obs), t(act), None, t(actm), t(olp), t(adv), t(np.ones((M, 1), np.float32)))
                og[tag] = torch.cat([x.reshape(-1) for x in torch.autograd.grad(loss, actors[0].params())]).numpy().astype(np.float64)
            finally:
                O.set_work_dtype(torch.float32)
        off = 0
        print("   surrogate GRADIENT per tensor:")
        for k, n in zip(names, sizes):
            a, b, c = gg[off:off + n], og["f32"][off:off + n], og["f64"][off:off + n]
            s_ = np.max(np.abs(c)) + 1e-300
            if "weight" in k:
                print(f"      {k:40s} hip {np.max(np.abs(a - c)) / s_:.2e}  ref32 {np.max(np.abs(b - c)) / s_:.2e}")
            off += n
        g = actor._fvp(d_obs, M, M, None, G.dev(v)).cpu().numpy().astype(np.float64)
        print(f"## M = {M} ({masked} kink-adjacent samples masked); whole vector: hip {np.max(np.abs(g - fv['f64'])) / np.max(np.abs(fv['f64'])):.2e} "
              f"ref32 {np.max(np.abs(fv['f32'] - fv['f64'])) / np.max(np.abs(fv['f64'])):.2e}")
        off = 0
        for k, n in zip(names, sizes):
            a, b, c = g[off:off + n], fv["f32"][off:off + n], fv["f64"][off:off + n]
            s = np.max(np.abs(c)) + 1e-300
            print(f"   {k:40s} n={n:6d} |f64|max {s:.2e}  hip {np.max(np.abs(a - c)) / s:.2e}  ref32 {np.max(np.abs(b - c)) / s:.2e}")
            off += n
        del r


if __name__ == "__main__":
    main()
