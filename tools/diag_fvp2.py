"""Determinism / state-carry-over probe of the HATRPO kernels at the Humanoid shape (M = 8000)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as G  # noqa: E402
from tests.helpers import SyntheticCase  # noqa: E402
from harl_amd.synthetic import Shapes  # noqa: E402


def main():
    spec = G.BASELINE_SHAPES["humanoid17"]
    N = int(os.environ.get("DIAG_N", "40"))
    case = SyntheticCase("diag", Shapes(**dict(spec["shapes"], N=N, A=1)), spec["seed"], algo_name="hatrpo", overrides=spec.get("overrides"))
    sh = case.shapes
    M = sh.T * sh.N
    torch.manual_seed(case.seed)
    np.random.seed(case.seed)
    r = G.build_runner(case)
    r.prep_training()
    rng = np.random.default_rng(5)
    actor = r.actor[0]
    net = actor.actor
    net.fold()
    obs = G.dev(case.data.obs[0][:-1].reshape(M, -1))
    act = G.dev(case.data.actions[0].reshape(M, -1))
    olp = G.dev(case.data.action_log_probs[0].reshape(M, -1))
    adv = G.dev(rng.standard_normal(M).astype(np.float32))
    one = G.dev(np.ones(M, dtype=np.float32))
    actm = G.dev(case.data.active_masks[0][:-1].reshape(M))
    names = [k for k, _ in net.named_parameters()]
    sizes = [p.numel() for _, p in net.named_parameters()]
    v = torch.zeros(sum(sizes), device=obs.device)
    off = 0
    for k, n in zip(names, sizes):
        if "fc_mean" in k:
            v[off:off + n] = torch.randn(n, device=obs.device)
        off += n

    def show(tag, a, b):
        off = 0
        out = []
        for k, n in zip(names, sizes):
            d = (a[off:off + n] - b[off:off + n]).abs().max().item()
            s = b[off:off + n].abs().max().item() + 1e-30
            if d > 0:
                out.append(f"{k}: {d / s:.1e}")
            off += n
        print(tag, "IDENTICAL" if not out else " | ".join(out))

    _, g1 = actor._surrogate(obs, M, act, None, olp, adv, None, one, actm, want_grad=True)
    _, g2 = actor._surrogate(obs, M, act, None, olp, adv, None, one, actm, want_grad=True)
    show("grad call 2 vs call 1:", g2, g1)
    f1 = actor._fvp(obs, M, M, None, v)
    f2 = actor._fvp(obs, M, M, None, v)
    show("fvp call 2 vs call 1:", f2, f1)
    _, g3 = actor._surrogate(obs, M, act, None, olp, adv, None, one, actm, want_grad=True)
    show("grad after the fvps vs call 1:", g3, g1)
    f3 = actor._fvp(obs, M, M, None, v)
    show("fvp after another surrogate vs call 1:", f3, f1)
    # the backward alone, fed with the gradient path's head kernel but through the FVP's separate head weight-gradient pass
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
