"""Per-launch time of the optimiser-step kernels against the batch size (fixed cost vs per-slab cost), hybrid and layer modes:

    python tools/diag_rows_sweep.py            (on the MI355X box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from harl_amd import _lib  # noqa: E402
from tests import gpu_checks as G  # noqa: E402
from harl_amd.synthetic import Shapes, make_buffers  # noqa: E402


def run(rows, mode, reps=4):
    os.environ["HARL_FUSED_UPDATE"] = mode
    sh = Shapes(T=rows, N=1, A=1, obs_dim=18, share_obs_dim=54, act_dim=5, discrete=False, hidden_sizes=[128, 128])
    d = make_buffers(sh, 3)
    actor, _, _ = G._mk_actor(sh, 1)
    dev = G.dev
    obs = dev(d.obs[0][:-1].reshape(rows, -1))
    act = dev(d.actions[0].reshape(rows, -1))
    rng = np.random.default_rng(0)
    adv = dev(rng.standard_normal(rows).astype(np.float32))
    factor = dev((1 + 0.1 * rng.standard_normal(rows)).astype(np.float32))
    active = dev(np.ones(rows, dtype=np.float32))
    old_logp = dev((-1.0 + 0.1 * rng.standard_normal((rows, 5))).astype(np.float32))
    actor.actor.fold()
    for _ in range(2):
        actor._forward_backward(obs, None, rows, act, None, old_logp, adv, None, factor, active)
    torch.cuda.synchronize()
    _lib.enable_kernel_timing(True)
    for _ in range(reps):
        actor._forward_backward(obs, None, rows, act, None, old_logp, adv, None, factor, active)
    t = _lib.collect_kernel_timing()
    _lib.enable_kernel_timing(False)
    return {k: round(v["avg_ms"], 4) for k, v in t.items() if v["avg_ms"] > 0.005}


if __name__ == "__main__":
    for rows in (65536, 131072, 262144, 524288, 819200):
        for mode in ("hybrid", "logp"):
            print(rows, mode, run(rows, mode), flush=True)
