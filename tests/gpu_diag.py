#!/usr/bin/env python3
"""Run every GPU parity check and print all error figures (no early exit).  Usage on the GPU box:
    python tests/gpu_diag.py [filter ...]  > gpurun_out/diag.txt
"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tests import gpu_checks as G  # noqa: E402
from tests.helpers import ALL_CASES  # noqa: E402


def main():
    filt = sys.argv[1:]
    jobs = [("gae", G.check_gae), ("elementwise", G.check_elementwise), ("adam", G.check_adam)]
    for spec in G.FWD_SHAPES:
        jobs.append((f"forward[{spec['name']}]", lambda s=spec: G.check_forward(s)))
    for spec in G.FWD_SHAPES:
        jobs.append((f"grad[{spec['name']}]", lambda s=spec: G.check_gradients(s)))
    for spec in G.FWD_SHAPES[:3]:
        jobs.append((f"noise[{spec['name']}]", lambda s=spec: G.check_gradient_noise(s)))
    for nm in G.BASELINE_SHAPES:
        jobs.append((f"baseline[{nm}]", lambda n=nm: G.check_baseline_shape(n)))
    jobs.append(("grad[mpe_box,mean,inactive]", lambda: G.check_gradients(G.FWD_SHAPES[0], agg="mean", inactive_p=0.3)))
    for name in ALL_CASES:
        jobs.append((f"train[{name}]", lambda n=name: G.check_train_golden(n)))
    for i in (0, 1, 2, 4):
        jobs.append((f"trpo[{G.FWD_SHAPES[i]['name']}]", lambda s=G.FWD_SHAPES[i]: G.check_trpo(s)))
    from tests.helpers import MAPPO_CASES, RNN_CASES, TRPO_CASES, TRPO_RNN_CASES
    for name in TRPO_CASES + TRPO_RNN_CASES + RNN_CASES + MAPPO_CASES:
        jobs.append((f"train[{name}]", lambda n=name: G.check_train_golden(n)))
    print("device:", torch.cuda.get_device_name(0), flush=True)
    results = {}
    for name, fn in jobs:
        if filt and not any(f in name for f in filt):
            continue
        t0 = time.time()
        try:
            res = fn()
            torch.cuda.synchronize()
            worst = max([v for k, v in res.items() if not k.startswith("_") and isinstance(v, float)] or [0.0])
            print(f"== {name}  ({time.time()-t0:.1f}s)  worst={worst:.3e}")
            for k, v in res.items():
                if isinstance(v, float) and (v > 1e-6 or len(res) < 14 or name.startswith("noise")):
                    print(f"     {k:48s} {v:.3e}")
                elif not isinstance(v, float):
                    print(f"     {k:48s} {v}")
            results[name] = {k: (v if isinstance(v, (int, float, str)) else str(v)) for k, v in res.items()}
        except Exception:
            print(f"== {name}  FAILED")
            traceback.print_exc(file=sys.stdout)
            results[name] = "EXCEPTION"
        sys.stdout.flush()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
