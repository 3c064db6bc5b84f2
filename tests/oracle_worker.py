"""Worker process of tests/gpu_checks._oracle_bench_runs: ONE oracle compute() + train() on the host copies of a bench
configuration's buffers and weights (CPU only -- the GPU is hidden from this process), result written with torch.save.

    python -m tests.oracle_worker <payload.pt> <out.pt> <tag> <f32|f64> <one-ulp seed | none> <keep_grad 0|1>

Test infrastructure, like everything under oracle/: never imported by harl_amd/."""
import sys

import torch


def main() -> int:
    import os
    pin, pout, tag, dtn, seed, keep = sys.argv[1:7]
    cores = os.environ.get("HARL_ORACLE_CORES")
    if cores:  # this worker's slice of the host (tests/gpu_checks.prefetch_full_size)
        try:
            os.sched_setaffinity(0, {int(c) for c in cores.split(",")})
        except (AttributeError, OSError, ValueError):
            pass
    from tests import gpu_checks as G
    payload = torch.load(pin, weights_only=False)
    res = G._oracle_bench_run(payload, tag, dtn, None if seed == "none" else int(seed), bool(int(keep)))
    torch.save(res, pout)
    return 0


if __name__ == "__main__":
    sys.exit(main())
