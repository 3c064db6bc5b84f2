"""`python bench.py --gpus N` must work as typed (missing item 1): with no launcher around it, bench.py re-executes
itself under torch.distributed.run, one rank per GPU.  Checked here without a GPU through --dry-run (rendezvous over gloo, one
all-reduce, ONE JSON line on stdout from rank 0), both self-spawned and under an explicit torchrun as the driver does it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 4])
def test_bench_self_spawns_ranks(n):
    out = _run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0", "--dry-run"])
    assert out["n_gpus"] == n and out["config"]["parallelism"] == f"dp{n}" and out["config"]["collective"] == "gloo"
    # the multi-GPU line answers the BASELINE's question: headline = strong scaling at the BASELINE's
    # global n_rollout_threads, the weak figure on the same line, and the rank count as the communicator itself reports it
    assert out["scaling"] == "strong" and out["config"]["n_rollout_threads_global"] == 4096
    assert out["config"]["n_rollout_threads_per_gpu"] == 4096 // n and out["config"]["ranks_seen"] == n
    assert out["weak"]["n_rollout_threads_global"] == 4096 * n


def test_bench_strong_size_of_cheetah6_is_configs2():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--config", "cheetah6", "--dry-run"])
    assert out["scaling"] == "strong" and out["config"]["n_rollout_threads_global"] == 8192
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--scaling", "weak", "--dry-run"])
    assert out["scaling"] == "weak" and out["config"]["n_rollout_threads_global"] == 8192 and out["weak"] is None


def test_bench_under_explicit_torchrun():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29611", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run"])
    assert out["n_gpus"] == 2


def test_bench_second_line_flag():
    """--with-strong-cheetah6: the invocation's own line, then BASELINE configs[2] (strong scaling, 8192 global threads)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--with-strong-cheetah6"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 2 and "MPE" in lines[0]["metric"] and "HalfCheetah" in lines[1]["metric"], p.stdout
    assert all(ln["n_gpus"] == 2 for ln in lines)


def test_bench_single_rank_dry():
    out = _run([sys.executable, "bench.py", "--dry-run"])
    assert out["n_gpus"] == 1 and out["config"]["collective"] == "none"
    assert out["scaling"] == "weak" and out["config"]["n_rollout_threads_global"] == 4096 and out["config"]["ranks_seen"] == 1
