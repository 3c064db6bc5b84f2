"""The drop-in seam under the UNMODIFIED reference launcher, on the CPU (BASELINE.json configs[0]).

Needs the reference checkout (``HARL_REFERENCE`` or /root/reference); skipped where it is absent (the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HARL_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "examples", "train.py")),
                                reason="reference checkout not available")

COMMON = ["--algo", "happo", "--env", "pettingzoo_mpe", "--exp_name", "dropin_test", "--n_rollout_threads", "4",
          "--episode_length", "50", "--num_env_steps", "800", "--eval_interval", "2", "--n_eval_rollout_threads", "2",
          "--eval_episodes", "2", "--log_interval", "1", "--cuda", "False"]


def _drive(tmp_path, mode, extra=(), stub=False):
    cmd = [sys.executable, os.path.join(ROOT, "tests", "dropin_driver.py"), "--reference", REF, "--mode", mode,
           "--log-dir", str(tmp_path)] + (["--stub-kernels"] if stub else []) + COMMON + list(extra)
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("DROPIN_RESULT ")][-1]
    return json.loads(line[len("DROPIN_RESULT "):]), p.stdout


def test_baseline_config0_reference_cpu_path(tmp_path):
    """BASELINE.json configs[0]: MPE simple_spread shapes, 3 agents, HAPPO, n_rollout_threads = 4, MLP policy on the
    reference's own CPU path through the unmodified examples/train.py (4 episodes, with evaluation and checkpoints)."""
    res, out = _drive(tmp_path, "reference")
    assert res["actor_class"] == "harl.algorithms.actors.happo.HAPPO"
    assert res["obs_type"] == "ndarray"
    assert {"actor_agent0.pt", "actor_agent1.pt", "actor_agent2.pt", "critic_agent.pt", "value_normalizer.pt"} <= set(res["saved"])
    assert "Evaluation average episode reward" in out


def test_dropin_under_unmodified_train_py_plumbing(tmp_path):
    """harl_amd.dropin.install() + the unmodified examples/train.py: the runner IS a subclass of the reference's
    OnPolicyBaseRunner (its constructor builds the environments, the logger and the run directory; its run(), eval(),
    save() and close() execute), actors / critic / buffers / ValueNorm are the harl_amd classes, buffers hold torch tensors.
    No GPU here: the C-ABI calls are recorded instead of executed, so this is plumbing only -- every kernel family of
    the rollout and of the update must have been reached."""
    res, out = _drive(tmp_path, "dropin", stub=True)
    assert res["base_classes"][0] == "harl_amd.dropin.OnPolicyHARunner"
    assert res["base_classes"][1] == "harl.runners.on_policy_base_runner.OnPolicyBaseRunner"
    assert res["actor_class"] == "harl_amd.happo.HAPPO" and res["critic_class"] == "harl_amd.v_critic.VCritic"
    assert res["buffer_class"] == "harl_amd.buffers.OnPolicyActorBuffer" and res["obs_type"] == "Tensor"
    assert {"actor_agent0.pt", "actor_agent1.pt", "actor_agent2.pt", "critic_agent.pt", "value_normalizer.pt"} <= set(res["saved"])
    assert "config.json" in res["run_dir_files"] and "progress.txt" in res["run_dir_files"]
    k = res["kernel_calls"]
    for name in ("harl_gae_returns", "harl_masked_moments", "harl_adam_fold", "harl_reduce_partials_multi",
                 "harl_update_fwd_actor", "harl_update_fwd_critic", "harl_mlp_bwd_dx", "harl_mlp_dw_partials",
                 "harl_mlp_x0n_wide"):  # hybrid optimiser step (nets.fused_update_ok): fused forward, layer-by-layer backward
        assert k.get(name, 0) > 0, (name, k)
    assert k.get("harl_update_logp", 0) + k.get("harl_actor_head_logp", 0) > 0, k   # rollout sampling + factor passes
    assert k["harl_adam_fold"] == 4 * (3 * 5 + 5), k                              # 4 episodes x (3 agents x 5 + 5 critic epochs)
    assert "Evaluation average episode reward" in out
