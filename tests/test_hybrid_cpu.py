"""Routing of the optimiser steps (nets.fused_update_ok / fused_last_ok), host side without a GPU: the launch sequence of a whole
train() with the C-ABI calls recorded instead of executed."""
import numpy as np
import pytest
import torch

from harl_amd.synthetic import Shapes, make_buffers
from tests.gpu_checks import Box, default_args
from tests.test_multidiscrete_cpu import stub_kernels  # noqa: F401  (fixture)


def _runner(hidden, obs=19, sobs=11, act=3, T=8, N=8, A=2, **over):
    from harl_amd.runner import RUNNER_REGISTRY
    a = default_args(hidden, ppo_epoch=2, critic_epoch=2, **over)
    train = dict(n_rollout_threads=N, episode_length=T, use_valuenorm=True, use_linear_lr_decay=False,
                 use_proper_time_limits=True, model_dir=None, eval_interval=25, use_eval=False, log_interval=1,
                 num_env_steps=T * N * 2)
    r = RUNNER_REGISTRY["happo"](dict(algo="happo"), dict(train=train, model=dict(a), algo=dict(a)), dict(state_type="EP"),
                                 obs_spaces=[Box((obs,))] * A, share_obs_space=Box((sobs,)), act_spaces=[Box((act,))] * A,
                                 device=torch.device("cpu"))
    sh = Shapes(T=T, N=N, A=A, obs_dim=obs, share_obs_dim=sobs, act_dim=act, hidden_sizes=list(hidden))
    d = make_buffers(sh, 4)
    for ag in range(A):
        b = r.actor_buffer[ag]
        b.obs.copy_(torch.from_numpy(d.obs[ag]))
        b.actions.copy_(torch.from_numpy(d.actions[ag]))
        b.action_log_probs.copy_(torch.from_numpy(d.action_log_probs[ag]))
    return r


def _train(r, calls):
    calls.clear()
    r.prep_training()
    r.train()
    return {k: len(v) for k, v in calls.items()}


def test_two_layer_networks_take_the_hybrid_step(stub_kernels, monkeypatch):  # noqa: F811
    """Default: fused forward + loss launch (harl_update_fwd_*) with layer 1's activation record requested, then the layer
    kernels' backward; HARL_FUSED_UPDATE=logp: the layer kernels for the whole step; =1: harl_update_bwd."""
    A, n_upd = 2, 2
    r = _runner([128, 128])
    n = _train(r, stub_kernels)
    assert n["harl_update_fwd_actor"] == A * n_upd and n["harl_update_fwd_critic"] == n_upd
    assert n["harl_mlp_bwd_dx"] == (A + 1) * n_upd and n["harl_mlp_dw_partials"] == (A + 1) * n_upd and "harl_mlp_bwd_dx_dw" not in n
    assert "harl_actor_head_loss" not in n and "harl_update_bwd" not in n and "harl_mlp_fwd_fused2x" not in n
    for c in stub_kernels["harl_update_fwd_actor"]:
        assert c[31] is not None and c[32] is not None and c[33] is not None  # xh1, rmask1, rstd1: the hybrid outputs
    # HARL_BWD_FUSED=1: the whole backward of the 128 x 128 layer in ONE launch (round 5: dx + dW_2' + the fused dW_1', inputs <= 32 wide)
    monkeypatch.setenv("HARL_BWD_FUSED", "1")
    n = _train(_runner([128, 128]), stub_kernels)
    assert n["harl_mlp_bwd_dx_dw"] == (A + 1) * n_upd and "harl_mlp_bwd_dx" not in n and "harl_mlp_dw_partials" not in n
    for c in stub_kernels["harl_mlp_bwd_dx_dw"]:
        assert c[8] is None and c[9] is not None and c[10] == 32 and c[11] is not None and c[12] is not None and c[14] == 1
    monkeypatch.delenv("HARL_BWD_FUSED")
    monkeypatch.setenv("HARL_FUSED_UPDATE", "logp")
    n = _train(_runner([128, 128]), stub_kernels)
    assert "harl_update_fwd_actor" not in n and n["harl_actor_head_loss"] == A * n_upd and n["harl_critic_head_loss"] == n_upd
    assert n["harl_update_logp"] == A  # the post-update log-prob passes stay on the fused forward-only launch
    monkeypatch.setenv("HARL_FUSED_UPDATE", "1")
    n = _train(_runner([128, 128]), stub_kernels)
    assert n["harl_update_bwd"] == (A + 1) * n_upd and "harl_mlp_bwd_dx" not in n
    for c in stub_kernels["harl_update_fwd_actor"]:
        assert c[31] is None


def test_deeper_networks_run_their_last_layer_inside_the_loss_launch(stub_kernels, monkeypatch):  # noqa: F811
    A, n_upd = 2, 2
    r = _runner([128, 128, 128])
    n = _train(r, stub_kernels)
    assert n["harl_update_last_actor"] == A * n_upd and n["harl_update_last_critic"] == n_upd
    assert "harl_actor_head_loss" not in n and "harl_critic_head_loss" not in n
    # hidden forward launches: only the log-prob passes (one third layer each: A post-update passes) -- the optimiser steps'
    # third layer is inside harl_update_last_*
    assert n.get("harl_mlp_fwd_hidden", 0) == A
    # the backward is the layer kernels': two bwd_dx and two hidden weight gradients per step
    assert n["harl_mlp_bwd_dx"] == 2 * (A + 1) * n_upd and n["harl_mlp_dw_partials"] == 2 * (A + 1) * n_upd
    monkeypatch.setenv("HARL_BWD_FUSED", "1")  # one launch per hidden Linear (layer 3 -> 2 writes dz_2, layer 2 -> 1 carries dW_1')
    n = _train(_runner([128, 128, 128]), stub_kernels)
    assert n["harl_mlp_bwd_dx_dw"] == 2 * (A + 1) * n_upd and "harl_mlp_bwd_dx" not in n and "harl_mlp_dw_partials" not in n
    assert sum(1 for c in stub_kernels["harl_mlp_bwd_dx_dw"] if c[11] is None and c[8] is not None) == (A + 1) * n_upd
    monkeypatch.delenv("HARL_BWD_FUSED")
    for c in stub_kernels["harl_update_last_actor"]:
        assert c[2] == 128 and c[12] is None  # width; identity row order (one minibatch)
    monkeypatch.setenv("HARL_FUSED_UPDATE", "logp")
    n = _train(_runner([128, 128, 128]), stub_kernels)
    assert "harl_update_last_actor" not in n and n["harl_actor_head_loss"] == A * n_upd


def test_shapes_outside_the_fused_launches_fall_back(stub_kernels):  # noqa: F811
    """33..64 inputs into 128-wide layers: the ACTOR step does not fit the LDS of one workgroup (harl_update_supported) and
    runs on the layer kernels; its forward-only passes and the critic's step keep the fused launch.  Unequal widths: layer
    kernels throughout."""
    A, n_upd = 2, 2
    n = _train(_runner([128, 128], obs=40, sobs=60), stub_kernels)
    assert "harl_update_fwd_actor" not in n and n["harl_actor_head_loss"] == A * n_upd
    assert n["harl_update_fwd_critic"] == n_upd and n["harl_update_logp"] == A
    n = _train(_runner([128, 64]), stub_kernels)
    assert not any(k.startswith("harl_update_") for k in n) and n["harl_actor_head_loss"] == A * n_upd
