"""MultiDiscrete action spaces, host side (no GPU): parameter container (reference state_dict names / order over a
group-contiguous arena), layer table, and the launch sequence of a whole train() / rollout step with the C-ABI calls recorded
instead of executed (the arithmetic itself is covered by the `-m gpu` goldens in tests/test_gpu_parity.py)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from harl_amd.synthetic import Shapes, actor_param_shapes, make_buffers, synthetic_state_dict
from tests.gpu_checks import Box, MultiDiscrete, default_args


@pytest.fixture
def stub_kernels(monkeypatch):
    """Replace the C-ABI dispatcher by a recorder (as tests/dropin_driver.py --stub-kernels does)."""
    from harl_amd import _lib
    calls = {}
    real_call = _lib.call

    def recorder(name, *args, tag=None):
        calls.setdefault(name, []).append(args)
        if name in ("harl_randperm_replay", "harl_rng_advance"):
            return real_call(name, *args, tag=tag)
        if name == "harl_masked_moments":  # (x, active, n, out3, scratch, stream): every entry active, so that train() proceeds
            ctypes.c_double.from_address(args[3] + 16).value = float(args[2])
        return None

    monkeypatch.setenv("HARL_DEVICE", "cpu")
    monkeypatch.setattr(_lib, "call", recorder)
    monkeypatch.setattr(_lib, "require_gpu", lambda device: None)
    monkeypatch.setattr(_lib, "stream", lambda: 0)
    monkeypatch.setattr(_lib, "scratch", lambda kind: 0)
    for mod in ("nets", "buffers", "happo", "hatrpo", "mappo", "v_critic", "valuenorm", "runner"):
        m = __import__(f"harl_amd.{mod}", fromlist=["x"])
        for nm in ("call", "stream"):
            if hasattr(m, nm):
                monkeypatch.setattr(m, nm, getattr(_lib, nm))

    class _Ev:
        def __init__(self, *a, **k):
            pass

        def record(self, *a):
            pass

        def wait(self, *a):
            pass

        def synchronize(self):
            pass

    monkeypatch.setattr(torch.cuda, "Event", _Ev)
    monkeypatch.setenv("HARL_SIDE_STREAM", "0")
    return calls


def _policy(nvec, hidden, **over):
    from harl_amd.nets import StochasticPolicy
    args = default_args(hidden, **over)
    return StochasticPolicy(args, Box((19,)), MultiDiscrete(nvec), torch.device("cpu")), args


def test_parameter_container_matches_reference_layout(stub_kernels):
    nvec, hidden = [41, 41, 41, 30], [128, 128]
    net, args = _policy(nvec, hidden)
    sh = Shapes(T=4, N=2, A=1, obs_dim=19, share_obs_dim=5, act_dim=sum(nvec), hidden_sizes=hidden, nvec=nvec)
    want = actor_param_shapes(sh, True)
    got = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert got == want                                   # the reference's names, shapes and ORDER (act.py:35-43)
    assert net._md_groups == [[0, 1, 2], [3]] and net._md_sp == [128, 64]
    assert net.act_w == 1 and net.n_heads == 4 and net.act_dim == 153
    sd = synthetic_state_dict(want, 3)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    # group 0 is ONE contiguous [123, 128] matrix of the arena, rows in head order; biases likewise
    w0 = net.pview("__md0.weight").numpy()
    assert np.array_equal(w0, np.concatenate([sd[f"act.action_outs.{k}.linear.weight"] for k in (0, 1, 2)], 0))
    assert np.array_equal(net.pview("__md1.bias").numpy(), sd["act.action_outs.3.linear.bias"])
    # every arena element belongs to exactly one registered parameter
    assert sum(p.numel() for p in net.parameters()) == net.n_params
    # gradients of the aliases are views of the gradient arena at the same offsets
    net.flat_grad.copy_(torch.arange(net.n_params, dtype=torch.float32))
    off, _ = net.offsets["act.action_outs.1.linear.weight"]
    assert net.get_parameter("act.action_outs.1.linear.weight").grad.reshape(-1)[0].item() == float(off)
    # table: the two groups are the last entries, partial layout of a 128 / 64 row weight-gradient GEMM
    rows = net._table_rows
    assert [r[4] for r in rows[-2:]] == [123, 30] and [r[10] for r in rows[-2:]] == [128, 64]
    assert [r[5] for r in rows[-2:]] == [128, 128]
    # both groups fold with the last hidden LayerNorm
    assert rows[-1][2] == rows[-2][2] == net.offsets["base.mlp.fc.5.weight"][0]
    # the packed matrices are full [sp][H] blocks (zero rows past the last head)
    assert rows[-1][6] - rows[-2][6] == 128 * 128 + 128
    assert float(net.pack_arena.abs().sum()) == 0.0      # (nothing folded yet: the recorder swallowed harl_fold_linear)


def test_unsupported_combinations_raise(stub_kernels):
    from harl_amd.hatrpo import HATRPO
    with pytest.raises(AssertionError):                  # the reference's own assertion (hatrpo.py:27-29)
        HATRPO(default_args([64, 64], kl_threshold=0.01, ls_step=10, accept_ratio=0.5, backtrack_coeff=0.8),
               Box((19,)), MultiDiscrete([3, 4]), device=torch.device("cpu"))
    with pytest.raises(NotImplementedError):
        _policy([3] * 9, [64, 64])                       # more than 8 heads
    with pytest.raises(NotImplementedError):
        _policy([3, 4], [256, 256])                      # panel (256-wide) trunks


@pytest.mark.parametrize("recurrent", [False, True])
def test_train_and_rollout_launch_sequence(stub_kernels, recurrent):
    from harl_amd.runner import RUNNER_REGISTRY
    calls = stub_kernels
    nvec = [41, 41, 41, 30]
    hidden = [64] if recurrent else [128, 128]
    T, N, A = 10, 6, 2
    sh = Shapes(T=T, N=N, A=A, obs_dim=19, share_obs_dim=11, act_dim=sum(nvec), hidden_sizes=hidden, nvec=nvec)
    a = default_args(hidden, ppo_epoch=2, critic_epoch=2, use_recurrent_policy=recurrent, data_chunk_length=5)
    train = dict(n_rollout_threads=N, episode_length=T, use_valuenorm=True, use_linear_lr_decay=False,
                 use_proper_time_limits=True, model_dir=None, eval_interval=25, use_eval=False, log_interval=1,
                 num_env_steps=T * N * 2)
    algo_args = dict(train=train, model=dict(a), algo=dict(a))
    r = RUNNER_REGISTRY["happo"](dict(algo="happo"), algo_args, dict(state_type="EP"),
                                 obs_spaces=[Box((sh.obs_dim,))] * A, share_obs_space=Box((sh.share_obs_dim,)),
                                 act_spaces=[MultiDiscrete(nvec)] * A, device=torch.device("cpu"))
    d = make_buffers(sh, 4, inactive_p=0.1, rnn=recurrent)
    for ag in range(A):
        b = r.actor_buffer[ag]
        assert tuple(b.actions.shape) == (T, N, 4) and tuple(b.action_log_probs.shape) == (T, N, 4)
        assert b.available_actions is None
        b.obs.copy_(torch.from_numpy(d.obs[ag]))
        b.actions.copy_(torch.from_numpy(d.actions[ag]))
        b.action_log_probs.copy_(torch.from_numpy(d.action_log_probs[ag]))
        b.active_masks.copy_(torch.from_numpy(d.active_masks[ag]))
    calls.clear()
    r.prep_training()
    infos, cinfo = r.train()
    assert len(infos) == A and set(infos[0]) == {"policy_loss", "dist_entropy", "actor_grad_norm", "ratio"}
    n_upd = A * 2                                         # agents x ppo_epoch (one mini-batch)
    assert len(calls["harl_md_head_loss"]) == n_upd
    # logits GEMMs: two groups x (optimiser steps + the log-prob passes of every agent: post-update, and pre-update unless
    # the first epoch's forward already is that pass -- feed-forward policies with one mini-batch, HAPPO.fuses_old_logp)
    n_lp = (2 if recurrent else 1) * A
    assert len(calls["harl_mlp_linear"]) == 2 * (n_upd + n_lp)
    assert len(calls["harl_md_head_logp"]) == n_lp
    if not recurrent:
        assert all(c[23] is not None for c in calls["harl_md_head_loss"][::2])   # first epoch emits log pi_old by position
    assert "harl_actor_head_loss" not in calls and "harl_actor_head_logp" not in calls
    # backward of the heads: per update one weight-gradient GEMM and one bwd_dx per group (HO = 128 and 64)
    head_dw = [c for c in calls["harl_mlp_dw_partials"] if c[3] in (128, 64) and c[10] == hidden[-1] and c[1] == 0]
    assert len([c for c in head_dw if c[3] == 128]) >= n_upd and len([c for c in calls["harl_mlp_bwd_dx"] if c[5] == 64]) >= n_upd
    loss_args = calls["harl_md_head_loss"][0]
    assert loss_args[2] == 2 and loss_args[4] == 4 and loss_args[11] == 4      # n_groups, n_heads, old_w = n_heads
    assert loss_args[16] is not None                                            # ent_scale (use_policy_active_masks)
    lp_args = calls["harl_md_head_logp"][-1]                                    # a post-update pass (factor product)
    assert lp_args[10] == 1 and lp_args[11] is not None                                                     # log-prob passes compare [B, 1] columns
    # rollout step
    calls.clear()
    obs = d.obs[0][0]
    rnn = np.zeros((N, 1, hidden[-1]), np.float32)
    masks = np.ones((N, 1), np.float32)
    acts, logp, rnn_out = r.actor[0].get_actions(obs, rnn, masks, None, deterministic=True)
    assert tuple(acts.shape) == (N, 4) and tuple(logp.shape) == (N, 1)
    assert len(calls["harl_md_head_logp"]) == 1 and calls["harl_md_head_logp"][0][13] is not None   # head_out requested
    lp, ent, dist = r.actor[0].evaluate_actions(obs, rnn, acts, masks, None, np.ones((N, 1), np.float32))
    assert tuple(lp.shape) == (N, 1) and ent.dim() == 0 and dist is None
