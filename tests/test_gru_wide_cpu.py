"""128-wide GRU policies (harl_amd/gru_wide.py), host side without a GPU: the composed launch sequence of a whole train() with
the C-ABI calls recorded instead of executed, and the opt-in gate (experimental until the GPU parity tests have run)."""
import numpy as np
import pytest
import torch

from harl_amd.synthetic import Shapes, make_buffers
from tests.gpu_checks import Box, default_args
from tests.test_multidiscrete_cpu import stub_kernels  # noqa: F401  (fixture)


def _runner(hidden, T=10, N=6, A=2):
    from harl_amd.runner import RUNNER_REGISTRY
    a = default_args(hidden, ppo_epoch=2, critic_epoch=2, use_recurrent_policy=True, data_chunk_length=5)
    train = dict(n_rollout_threads=N, episode_length=T, use_valuenorm=True, use_linear_lr_decay=False,
                 use_proper_time_limits=True, model_dir=None, eval_interval=25, use_eval=False, log_interval=1,
                 num_env_steps=T * N * 2)
    return RUNNER_REGISTRY["happo"](dict(algo="happo"), dict(train=train, model=dict(a), algo=dict(a)), dict(state_type="EP"),
                                    obs_spaces=[Box((19,))] * A, share_obs_space=Box((11,)), act_spaces=[Box((3,))] * A,
                                    device=torch.device("cpu"))


def test_gru128_is_opt_in(stub_kernels, monkeypatch):  # noqa: F811
    monkeypatch.delenv("HARL_GRU128", raising=False)
    with pytest.raises(NotImplementedError):
        _runner([128, 128])
    monkeypatch.setenv("HARL_GRU128", "1")
    r = _runner([128, 128])
    assert r.actor[0].actor.gru_wide and r.critic.critic.gru_wide
    from harl_amd.hatrpo import HATRPO
    with pytest.raises(NotImplementedError):
        HATRPO(default_args([128, 128], use_recurrent_policy=True, kl_threshold=0.01, ls_step=10, accept_ratio=0.5,
                            backtrack_coeff=0.8), Box((19,)), Box((3,)), device=torch.device("cpu"))


def test_gru128_launch_sequence(stub_kernels, monkeypatch):  # noqa: F811
    monkeypatch.setenv("HARL_GRU128", "1")
    calls = stub_kernels
    T, N, A, H = 10, 6, 2, 128
    r = _runner([128, 128], T, N, A)
    sh = Shapes(T=T, N=N, A=A, obs_dim=19, share_obs_dim=11, act_dim=3, hidden_sizes=[128, 128])
    d = make_buffers(sh, 4, inactive_p=0.1, rnn=True)
    for ag in range(A):
        b = r.actor_buffer[ag]
        b.obs.copy_(torch.from_numpy(d.obs[ag]))
        b.actions.copy_(torch.from_numpy(d.actions[ag]))
        b.action_log_probs.copy_(torch.from_numpy(d.action_log_probs[ag]))
        b.rnn_states.copy_(torch.from_numpy(np.zeros((T + 1, N, 1, H), np.float32)))
    calls.clear()
    r.prep_training()
    infos, cinfo = r.train()
    assert len(infos) == A
    assert "harl_gru_fwd" not in calls and "harl_gru_bwd" not in calls      # the 64-wide fused kernels are not on this path
    L_full, L_chunk, n_upd = T, 5, 2
    # per agent: two full-length log-prob passes + n_upd chunked forward/backward passes; critic: n_upd chunked passes
    n_fwd_steps = A * (2 * L_full + n_upd * L_chunk) + n_upd * L_chunk
    n_bwd_steps = (A + 1) * n_upd * L_chunk
    assert len(calls["harl_gru_cell_fwd"]) == n_fwd_steps
    assert len(calls["harl_gru_cell_bwd"]) == n_bwd_steps
    n_passes_fwd = A * (2 + n_upd) + n_upd
    assert len(calls["harl_gru_cell_init"]) == n_passes_fwd and len(calls["harl_rownorm"]) == n_passes_fwd
    # GEMMs: 3 input-gate launches per pass + 3 per forward step + 3 per backward step except the first time step of a chunk
    n_bwd_passes = (A + 1) * n_upd
    assert len(calls["harl_mlp_linear"]) == 3 * n_passes_fwd + 3 * n_fwd_steps + 3 * (n_bwd_steps - n_bwd_passes)
    # every cell launch works on one time step of m_pad = 32 sequences at width 128; saved gates only in training passes
    assert all(c[8] == H and c[9] == 32 for c in calls["harl_gru_cell_fwd"])
    n_saving = sum(1 for c in calls["harl_gru_cell_fwd"] if c[10] is not None)
    assert n_saving == (A + 1) * n_upd * L_chunk
    # the last step of a pass has no successor: no next-step h~ / masks
    assert sum(1 for c in calls["harl_gru_cell_fwd"] if c[15] is None) == n_passes_fwd
    assert sum(1 for c in calls["harl_gru_cell_bwd"] if c[1] is None) == n_bwd_passes
    # weight gradients of the six gate blocks run on the ordinary two-operand kernel at HO = K = 128
    gate_dw = [c for c in calls["harl_mlp_dw_partials"] if c[3] == H and c[10] == H]
    assert len(gate_dw) >= 6 * n_bwd_passes
