"""HATRPO's Fisher-vector product on the configurations added in round 4, host side without a GPU: which C-ABI entry points the
tangent pass launches (kernels stubbed, tests/test_multidiscrete_cpu.py::stub_kernels).  The arithmetic itself is the GPU
tests' business (test_hatrpo_activation_* / _composed_gru_* / _width256_* against the oracle and the reference's goldens)."""
import numpy as np
import pytest
import torch

from tests.gpu_checks import Box, Discrete, default_args
from tests.test_multidiscrete_cpu import stub_kernels  # noqa: F401  (fixture)

TRPO = dict(kl_threshold=0.01, ls_step=10, accept_ratio=0.5, backtrack_coeff=0.8)


@pytest.fixture(autouse=True)
def _gru_wide_follows_the_recorder(stub_kernels, monkeypatch):  # noqa: F811
    """gru_wide binds ``call`` / ``stream`` by name at its first import (inside whichever test got there first): point it at
    THIS test's recorder."""
    from harl_amd import _lib, gru_wide
    monkeypatch.setattr(gru_wide, "call", _lib.call)
    monkeypatch.setattr(gru_wide, "stream", _lib.stream)


def _fvp_calls(calls, hidden, obs_dim, space, m=64, **over):
    from harl_amd.hatrpo import HATRPO
    from harl_amd.nets import build_seq
    dev = torch.device("cpu")
    t = HATRPO(default_args(hidden, **TRPO, **over), Box((obs_dim,)), space, device=dev)
    net = t.actor
    f = lambda *s: torch.zeros(*s, dtype=torch.float32)  # noqa: E731
    obs, act = f(m, obs_dim), f(m, net.act_w)
    seq = None
    if net.recurrent:  # two sequences of L = m / 2 steps (padded to a 32-sequence slab)
        seq = build_seq(dev, m // 2, 2, hidden[-1] * net.recurrent_n, h0=f(2, hidden[-1] * net.recurrent_n), masks_src=torch.ones(m))
    rows = m if seq is None else seq["L"] * seq["m_pad"]
    net.fold()
    t._surrogate(obs, rows, act, None, f(m, net.act_w), f(m), None, f(m), torch.ones(m), want_grad=True, seq=seq)
    calls.clear()
    out = t._fvp(obs, rows, m, None, torch.zeros(net.n_params), seq=seq)
    assert tuple(out.shape) == (net.n_params,)
    return t, {k: len(v) for k, v in calls.items()}


@pytest.mark.parametrize("act", ["tanh", "selu"])
def test_activation_tangent_launches(stub_kernels, act):  # noqa: F811
    t, n = _fvp_calls(stub_kernels, [128, 128, 128], 40, Box((3,)), activation_func=act)
    assert n["harl_act_ln_tangent"] == 3                      # one per layer
    assert n["harl_mlp_linear_wide"] == 1                     # layer 0: W'_dot x0n + b'_dot (the inputs carry no tangent)
    assert n["harl_mlp_linear"] == 2 * 2                      # hidden layers: W' x_dot and W'_dot x_hat + b'_dot
    assert "harl_mlp_tangent_hidden2" not in n and "harl_mlp_tangent_wide" not in n and "harl_mlp_tangent_input" not in n
    assert n["harl_act_bwd"] == 3 and n["harl_actor_head_fvp"] == 1   # J^T: the generic backward of these networks


def test_width256_tangent_launches(stub_kernels):  # noqa: F811
    t, n = _fvp_calls(stub_kernels, [256, 256], 44, Discrete(6))
    assert n["harl_mlp_panel_tangent"] == 2 and n["harl_head_dw_rows256"] == 1 and n["harl_mlp_panel_bwd"] == 1
    args = stub_kernels["harl_mlp_panel_tangent"]
    assert args[0][0] is None and args[0][3] == t.actor.kp0 and args[0][4] is None    # first layer: no x_in_dot, no W'
    assert args[1][0] is not None and args[1][3] == 256 and args[1][4] is not None    # hidden: both halves


@pytest.mark.parametrize("hidden,rn", [([128, 128], 1), ([64], 2)])
def test_composed_gru_tangent_launches(stub_kernels, hidden, rn):  # noqa: F811
    m = 20  # two sequences of 10 steps
    t, n = _fvp_calls(stub_kernels, hidden, 18, Box((4,)), m=m, use_recurrent_policy=True, recurrent_n=rn)
    assert t.actor.gru_wide and t.actor.recurrent_n == rn
    L = m // 2
    assert n["harl_gru_cell_tangent"] == L * rn
    # gate products travel three to a launch (harl_mlp_linear3, round 6).  Per layer: 2 input-side launches over all steps + per
    # step 1 (W_h_dot h~) and, except at the first step, 1 more (W_h h~_dot); the backward (gru_wide.backward) adds 1 per step but
    # the first; what a layer sends to the one below (stacked GRUs) stays three single harl_mlp_linear launches
    tangent_launches = rn * (2 + L + (L - 1))
    backward_launches = rn * (L - 1)
    assert n["harl_mlp_linear3"] == tangent_launches + backward_launches
    assert n.get("harl_mlp_linear", 0) == 3 * (rn - 1)
    assert n["harl_act_ln_tangent"] == 1                       # rnn.norm's tangent (no activation)
    ln = stub_kernels["harl_act_ln_tangent"][0]
    assert ln[1] is None and ln[3] is None and ln[7] == 0
    first = stub_kernels["harl_gru_cell_tangent"][0]
    assert first[6] is None and first[7] is None and first[8] is None and first[17] is None   # h0 carries no tangent
    last = stub_kernels["harl_gru_cell_tangent"][L - 1]
    assert last[18] is None and last[22] is None               # no next step: no masks, no h~_dot out
    assert "harl_gru_tangent" not in n and "harl_gru_gates" not in n   # the fused 64-wide kernels are not on this path
