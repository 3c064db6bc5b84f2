"""Deterministic synthetic rollout data and parameters (SURVEY.md §8d recipe).

Pure NumPy; shared by ``bench.py``, the tests and ``oracle/gen_golden.py`` so the
reference, the oracle and the HIP path all see byte-identical inputs.  Nothing in
here is part of the compute path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class Shapes:
    T: int                 # episode_length
    N: int                 # n_rollout_threads
    A: int                 # num_agents
    obs_dim: int
    share_obs_dim: int
    act_dim: int           # Box: dimension; Discrete: number of actions
    discrete: bool = False
    hidden_sizes: Sequence[int] = (128, 128)
    nvec: Optional[Sequence[int]] = None   # MultiDiscrete: actions per head (act_dim = their sum, discrete = True)
    recurrent_n: int = 1   # stacked GRU layers of recurrent policies (models/base/rnn.py:14)

    @property
    def act_shape(self) -> int:  # width of the stored `actions` / `action_log_probs`
        if self.nvec is not None:
            return len(self.nvec)
        return 1 if self.discrete else self.act_dim


def _rnn_shapes(h: int, recurrent_n: int = 1) -> List[Tuple[str, Tuple[int, ...]]]:
    """RNNLayer (models/base/rnn.py:8-21): nn.GRU(h, h, recurrent_n) parameters in registration order (per layer: weight_ih,
    weight_hh, bias_ih, bias_hh), then the LayerNorm."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    for l in range(recurrent_n):
        out += [(f"rnn.rnn.weight_ih_l{l}", (3 * h, h)), (f"rnn.rnn.weight_hh_l{l}", (3 * h, h)), (f"rnn.rnn.bias_ih_l{l}", (3 * h,)),
                (f"rnn.rnn.bias_hh_l{l}", (3 * h,))]
    return out + [("rnn.norm.weight", (h,)), ("rnn.norm.bias", (h,))]


def actor_param_shapes(sh: Shapes, use_feature_normalization: bool = True,
                       recurrent: bool = False) -> List[Tuple[str, Tuple[int, ...]]]:
    """Parameter names/shapes in the reference's ``StochasticPolicy.parameters()`` order
    (harl/models/policy_models/stochastic_policy.py:11-54; probe in SURVEY.md §8a M1)."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    if use_feature_normalization:
        out += [("base.feature_norm.weight", (sh.obs_dim,)), ("base.feature_norm.bias", (sh.obs_dim,))]
    d = sh.obs_dim
    for i, h in enumerate(sh.hidden_sizes):
        out += [(f"base.mlp.fc.{3*i}.weight", (h, d)), (f"base.mlp.fc.{3*i}.bias", (h,)),
                (f"base.mlp.fc.{3*i+2}.weight", (h,)), (f"base.mlp.fc.{3*i+2}.bias", (h,))]
        d = h
    if recurrent:
        out += _rnn_shapes(d, sh.recurrent_n)
    if sh.nvec is not None:  # act.py:35-43: nn.ModuleList of Categoricals
        for k, n in enumerate(sh.nvec):
            out += [(f"act.action_outs.{k}.linear.weight", (int(n), d)), (f"act.action_outs.{k}.linear.bias", (int(n),))]
    elif sh.discrete:
        out += [("act.action_out.linear.weight", (sh.act_dim, d)), ("act.action_out.linear.bias", (sh.act_dim,))]
    else:
        out += [("act.action_out.log_std", (sh.act_dim,)),
                ("act.action_out.fc_mean.weight", (sh.act_dim, d)), ("act.action_out.fc_mean.bias", (sh.act_dim,))]
    return out


def critic_param_shapes(sh: Shapes, use_feature_normalization: bool = True,
                        recurrent: bool = False) -> List[Tuple[str, Tuple[int, ...]]]:
    """``VNet.parameters()`` order (harl/models/value_function_models/v_net.py:10-46)."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    if use_feature_normalization:
        out += [("base.feature_norm.weight", (sh.share_obs_dim,)), ("base.feature_norm.bias", (sh.share_obs_dim,))]
    d = sh.share_obs_dim
    for i, h in enumerate(sh.hidden_sizes):
        out += [(f"base.mlp.fc.{3*i}.weight", (h, d)), (f"base.mlp.fc.{3*i}.bias", (h,)),
                (f"base.mlp.fc.{3*i+2}.weight", (h,)), (f"base.mlp.fc.{3*i+2}.bias", (h,))]
        d = h
    if recurrent:
        out += _rnn_shapes(d, sh.recurrent_n)
    out += [("v_out.weight", (1, d)), ("v_out.bias", (1,))]
    return out


def synthetic_state_dict(shapes: List[Tuple[str, Tuple[int, ...]]], seed: int, std_x_coef: float = 1.0) -> Dict[str, np.ndarray]:
    """Seed-reproducible, deliberately *non-trivial* parameters: LayerNorm affine terms and
    biases are perturbed so that every gradient path (incl. LN weight/bias, log_std) is exercised."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for name, shp in shapes:
        if name.endswith("log_std"):
            v = std_x_coef + 0.1 * rng.standard_normal(shp)
        elif len(shp) == 2:  # Linear weight [out, in]
            scale = (0.3 if ("action_out" in name or name.startswith("v_out")) else 1.4) / np.sqrt(shp[1])  # (incl. action_outs.k)
            v = scale * rng.standard_normal(shp)
        elif name == "rnn.norm.weight":
            v = 1.0 + 0.1 * rng.standard_normal(shp)
        elif "feature_norm.weight" in name or (name.startswith("base.mlp.fc.") and int(name.split(".")[3]) % 3 == 2 and name.endswith("weight")):
            v = 1.0 + 0.1 * rng.standard_normal(shp)
        else:  # biases (Linear and LayerNorm)
            v = 0.1 * rng.standard_normal(shp)
        sd[name] = v.astype(np.float32)
    return sd


@dataclass
class SyntheticBuffers:
    """Host arrays in the reference buffer shapes (actor_buffer.py:31-76, critic_buffer_ep.py:27-71)."""
    obs: List[np.ndarray]               # A x [T+1, N, D_o]
    actions: List[np.ndarray]           # A x [T, N, act_shape]
    action_log_probs: List[np.ndarray]  # A x [T, N, act_shape]
    masks: List[np.ndarray]             # A x [T+1, N, 1]
    active_masks: List[np.ndarray]      # A x [T+1, N, 1]
    available_actions: List[Optional[np.ndarray]]  # A x ([T+1, N, n_act] | None)
    share_obs: np.ndarray               # [T+1, N, D_s]
    rewards: np.ndarray                 # [T, N, 1]
    value_preds: np.ndarray             # [T+1, N, 1]
    critic_masks: np.ndarray            # [T+1, N, 1]
    bad_masks: np.ndarray               # [T+1, N, 1]
    fp: Optional[dict] = None           # FP state type: share_obs/value_preds/masks/bad_masks [T+1,N,A,.], rewards [T,N,A,1]
    rnn: Optional[dict] = None          # recurrent policies: actor = A x [T+1,N,1,H], critic = [T+1,N,1,H], critic_fp = [T+1,N,A,1,H]


def make_buffers(sh: Shapes, seed: int, inactive_p: float = 0.0, unavailable_p: float = 0.0,
                 fp: bool = False, rnn: bool = False) -> SyntheticBuffers:
    """obs/share_obs/rewards/value_preds ~ N(0,1); Box actions ~ N(0,1) with stored
    log-probs -1+0.1 N(0,1); Discrete actions ~ U{0..n-1} stored as fp32 with log-probs
    log(1/n)+0.05 N(0,1); masks 0 w.p. 0.04 with bad_masks 0 at the same places; critic masks =
    agent-0 masks.  ``inactive_p`` / ``unavailable_p`` > 0 add dead agents / masked actions for the
    edge-case tests (the taken action always stays available)."""
    rng = np.random.default_rng(seed)
    T, N, A = sh.T, sh.N, sh.A
    f32 = np.float32
    obs, actions, logp, masks, active, avail = [], [], [], [], [], []
    base_mask = (rng.random((T + 1, N, 1)) >= 0.04).astype(f32)
    for _ in range(A):
        obs.append(rng.standard_normal((T + 1, N, sh.obs_dim)).astype(f32))
        if sh.nvec is not None:
            # one index per head; the stored log-probs are the SUMMED log-prob broadcast over the heads' columns
            # (actor_buffer.py:98 assigns the policy's [N, 1] output into the [N, n_heads] slot) -- plus a little per-column
            # noise so that the kernels' per-column arithmetic is exercised
            nh = len(sh.nvec)
            a = np.stack([rng.integers(0, int(n), size=(T, N)) for n in sh.nvec], axis=-1)
            actions.append(a.astype(f32))
            base = sum(np.log(1.0 / int(n)) for n in sh.nvec) + 0.05 * rng.standard_normal((T, N, 1))
            logp.append((base + 0.01 * rng.standard_normal((T, N, nh))).astype(f32))
            avail.append(None)
        elif sh.discrete:
            a = rng.integers(0, sh.act_dim, size=(T, N, 1))
            actions.append(a.astype(f32))
            logp.append((np.log(1.0 / sh.act_dim) + 0.05 * rng.standard_normal((T, N, 1))).astype(f32))
            av = np.ones((T + 1, N, sh.act_dim), dtype=f32)
            if unavailable_p > 0:
                av = (rng.random((T + 1, N, sh.act_dim)) >= unavailable_p).astype(f32)
                np.put_along_axis(av[:-1], a, 1.0, axis=-1)
            avail.append(av)
        else:
            actions.append(rng.standard_normal((T, N, sh.act_dim)).astype(f32))
            logp.append((-1.0 + 0.1 * rng.standard_normal((T, N, sh.act_dim))).astype(f32))
            avail.append(None)
        masks.append(base_mask.copy())
        am = np.ones((T + 1, N, 1), dtype=f32)
        if inactive_p > 0:
            am = (rng.random((T + 1, N, 1)) >= inactive_p).astype(f32)
        active.append(am)
    share_obs = rng.standard_normal((T + 1, N, sh.share_obs_dim)).astype(f32)
    rewards = rng.standard_normal((T, N, 1)).astype(f32)
    value_preds = rng.standard_normal((T + 1, N, 1)).astype(f32)
    bad = np.where(base_mask == 0.0, 0.0, 1.0).astype(f32)
    out = SyntheticBuffers(obs, actions, logp, masks, active, avail, share_obs, rewards, value_preds,
                           base_mask.copy(), bad)
    if fp:  # feature-pruned (per-agent) critic inputs (on_policy_critic_buffer_fp.py:13-83); drawn AFTER the EP arrays
        m_fp = np.repeat(base_mask[:, :, None, :], A, axis=2)
        out.fp = dict(share_obs=rng.standard_normal((T + 1, N, A, sh.share_obs_dim)).astype(f32),
                      rewards=rng.standard_normal((T, N, A, 1)).astype(f32),
                      value_preds=rng.standard_normal((T + 1, N, A, 1)).astype(f32),
                      masks=m_fp.astype(f32), bad_masks=np.where(m_fp == 0.0, 0.0, 1.0).astype(f32))
    if rnn:  # stored GRU hidden states (drawn last so the other arrays do not depend on the flag)
        hh, rn = sh.hidden_sizes[-1], sh.recurrent_n
        out.rnn = dict(actor=[(0.3 * rng.standard_normal((T + 1, N, rn, hh))).astype(f32) for _ in range(A)],
                       critic=(0.3 * rng.standard_normal((T + 1, N, rn, hh))).astype(f32))
        if fp:  # per-agent critic hidden states (on_policy_critic_buffer_fp.py:48-56)
            out.rnn["critic_fp"] = (0.3 * rng.standard_normal((T + 1, N, A, rn, hh))).astype(f32)
    return out
