"""CPU oracle for HARL's on-policy sequential-update path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
``harl_amd`` (the product) never imports anything under ``oracle/``.

It is an independent restatement, in plain torch-CPU fp32 + NumPy, of the algorithm
the reference implements in these files (paths relative to the reference root):

* GAE / returns reverse scan   harl/common/buffers/on_policy_critic_buffer_ep.py:97-200
* ValueNorm (PopArt-style)     harl/common/valuenorm.py:38-92
* actor / critic MLP           harl/models/base/mlp.py:7-70,
                               harl/models/policy_models/stochastic_policy.py:88-127,
                               harl/models/value_function_models/v_net.py:48-67
* action heads                 harl/models/base/act.py:104-157 (MultiDiscrete: :117-141), harl/models/base/distributions.py:7-89
* GRU layer                    harl/models/base/rnn.py:8-81
* HATRPO / HAA2C / MAPPO       harl/algorithms/actors/hatrpo.py:37-194, harl/utils/trpo_util.py:5-158,
                               harl/algorithms/actors/haa2c.py:28-153, harl/algorithms/actors/mappo.py:36-234
* HAPPO update / train         harl/algorithms/actors/happo.py:28-158
* V-critic update / train      harl/algorithms/critics/v_critic.py:75-200
* sequential update + factor   harl/runners/on_policy_ha_runner.py:11-130
* minibatch sampling           harl/common/buffers/on_policy_actor_buffer.py:114-178,
                               harl/common/buffers/on_policy_critic_buffer_ep.py:202-250

Third-party arithmetic on the path is PyTorch's (Linear / LayerNorm / autograd /
Adam / clip_grad_norm_ / randperm) and NumPy's (GAE, nanmean/nanstd), exactly as in
the reference; torch 2.10.0 / NumPy 2.2.6 are the versions
the golden fixtures were generated with.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference from
/root/reference, runs ``compute_returns`` + ``OnPolicyHARunner.train()`` on seeded
synthetic buffers and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against those fixtures
(returns and minibatch indices bit-exact, losses / grad-norms / params <= 1e-6 rel).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LOG_SQRT_2PI = math.log(math.sqrt(2.0 * math.pi))


# --------------------------------------------------------------------------------------
# configuration (the subset of happo.yaml / hatrpo.yaml the path reads)
# --------------------------------------------------------------------------------------
@dataclass
class PathConfig:
    """Knobs read by the path; defaults = harl/configs/algos_cfgs/happo.yaml."""

    hidden_sizes: Sequence[int] = (128, 128)
    use_feature_normalization: bool = True
    std_x_coef: float = 1.0
    std_y_coef: float = 0.5
    lr: float = 5e-4
    critic_lr: float = 5e-4
    opti_eps: float = 1e-5
    weight_decay: float = 0.0
    ppo_epoch: int = 5
    critic_epoch: int = 5
    use_clipped_value_loss: bool = True
    clip_param: float = 0.2
    actor_num_mini_batch: int = 1
    critic_num_mini_batch: int = 1
    entropy_coef: float = 0.01
    value_loss_coef: float = 1.0
    use_max_grad_norm: bool = True
    max_grad_norm: float = 10.0
    use_gae: bool = True
    gamma: float = 0.99
    gae_lambda: float = 0.95
    use_huber_loss: bool = True
    use_policy_active_masks: bool = True
    huber_delta: float = 10.0
    action_aggregation: str = "prod"
    fixed_order: bool = False
    use_proper_time_limits: bool = True
    use_valuenorm: bool = True
    use_recurrent_policy: bool = False
    use_naive_recurrent_policy: bool = False
    data_chunk_length: int = 10
    recurrent_n: int = 1
    activation_func: str = "relu"

    @property
    def recurrent(self) -> bool:
        return self.use_recurrent_policy or self.use_naive_recurrent_policy

    @staticmethod
    def from_reference_dicts(train: dict, model: dict, algo: dict) -> "PathConfig":
        merged = {**train, **model, **algo}
        kw = {k: merged[k] for k in PathConfig.__dataclass_fields__ if k in merged}
        cfg = PathConfig(**kw)
        set_activation(cfg.activation_func)  # the functional network code below reads it (one configuration at a time)
        return cfg


# --------------------------------------------------------------------------------------
# ValueNorm   (harl/common/valuenorm.py)
# --------------------------------------------------------------------------------------
class OracleValueNorm:
    """State = (running_mean[1], running_mean_sq[1], debiasing_term[]), beta = 0.99999."""

    def __init__(self, beta: float = 0.99999, epsilon: float = 1e-5):
        self.beta = beta
        self.epsilon = epsilon
        self.running_mean = torch.zeros(1)
        self.running_mean_sq = torch.zeros(1)
        self.debiasing_term = torch.tensor(0.0)

    def mean_var(self) -> Tuple[torch.Tensor, torch.Tensor]:  # valuenorm.py:38-45
        d = self.debiasing_term.clamp(min=self.epsilon)
        mean = self.running_mean / d
        mean_sq = self.running_mean_sq / d
        var = (mean_sq - mean**2).clamp(min=1e-2)
        return mean, var

    @torch.no_grad()
    def update(self, x: torch.Tensor) -> None:  # valuenorm.py:47-64
        x = _t(x)
        bm = x.mean(dim=0)
        bsq = (x**2).mean(dim=0)
        w = self.beta
        self.running_mean.mul_(w).add_(bm * (1.0 - w))
        self.running_mean_sq.mul_(w).add_(bsq * (1.0 - w))
        self.debiasing_term.mul_(w).add_(1.0 * (1.0 - w))

    def normalize(self, x) -> torch.Tensor:  # valuenorm.py:66-76
        x = _t(x)
        mean, var = self.mean_var()
        return (x - mean[None]) / torch.sqrt(var)[None]

    def denormalize(self, x) -> np.ndarray:  # valuenorm.py:78-92 (returns NumPy)
        x = _t(x)
        mean, var = self.mean_var()
        return (x * torch.sqrt(var)[None] + mean[None]).numpy()

    def state(self) -> Dict[str, np.ndarray]:
        return {
            "running_mean": self.running_mean.numpy().copy(),
            "running_mean_sq": self.running_mean_sq.numpy().copy(),
            "debiasing_term": self.debiasing_term.numpy().copy(),
        }

    def load_state(self, s) -> None:
        # (the reference's state is float32; the values are rounded to float32 first in every working precision)
        f = lambda k: torch.as_tensor(np.array(s[k], dtype=np.float32)).to(WORK_DTYPE)  # noqa: E731
        self.running_mean = f("running_mean").reshape(1).clone()
        self.running_mean_sq = f("running_mean_sq").reshape(1).clone()
        self.debiasing_term = f("debiasing_term").reshape(()).clone()


# Working precision.  float32 = the reference's arithmetic (what the goldens pin).  oracle/gen_noise_floor.py re-runs the
# same update in float64 to measure how far the reference's OWN fp32 results sit from exact arithmetic, case by case
# (tests/golden/noise/*.npz): the yardstick for the GPU path's tolerances on ill-conditioned quantities.
WORK_DTYPE = torch.float32


STEP_HOOK = None  # callable(net) invoked after every Adam step (sensitivity measurements only)
# callable(phase, obj, sample, vn): phase "pre" at the start of every update() (before ValueNorm moves), "post" after
# backward() and before the clip / Adam step (sensitivity measurements only: oracle/gen_noise_floor.py)
GRAD_HOOK = None


def set_work_dtype(dt) -> None:
    global WORK_DTYPE
    WORK_DTYPE = dt


def _np_work():
    return np.float64 if WORK_DTYPE == torch.float64 else np.float32


def _t(x) -> torch.Tensor:
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(WORK_DTYPE)


# --------------------------------------------------------------------------------------
# GAE / returns reverse scan   (on_policy_critic_buffer_ep.py:97-200, all 8 branches)
# --------------------------------------------------------------------------------------
def compute_returns(
    rewards: np.ndarray,      # [T, N, 1]
    value_preds: np.ndarray,  # [T+1, N, 1]   (slot T is overwritten with next_value in the GAE branches)
    masks: np.ndarray,        # [T+1, N, 1]
    bad_masks: np.ndarray,    # [T+1, N, 1]
    next_value: np.ndarray,   # [N, 1]
    gamma: float,
    gae_lambda: float,
    use_gae: bool = True,
    use_proper_time_limits: bool = True,
    value_normalizer: Optional[OracleValueNorm] = None,
    fp_order: bool = False,
) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (returns[T+1,N,1], value_preds[T+1,N,1]) as fp32 NumPy, same op order as the reference."""
    T = rewards.shape[0]
    value_preds = value_preds.copy()
    returns = np.zeros_like(value_preds)
    den = (lambda v: value_normalizer.denormalize(v)) if value_normalizer is not None else (lambda v: v)
    if use_gae:
        value_preds[-1] = next_value
        gae = 0
        for step in reversed(range(T)):
            delta = rewards[step] + gamma * den(value_preds[step + 1]) * masks[step + 1] - den(value_preds[step])
            if fp_order and use_proper_time_limits and value_normalizer is not None:
                gae = delta + gamma * gae_lambda * gae * masks[step + 1]  # on_policy_critic_buffer_fp.py:130
            else:
                gae = delta + gamma * gae_lambda * masks[step + 1] * gae
            if use_proper_time_limits:
                gae = bad_masks[step + 1] * gae
            returns[step] = gae + den(value_preds[step])
    else:
        returns[-1] = next_value
        for step in reversed(range(T)):
            if use_proper_time_limits:
                returns[step] = (returns[step + 1] * gamma * masks[step + 1] + rewards[step]) * bad_masks[
                    step + 1
                ] + (1 - bad_masks[step + 1]) * den(value_preds[step])
            else:
                returns[step] = returns[step + 1] * gamma * masks[step + 1] + rewards[step]
    return returns.astype(np.float32), value_preds


def advantages_from_returns(returns, value_preds, value_normalizer: Optional[OracleValueNorm]):
    """on_policy_ha_runner.py:26-33."""
    if value_normalizer is not None:
        return returns[:-1] - value_normalizer.denormalize(value_preds[:-1])
    return returns[:-1] - value_preds[:-1]


def normalize_advantages(adv: np.ndarray, active_masks_tm1: np.ndarray) -> np.ndarray:
    """happo.py:122-127 (EP): masked mean / population std via the NaN trick."""
    cp = adv.copy()
    cp[active_masks_tm1 == 0.0] = np.nan
    mean = np.nanmean(cp)
    std = np.nanstd(cp)
    return (adv - mean) / (std + 1e-5)


# --------------------------------------------------------------------------------------
# networks (functional; parameters keyed exactly like the reference state_dict)
# --------------------------------------------------------------------------------------
def _linear_keys(sd_keys: Sequence[str]) -> List[int]:
    idx = sorted({int(k.split(".")[3]) for k in sd_keys if k.startswith("base.mlp.fc.") and k.endswith(".weight")})
    return idx


ACTIVATION = "relu"
_ACT_FN = {"relu": F.relu, "leaky_relu": F.leaky_relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "selu": F.selu}


def set_activation(name: str) -> None:
    """activation_func of the MLP layers (models_tools.py:28-50: get_active_func; defaults as torch.nn's modules)."""
    global ACTIVATION
    assert name in _ACT_FN, name
    ACTIVATION = name


def mlp_base_forward(p: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """MLPBase: [LayerNorm(obs)] -> (Linear, activation, LayerNorm) x k   (mlp.py:25-38,64-70)."""
    if "base.feature_norm.weight" in p:
        x = F.layer_norm(x, (x.shape[-1],), p["base.feature_norm.weight"], p["base.feature_norm.bias"], 1e-5)
    idx = _linear_keys(list(p.keys()))
    # fc.{0,3,6,..} are Linear, fc.{2,5,8,..} LayerNorm (Sequential [Linear, act, LN] x k)
    lin = [i for i in idx if i % 3 == 0]
    for i in lin:
        x = F.linear(x, p[f"base.mlp.fc.{i}.weight"], p[f"base.mlp.fc.{i}.bias"])
        x = _ACT_FN[ACTIVATION](x)
        x = F.layer_norm(x, (x.shape[-1],), p[f"base.mlp.fc.{i+2}.weight"], p[f"base.mlp.fc.{i+2}.bias"], 1e-5)
    return x


def rnn_layer_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, hxs: torch.Tensor, masks: torch.Tensor):
    """RNNLayer.forward (models/base/rnn.py:23-81): nn.GRU with ``recurrent_n`` stacked layers (gate order r, z, n; layer l > 0
    reads layer l - 1's output of the same step) with EVERY layer's hidden state multiplied by the mask at every step (the
    reference multiplies at segment starts -- ``masks.repeat(1, recurrent_n)`` / ``.repeat(recurrent_n, 1, 1)``, rnn.py:27,67;
    elsewhere the mask is 1), then LayerNorm on the top layer's output.  x: [N, H] (one step) or [T*N, H] (t-major sequence);
    hxs: [N, recurrent_n, H]; masks: [N|T*N, 1].  Returns (y [T*N, H], hxs [N, recurrent_n, H])."""
    n_layers = 0
    while f"rnn.rnn.weight_ih_l{n_layers}" in p:
        n_layers += 1
    N = hxs.shape[0]
    T = x.shape[0] // N
    H = p["rnn.rnn.weight_hh_l0"].shape[1]
    xs = x.view(T, N, -1)
    ms = masks.view(T, N, 1)
    h = [hxs[:, l, :] for l in range(n_layers)]
    outs = []
    for t in range(T):
        inp = xs[t]
        for l in range(n_layers):
            Wih, Whh = p[f"rnn.rnn.weight_ih_l{l}"], p[f"rnn.rnn.weight_hh_l{l}"]
            bih, bhh = p[f"rnn.rnn.bias_ih_l{l}"], p[f"rnn.rnn.bias_hh_l{l}"]
            hl = h[l] * ms[t]
            gi = F.linear(inp, Wih, bih)
            gh = F.linear(hl, Whh, bhh)
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h[l] = (1 - z) * n + z * hl
            inp = h[l]
        outs.append(inp)
    y = torch.stack(outs, 0).reshape(T * N, H)
    y = F.layer_norm(y, (H,), p["rnn.norm.weight"], p["rnn.norm.bias"], 1e-5)
    return y, torch.stack(h, 1)


def actor_evaluate_actions(
    p: Dict[str, torch.Tensor],
    cfg: PathConfig,
    obs: torch.Tensor,
    action: torch.Tensor,
    available_actions: Optional[torch.Tensor],
    active_masks: Optional[torch.Tensor],
    rnn_states: Optional[torch.Tensor] = None,
    masks: Optional[torch.Tensor] = None,
):
    """StochasticPolicy.evaluate_actions for MLP policies (stochastic_policy.py:88-127, act.py:104-157).

    Returns (action_log_probs [B, D_a | 1], dist_entropy scalar, dist-params dict).
    """
    feat = mlp_base_forward(p, obs)
    if "rnn.rnn.weight_ih_l0" in p:
        feat, _ = rnn_layer_forward(p, feat, rnn_states, masks)
    am = active_masks if cfg.use_policy_active_masks else None
    if "act.action_outs.0.linear.weight" in p:  # MultiDiscrete -> one Categorical per head (act.py:35-43,117-141)
        lps, ents, all_logits = [], [], []
        k = 0
        while f"act.action_outs.{k}.linear.weight" in p:
            logits = F.linear(feat, p[f"act.action_outs.{k}.linear.weight"], p[f"act.action_outs.{k}.linear.bias"])
            logits = logits - logits.logsumexp(dim=-1, keepdim=True)  # torch Categorical(logits=...) normalisation
            lps.append(logits.gather(-1, action[:, k].long().unsqueeze(-1)))
            e = -(torch.clamp(logits, min=torch.finfo(logits.dtype).min) * F.softmax(logits, dim=-1)).sum(-1)  # [m]
            # act.py:126-133: entropy() is [m] and active_masks [m, 1] -> the product BROADCASTS to [m, m]
            ents.append((e * am) / am.sum() if am is not None else e / lps[-1].size(0))
            all_logits.append(logits)
            k += 1
        logp = torch.cat(lps, dim=-1).sum(dim=-1, keepdim=True)                    # act.py:134-136
        if am is not None:
            dist_entropy = torch.cat(ents, dim=-1).sum(dim=-1, keepdim=True).mean()  # act.py:137-139
        else:  # [m] pieces: cat -> [heads * m], sum(dim=-1, keepdim=True) -> [1], mean
            dist_entropy = torch.cat(ents, dim=-1).sum(dim=-1, keepdim=True).mean()
        return logp, dist_entropy, {"logits": torch.cat(all_logits, dim=-1)}
    if "act.action_out.log_std" in p:  # Box -> DiagGaussian (distributions.py:58-89)
        mean = F.linear(feat, p["act.action_out.fc_mean.weight"], p["act.action_out.fc_mean.bias"])
        std = torch.sigmoid(p["act.action_out.log_std"] / cfg.std_x_coef) * cfg.std_y_coef
        std = std.expand_as(mean)
        var = std**2
        logp = -((action - mean) ** 2) / (2 * var) - std.log() - LOG_SQRT_2PI
        ent = (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).sum(-1)
        dist = {"mean": mean, "std": std}
    else:  # Discrete -> Categorical (distributions.py:37-55)
        logits = F.linear(feat, p["act.action_out.linear.weight"], p["act.action_out.linear.bias"])
        if available_actions is not None:
            logits = torch.where(available_actions == 0, torch.full_like(logits, -1e10), logits)
        logits = logits - logits.logsumexp(dim=-1, keepdim=True)
        logp = logits.gather(-1, action.long())
        probs = F.softmax(logits, dim=-1)
        ent = -(torch.clamp(logits, min=torch.finfo(logits.dtype).min) * probs).sum(-1)
        dist = {"logits": logits}
    if am is not None:
        dist_entropy = (ent * am.squeeze(-1)).sum() / am.sum()
    else:
        dist_entropy = ent.mean()
    return logp, dist_entropy, dist


def critic_forward(p: Dict[str, torch.Tensor], share_obs: torch.Tensor, rnn_states=None, masks=None) -> torch.Tensor:
    """VNet.forward (v_net.py:48-67)."""
    feat = mlp_base_forward(p, share_obs)
    if "rnn.rnn.weight_ih_l0" in p:
        feat, _ = rnn_layer_forward(p, feat, rnn_states, masks)
    return F.linear(feat, p["v_out.weight"], p["v_out.bias"])


class _Net:
    """Parameter dict + torch.optim.Adam, in the reference's parameters() order."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], lr: float, eps: float, weight_decay: float):
        self.p = {k: v.detach().clone().to(WORK_DTYPE).requires_grad_(True) for k, v in state_dict.items()}
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr, eps=eps, weight_decay=weight_decay)

    def params(self) -> List[torch.Tensor]:
        return list(self.p.values())

    def flat(self) -> np.ndarray:
        return torch.cat([v.detach().reshape(-1) for v in self.p.values()]).numpy().copy()

    def flat_grad(self) -> np.ndarray:
        return torch.cat([v.grad.reshape(-1) for v in self.p.values()]).numpy().copy()


def _grad_norm_step(net: _Net, cfg: PathConfig) -> torch.Tensor:
    """clip_grad_norm_ | get_grad_norm, then Adam.step (happo.py:93-100, v_critic.py:148-155)."""
    if cfg.use_max_grad_norm:
        gn = torch.nn.utils.clip_grad_norm_(net.params(), cfg.max_grad_norm)
    else:
        s = 0
        for q in net.params():
            if q.grad is not None:
                s += q.grad.norm() ** 2
        gn = torch.tensor(math.sqrt(s))
    net.opt.step()
    if STEP_HOOK is not None:  # oracle/gen_noise_floor.py: rounding-level noise injected after every optimiser step
        STEP_HOOK(net)
    return gn


# --------------------------------------------------------------------------------------
# HAPPO   (happo.py)
# --------------------------------------------------------------------------------------
class OracleHAPPO:
    clipped = True

    def __init__(self, state_dict, cfg: PathConfig):
        self.cfg = cfg
        self.net = _Net(state_dict, cfg.lr, cfg.opti_eps, cfg.weight_decay)
        self.trace: List[dict] = []  # one entry per update(): loss, entropy, grad norm, ratio mean, flat grad (pre-clip)

    def evaluate_actions(self, obs, action, available_actions=None, active_masks=None, rnn_states=None, masks=None):
        return actor_evaluate_actions(
            self.net.p, self.cfg, _t(obs), _t(action),
            None if available_actions is None else _t(available_actions),
            None if active_masks is None else _t(active_masks),
            None if rnn_states is None else _t(rnn_states), None if masks is None else _t(masks),
        )

    def update(self, sample, keep_grad: bool = False):  # happo.py:28-102
        cfg = self.cfg
        obs, actions, active, old_logp, adv, avail, factor = (None if s is None else _t(s) for s in sample[:7])
        if factor is None:  # MAPPO (mappo.py:36-95): no sequential-update factor
            factor = torch.ones_like(adv)
        rnn, msk = (_t(sample[7]), _t(sample[8])) if len(sample) > 7 else (None, None)
        if GRAD_HOOK is not None:
            GRAD_HOOK("pre", self, sample, None)
        logp, ent, _ = actor_evaluate_actions(self.net.p, cfg, obs, actions, avail, active, rnn, msk)
        agg = getattr(torch, cfg.action_aggregation)
        imp = agg(torch.exp(logp - old_logp), dim=-1, keepdim=True)
        surr1 = imp * adv
        if self.clipped:
            surr2 = torch.clamp(imp, 1.0 - cfg.clip_param, 1.0 + cfg.clip_param) * adv
            surr = torch.min(surr1, surr2)
        else:  # HAA2C (haa2c.py:70-80)
            surr = surr1
        if cfg.use_policy_active_masks:
            pl = (-torch.sum(factor * surr, dim=-1, keepdim=True) * active).sum() / active.sum()
        else:
            pl = -torch.sum(factor * surr, dim=-1, keepdim=True).mean()
        self.net.opt.zero_grad()
        (pl - ent * cfg.entropy_coef).backward()
        if GRAD_HOOK is not None:
            GRAD_HOOK("post", self, sample, None)
        g = self.net.flat_grad() if keep_grad else None
        gn = _grad_norm_step(self.net, cfg)
        return pl.detach(), ent.detach(), gn.detach(), imp.detach(), g

    def train(self, buf: "OracleActorBuffer", advantages: np.ndarray, keep_grad: bool = False,
              state_type: str = "EP") -> dict:  # happo.py:104-158
        cfg = self.cfg
        info = {"policy_loss": 0.0, "dist_entropy": 0.0, "actor_grad_norm": 0.0, "ratio": 0.0}
        if np.all(buf.active_masks[:-1] == 0.0):
            return info
        if state_type == "EP":
            advantages = normalize_advantages(advantages, buf.active_masks[:-1])
        for _ in range(cfg.ppo_epoch):
            if cfg.use_recurrent_policy:
                gen = buf.recurrent_generator(advantages, cfg.actor_num_mini_batch, cfg.data_chunk_length)
            elif cfg.use_naive_recurrent_policy:
                gen = buf.naive_recurrent_generator(advantages, cfg.actor_num_mini_batch)
            else:
                gen = buf.feed_forward_generator(advantages, cfg.actor_num_mini_batch)
            for sample, idx in gen:
                pl, ent, gn, imp, g = self.update(sample, keep_grad)
                info["policy_loss"] += pl.item()
                info["dist_entropy"] += ent.item()
                info["actor_grad_norm"] += float(gn)
                info["ratio"] += float(imp.mean())
                self.trace.append(
                    {"policy_loss": pl.item(), "dist_entropy": ent.item(), "grad_norm": float(gn),
                     "ratio": float(imp.mean()), "indices": idx, "grad": g}
                )
        n = cfg.ppo_epoch * cfg.actor_num_mini_batch
        return {k: v / n for k, v in info.items()}


class OracleHAA2C(OracleHAPPO):
    """harl/algorithms/actors/haa2c.py: HAPPO with the unclipped surrogate; cfg.ppo_epoch carries a2c_epoch."""
    clipped = False


class OracleMAPPO(OracleHAPPO):
    """harl/algorithms/actors/mappo.py: the HAPPO update without the factor (update :36-95, train :97-147 are otherwise
    the same code), plus parameter sharing (share_param_train :149-234): one actor, every agent's minibatch drawn from
    its own buffer with its own randperm and concatenated into a single update."""

    def share_param_train(self, bufs: List["OracleActorBuffer"], advantages: np.ndarray, state_type: str = "EP") -> dict:
        cfg = self.cfg
        A = len(bufs)
        info = {"policy_loss": 0.0, "dist_entropy": 0.0, "actor_grad_norm": 0.0, "ratio": 0.0}
        if state_type == "EP":  # ONE mean/std over all agents' active entries (mappo.py:165-183)
            cp = np.stack([advantages.copy() for _ in range(A)])
            for a in range(A):
                cp[a][bufs[a].active_masks[:-1] == 0.0] = np.nan
            mean, std = np.nanmean(cp), np.nanstd(cp)
            adv_list = [(advantages - mean) / (std + 1e-5) for _ in range(A)]
        else:
            adv_list = [advantages[:, :, a] for a in range(A)]
        for _ in range(cfg.ppo_epoch):
            gens = []
            for a in range(A):  # generators are created (and their permutations drawn) in agent order, lazily at first next()
                if cfg.use_recurrent_policy:
                    gens.append(bufs[a].recurrent_generator(adv_list[a], cfg.actor_num_mini_batch, cfg.data_chunk_length))
                elif cfg.use_naive_recurrent_policy:
                    gens.append(bufs[a].naive_recurrent_generator(adv_list[a], cfg.actor_num_mini_batch))
                else:
                    gens.append(bufs[a].feed_forward_generator(adv_list[a], cfg.actor_num_mini_batch))
            for _mb in range(cfg.actor_num_mini_batch):
                parts = [next(g)[0] for g in gens]
                # (recurrent samples: rows are [L*m, .] l-major per agent and the reference concatenates the AGENTS on axis 0, next
                # to rnn_states [A*m]; RNNLayer.forward reads the result as (T = L, N = A*m), rnn.py:40-44 -- rnn_layer_forward
                # does the same view, so the concatenation reproduces it)
                cat = [None if parts[0][i] is None else np.concatenate([p[i] for p in parts], axis=0)
                       for i in range(len(parts[0]))]
                pl, ent, gn, imp, g = self.update(tuple(cat))
                info["policy_loss"] += pl.item()
                info["dist_entropy"] += ent.item()
                info["actor_grad_norm"] += float(gn)
                info["ratio"] += float(imp.mean())
                self.trace.append({"policy_loss": pl.item(), "dist_entropy": ent.item(), "grad_norm": float(gn),
                                   "ratio": float(imp.mean()), "indices": None, "grad": g})
        n = cfg.ppo_epoch * cfg.actor_num_mini_batch
        return {k: v / n for k, v in info.items()}


# --------------------------------------------------------------------------------------
# V critic   (v_critic.py)
# --------------------------------------------------------------------------------------
def huber(e: torch.Tensor, d: float) -> torch.Tensor:  # models_tools.py:64-68
    a = (abs(e) <= d).to(e.dtype)
    b = (abs(e) > d).to(e.dtype)
    return a * e**2 / 2 + b * d * (abs(e) - d / 2)


class OracleVCritic:
    def __init__(self, state_dict, cfg: PathConfig):
        self.cfg = cfg
        lr_cfg = cfg.critic_lr
        self.net = _Net(state_dict, lr_cfg, cfg.opti_eps, cfg.weight_decay)
        self.trace: List[dict] = []

    def get_values(self, share_obs, rnn_states=None, masks=None) -> torch.Tensor:
        return critic_forward(self.net.p, _t(share_obs), None if rnn_states is None else _t(rnn_states),
                              None if masks is None else _t(masks))

    def value_loss(self, values, value_preds, returns, vn: Optional[OracleValueNorm]):  # v_critic.py:75-114
        cfg = self.cfg
        clipped = value_preds + (values - value_preds).clamp(-cfg.clip_param, cfg.clip_param)
        if vn is not None:
            vn.update(returns)
            e_c = vn.normalize(returns) - clipped
            e_o = vn.normalize(returns) - values
        else:
            e_c = returns - clipped
            e_o = returns - values
        if cfg.use_huber_loss:
            l_c, l_o = huber(e_c, cfg.huber_delta), huber(e_o, cfg.huber_delta)
        else:
            l_c, l_o = e_c**2 / 2, e_o**2 / 2
        loss = torch.max(l_o, l_c) if cfg.use_clipped_value_loss else l_o
        return loss.mean()

    def update(self, sample, vn, keep_grad: bool = False):  # v_critic.py:116-157
        share_obs, value_preds, returns = (_t(s) for s in sample[:3])
        rnn, msk = (_t(sample[3]), _t(sample[4])) if len(sample) > 3 else (None, None)
        if GRAD_HOOK is not None:
            GRAD_HOOK("pre", self, sample, vn)
        values = critic_forward(self.net.p, share_obs, rnn, msk)
        loss = self.value_loss(values, value_preds, returns, vn)
        self.net.opt.zero_grad()
        (loss * self.cfg.value_loss_coef).backward()
        if GRAD_HOOK is not None:
            GRAD_HOOK("post", self, sample, vn)
        g = self.net.flat_grad() if keep_grad else None
        gn = _grad_norm_step(self.net, self.cfg)
        return loss.detach(), gn.detach(), g

    def train(self, buf: "OracleCriticBufferEP", vn, keep_grad: bool = False) -> dict:  # v_critic.py:159-200
        cfg = self.cfg
        info = {"value_loss": 0.0, "critic_grad_norm": 0.0}
        for _ in range(cfg.critic_epoch):
            if cfg.use_recurrent_policy:
                gen = buf.recurrent_generator(cfg.critic_num_mini_batch, cfg.data_chunk_length)
            elif cfg.use_naive_recurrent_policy:
                gen = buf.naive_recurrent_generator(cfg.critic_num_mini_batch)
            else:
                gen = buf.feed_forward_generator(cfg.critic_num_mini_batch)
            for sample, idx in gen:
                loss, gn, g = self.update(sample, vn, keep_grad)
                info["value_loss"] += loss.item()
                info["critic_grad_norm"] += float(gn)
                self.trace.append({"value_loss": loss.item(), "grad_norm": float(gn), "indices": idx, "grad": g})
        n = cfg.critic_epoch * cfg.critic_num_mini_batch
        return {k: v / n for k, v in info.items()}


# --------------------------------------------------------------------------------------
# buffers (host NumPy, reference shapes; only what train() reads)
# --------------------------------------------------------------------------------------
def chunk_rows(chunk_ids: np.ndarray, T: int, N: int, L: int) -> Tuple[np.ndarray, np.ndarray]:
    """Chunked recurrent sampling (on_policy_actor_buffer.py:255-322): arrays are cast thread-major [N*T] and chunk c is
    rows [c*L, (c+1)*L) there, i.e. thread n = (c*L)//T, start time t0 = (c*L)%T.  Returns (rows [L*m] into the
    t-major flattening row = t*N + n, ordered l-major like the reference's stack+flatten; first_rows [m] = rows of l=0)."""
    start = chunk_ids * L
    n, t0 = start // T, start % T
    rows = ((t0[None, :] + np.arange(L)[:, None]) * N + n[None, :]).reshape(-1)
    return rows, t0 * N + n


def minibatch_indices(batch_size: int, num_mini_batch: int) -> List[np.ndarray]:
    """on_policy_actor_buffer.py:121-135: one torch.randperm draw on the global CPU generator,
    remainder rows dropped."""
    assert batch_size >= num_mini_batch
    m = batch_size // num_mini_batch
    rand = torch.randperm(batch_size).numpy()
    return [rand[i * m:(i + 1) * m] for i in range(num_mini_batch)]


@dataclass
class OracleActorBuffer:
    obs: np.ndarray               # [T+1, N, D_o]
    actions: np.ndarray           # [T, N, D_a]
    action_log_probs: np.ndarray  # [T, N, D_a]
    masks: np.ndarray             # [T+1, N, 1]
    active_masks: np.ndarray      # [T+1, N, 1]
    available_actions: Optional[np.ndarray] = None  # [T+1, N, n_act] (Discrete only)
    factor: Optional[np.ndarray] = None             # [T, N, 1]
    rnn_states: Optional[np.ndarray] = None         # [T+1, N, recurrent_n, H]

    def update_factor(self, factor):
        self.factor = factor.copy()

    def _gather(self, rows, first_rows, advantages):
        T, N = self.actions.shape[:2]
        f = lambda a: a.reshape(T * N, -1)[rows]  # noqa: E731
        rnn = self.rnn_states[:-1].reshape(T * N, *self.rnn_states.shape[2:])[first_rows]
        return (f(self.obs[:-1]), f(self.actions), f(self.active_masks[:-1]), f(self.action_log_probs), f(advantages),
                None if self.available_actions is None else f(self.available_actions[:-1]),
                None if self.factor is None else f(self.factor), rnn, f(self.masks[:-1]))

    def recurrent_generator(self, advantages: np.ndarray, num_mini_batch: int, L: int):
        """on_policy_actor_buffer.py:223-326."""
        T, N = self.actions.shape[:2]
        assert T % L == 0 and (T * N) // L >= 2
        for chunks in minibatch_indices((T * N) // L, num_mini_batch):
            rows, first = chunk_rows(chunks, T, N, L)
            yield self._gather(rows, first, advantages), chunks

    def naive_recurrent_generator(self, advantages: np.ndarray, num_mini_batch: int):
        """on_policy_actor_buffer.py:180-221: whole columns, full-length sequences, rnn_states[0]."""
        T, N = self.actions.shape[:2]
        per = N // num_mini_batch
        perm = torch.randperm(N).numpy()
        for b in range(num_mini_batch):
            ids = perm[b * per:(b + 1) * per]
            rows = (np.arange(T)[:, None] * N + ids[None, :]).reshape(-1)
            yield self._gather(rows, ids, advantages), ids

    def feed_forward_generator(self, advantages: np.ndarray, num_mini_batch: int):
        T, N = self.actions.shape[:2]
        sampler = minibatch_indices(T * N, num_mini_batch)
        obs = self.obs[:-1].reshape(T * N, -1)
        actions = self.actions.reshape(T * N, -1)
        active = self.active_masks[:-1].reshape(-1, 1)
        logp = self.action_log_probs.reshape(T * N, -1)
        avail = None if self.available_actions is None else self.available_actions[:-1].reshape(T * N, -1)
        factor = None if self.factor is None else self.factor.reshape(-1, 1)
        adv = advantages.reshape(-1, 1)
        for idx in sampler:
            yield (
                obs[idx], actions[idx], active[idx], logp[idx], adv[idx],
                None if avail is None else avail[idx], None if factor is None else factor[idx],
            ), idx


@dataclass
class OracleCriticBufferEP:
    share_obs: np.ndarray    # [T+1, N, D_s]
    rewards: np.ndarray      # [T, N, 1]
    value_preds: np.ndarray  # [T+1, N, 1]
    masks: np.ndarray        # [T+1, N, 1]
    bad_masks: np.ndarray    # [T+1, N, 1]
    returns: np.ndarray = field(default=None)
    rnn_states_critic: Optional[np.ndarray] = None  # [T+1, N, recurrent_n, H]

    def _gather(self, rows, first_rows):
        B = int(np.prod(self.rewards.shape[:-1]))
        f = lambda a: a.reshape(B, -1)[rows]  # noqa: E731
        rnn = self.rnn_states_critic[:-1].reshape(B, *self.rnn_states_critic.shape[-2:])[first_rows]
        return f(self.share_obs[:-1]), f(self.value_preds[:-1]), f(self.returns[:-1]), rnn, f(self.masks[:-1])

    def _cols(self):
        """(T, number of independent columns): N for EP, N*A for FP (critic_buffer_fp.py treats every (thread, agent)
        pair as a column, ordered c = n*A + a by _ma_cast / the reshape in the naive generator)."""
        return self.rewards.shape[0], int(np.prod(self.rewards.shape[1:-1]))

    def recurrent_generator(self, num_mini_batch: int, L: int):
        """on_policy_critic_buffer_ep.py:285-369 (EP), on_policy_critic_buffer_fp.py:306-390 (FP)."""
        T, N = self._cols()
        for chunks in minibatch_indices((T * N) // L, num_mini_batch):
            rows, first = chunk_rows(chunks, T, N, L)
            yield self._gather(rows, first), chunks

    def naive_recurrent_generator(self, num_mini_batch: int):
        """on_policy_critic_buffer_ep.py:252-283, on_policy_critic_buffer_fp.py:262-304."""
        T, N = self._cols()
        per = N // num_mini_batch
        perm = torch.randperm(N).numpy()
        for b in range(num_mini_batch):
            ids = perm[b * per:(b + 1) * per]
            rows = (np.arange(T)[:, None] * N + ids[None, :]).reshape(-1)
            yield self._gather(rows, ids), ids

    fp = False  # OracleCriticBufferFP: arrays carry an agent axis [T(+1), N, A, .] (on_policy_critic_buffer_fp.py)

    def compute_returns(self, next_value, vn, cfg: PathConfig):
        self.returns, self.value_preds = compute_returns(
            self.rewards, self.value_preds, self.masks, self.bad_masks, next_value,
            cfg.gamma, cfg.gae_lambda, cfg.use_gae, cfg.use_proper_time_limits, vn, fp_order=self.fp,
        )

    def feed_forward_generator(self, num_mini_batch: int):
        B = int(np.prod(self.rewards.shape[:-1]))  # T*N (EP) or T*N*A (FP), row-major flattening
        T, N = B, 1
        sampler = minibatch_indices(B, num_mini_batch)
        so = self.share_obs[:-1].reshape(T * N, -1)
        vp = self.value_preds[:-1].reshape(-1, 1)
        rt = self.returns[:-1].reshape(-1, 1)
        for idx in sampler:
            yield (so[idx], vp[idx], rt[idx]), idx


class OracleCriticBufferFP(OracleCriticBufferEP):
    fp = True


# --------------------------------------------------------------------------------------
# the sequential update   (on_policy_ha_runner.py:11-130)
# --------------------------------------------------------------------------------------
def ha_train(
    actors: List[OracleHAPPO],
    critic: OracleVCritic,
    actor_buffers: List[OracleActorBuffer],
    critic_buffer: OracleCriticBufferEP,
    vn: Optional[OracleValueNorm],
    cfg: PathConfig,
    keep_grad: bool = False,
):
    """Returns (actor_train_infos in update order, critic_train_info, extras)."""
    T, N = actor_buffers[0].actions.shape[:2]
    A = len(actors)
    factor = np.ones((T, N, 1), dtype=_np_work())
    advantages = advantages_from_returns(critic_buffer.returns, critic_buffer.value_preds, vn)
    fp = getattr(critic_buffer, "fp", False)
    if fp:  # global advantage normalisation over all agents' active entries (on_policy_ha_runner.py:36-45)
        am = np.stack([b.active_masks for b in actor_buffers], axis=2)
        cp = advantages.copy()
        cp[am[:-1] == 0.0] = np.nan
        advantages = (advantages - np.nanmean(cp)) / (np.nanstd(cp) + 1e-5)
    order = list(range(A)) if cfg.fixed_order else list(torch.randperm(A).numpy())
    infos, factors = [], []
    for a in order:
        buf = actor_buffers[a]
        buf.update_factor(factor)
        flat = lambda v: v.reshape(T * N, -1)  # noqa: E731
        avail = None if buf.available_actions is None else flat(buf.available_actions[:-1])
        args = (flat(buf.obs[:-1]), flat(buf.actions), avail, flat(buf.active_masks[:-1]))
        if cfg.recurrent:  # rnn_states[0:1] -> full-length unroll with mask resets (on_policy_ha_runner.py:70-72)
            args = args + (buf.rnn_states[0], flat(buf.masks[:-1]))
        old_logp, _, _ = actors[a].evaluate_actions(*args)
        if fp:
            infos.append(actors[a].train(buf, advantages[:, :, a].copy(), keep_grad, state_type="FP"))
        else:
            infos.append(actors[a].train(buf, advantages.copy(), keep_grad))
        new_logp, _, _ = actors[a].evaluate_actions(*args)
        agg = getattr(torch, cfg.action_aggregation)
        factor = factor * agg(torch.exp(new_logp - old_logp), dim=-1).reshape(T, N, 1).detach().numpy()
        factors.append(factor.copy())
    cinfo = critic.train(critic_buffer, vn, keep_grad)
    return infos, cinfo, {"agent_order": [int(x) for x in order], "factors": factors, "advantages": advantages}


def ma_train(actors: List["OracleMAPPO"], critic: OracleVCritic, actor_buffers: List[OracleActorBuffer],
             critic_buffer: OracleCriticBufferEP, vn: Optional[OracleValueNorm], cfg: PathConfig, share_param: bool = False):
    """OnPolicyMARunner.train (runners/on_policy_ma_runner.py:10-64): no factor, agents in index order; with parameter
    sharing one share_param_train call and a stray ``torch.randperm(num_agents)`` draw (:42-43)."""
    A = len(actor_buffers)
    advantages = advantages_from_returns(critic_buffer.returns, critic_buffer.value_preds, vn)
    fp = getattr(critic_buffer, "fp", False)
    if fp:
        am = np.stack([b.active_masks for b in actor_buffers], axis=2)
        cp = advantages.copy()
        cp[am[:-1] == 0.0] = np.nan
        advantages = (advantages - np.nanmean(cp)) / (np.nanstd(cp) + 1e-5)
    infos = []
    if share_param:
        info = actors[0].share_param_train(actor_buffers, advantages.copy(), "FP" if fp else "EP")
        for _ in torch.randperm(A):
            infos.append(info)
    else:
        for a in range(A):
            adv_a = advantages[:, :, a].copy() if fp else advantages.copy()
            infos.append(actors[a].train(actor_buffers[a], adv_a, state_type="FP" if fp else "EP"))
    cinfo = critic.train(critic_buffer, vn)
    return infos, cinfo, {"agent_order": list(range(A)), "factors": [], "advantages": advantages}


# --------------------------------------------------------------------------------------
# HATRPO   (harl/algorithms/actors/hatrpo.py:37-247, harl/utils/trpo_util.py:5-158)
# --------------------------------------------------------------------------------------
@dataclass
class TrpoConfig:
    kl_threshold: float = 0.01
    ls_step: int = 10
    accept_ratio: float = 0.5
    backtrack_coeff: float = 0.8


def _dist_params(p, cfg: PathConfig, obs, avail, rnn=None, masks=None):
    """(kind, tensors) of the action distribution: Gaussian (mean, std) or Categorical normalised logits."""
    feat = mlp_base_forward(p, obs)
    if "rnn.rnn.weight_ih_l0" in p:
        feat, _ = rnn_layer_forward(p, feat, rnn, masks)
    if "act.action_out.log_std" in p:
        mean = F.linear(feat, p["act.action_out.fc_mean.weight"], p["act.action_out.fc_mean.bias"])
        std = (torch.sigmoid(p["act.action_out.log_std"] / cfg.std_x_coef) * cfg.std_y_coef).expand_as(mean)
        return "normal", (mean, std)
    logits = F.linear(feat, p["act.action_out.linear.weight"], p["act.action_out.linear.bias"])
    if avail is not None:
        logits = torch.where(avail == 0, torch.full_like(logits, -1e10), logits)
    return "categorical", (logits - logits.logsumexp(dim=-1, keepdim=True),)


def trpo_kl(p_new, p_old, cfg: PathConfig, obs, avail, rnn=None, masks=None) -> torch.Tensor:
    """kl_divergence (trpo_util.py:65-92): [B, 1]; Gaussian analytic KL(old || new) in float64 summed over dims,
    Categorical `kl_approx` on the normalised logits (trpo_util.py:47-51)."""
    kind, new = _dist_params(p_new, cfg, obs, avail, rnn, masks)
    with torch.no_grad():
        _, old = _dist_params(p_old, cfg, obs, avail, rnn, masks)
    if kind == "categorical":
        q, pp = new[0], old[0]
        kl = torch.exp(q - pp) - 1 - q + pp
    else:
        (mq, sq), (mp, sp) = new, old
        var_ratio = (sp.to(torch.float64) / sq.to(torch.float64)).pow(2)
        t1 = ((mp.to(torch.float64) - mq.to(torch.float64)) / sq.to(torch.float64)).pow(2)
        kl = 0.5 * (var_ratio + t1 - 1 - var_ratio.log())
    return kl.sum(1, keepdim=True)


def consume_policy_init_rng(shapes: Dict[str, Tuple[int, ...]], gain: float = 0.01) -> None:
    """HATRPO.update builds a fresh ``StochasticPolicy`` as the "old actor" snapshot (hatrpo.py:127-130); its
    construction draws from the GLOBAL CPU generator (nn.Linear's default init, then orthogonal_), so those draws
    are part of the RNG stream of train() and must be replayed for the later permutations to match."""
    relu_gain = torch.nn.init.calculate_gain("relu")
    for name, shp in shapes.items():
        if name == "rnn.rnn.weight_ih_l0":  # RNNLayer (rnn.py:8-21): nn.GRU's own uniform init, then orthogonal_ on the weights
            H = shp[1]
            n_layers = sum(1 for k_ in shapes if k_.startswith("rnn.rnn.weight_ih_l"))  # stacked layers (rnn.py:14)
            gru = torch.nn.GRU(H, H, num_layers=n_layers)
            for pn, prm in gru.named_parameters():
                if "weight" in pn:
                    torch.nn.init.orthogonal_(prm)
        elif name.startswith("rnn."):
            continue
        elif len(shp) == 2:  # every Linear, in parameters() order: hidden layers (relu gain), then the head (gain)
            lin = torch.nn.Linear(shp[1], shp[0])
            torch.nn.init.orthogonal_(lin.weight.data, gain=gain if ("action_out" in name) else relu_gain)


class OracleHATRPO:
    """update() = loss gradient -> CG(10) with Fisher-vector products (double backward) -> backtracking line search."""

    def __init__(self, state_dict, cfg: PathConfig, tcfg: TrpoConfig):
        self.cfg, self.tcfg = cfg, tcfg
        self.p = {k: v.detach().clone().to(WORK_DTYPE).requires_grad_(True) for k, v in state_dict.items()}
        self.trace: List[dict] = []

    def params(self):
        return list(self.p.values())

    def flat(self) -> torch.Tensor:
        return torch.cat([v.data.reshape(-1) for v in self.params()])

    def set_flat(self, vec: torch.Tensor) -> None:
        i = 0
        for v in self.params():
            n = v.numel()
            v.data.copy_(vec[i:i + n].view(v.shape))
            i += n

    def evaluate_actions(self, obs, action, available_actions=None, active_masks=None, rnn_states=None, masks=None):
        return actor_evaluate_actions(self.p, self.cfg, _t(obs), _t(action),
                                      None if available_actions is None else _t(available_actions),
                                      None if active_masks is None else _t(active_masks),
                                      None if rnn_states is None else _t(rnn_states), None if masks is None else _t(masks))

    def fvp(self, obs, avail, vec: torch.Tensor, rnn=None, masks=None) -> torch.Tensor:  # trpo_util.py:132-158
        old = {k: v.detach() for k, v in self.p.items()}
        kl = trpo_kl(self.p, old, self.cfg, obs, avail, rnn, masks).mean()
        g = torch.autograd.grad(kl, self.params(), create_graph=True, allow_unused=True)
        gflat = torch.cat([x.reshape(-1) for x in g if x is not None])
        hv = torch.autograd.grad((gflat * vec).sum(), self.params(), allow_unused=True)
        return torch.cat([x.contiguous().reshape(-1) for x in hv if x is not None]).data + 0.1 * vec

    def surrogate(self, obs, actions, avail, active, old_logp, adv, factor, rnn=None, masks=None):
        logp, ent, _ = actor_evaluate_actions(self.p, self.cfg, obs, actions, avail, active, rnn, masks)
        ratio = getattr(torch, self.cfg.action_aggregation)(torch.exp(logp - old_logp), dim=-1, keepdim=True)
        if self.cfg.use_policy_active_masks:
            loss = (torch.sum(ratio * factor * adv, dim=-1, keepdim=True) * active).sum() / active.sum()
        else:
            loss = torch.sum(ratio * factor * adv, dim=-1, keepdim=True).mean()
        return loss, ent, ratio

    def update(self, sample):  # hatrpo.py:37-194
        t = self.tcfg
        obs, actions, active, old_logp, adv, avail, factor = (None if s is None else _t(s) for s in sample[:7])
        rnn, msk = (_t(sample[7]), _t(sample[8])) if len(sample) > 7 else (None, None)  # recurrent generators
        loss, ent, ratio = self.surrogate(obs, actions, avail, active, old_logp, adv, factor, rnn, msk)
        g = torch.autograd.grad(loss, self.params(), allow_unused=True)
        g = torch.cat([x.reshape(-1) for x in g if x is not None]).data
        # conjugate gradient, 10 steps, residual tolerance 1e-10 (trpo_util.py:96-129)
        x = torch.zeros_like(g)
        r, pvec = g.clone(), g.clone()
        rdotr = torch.dot(r, r)
        cg_x = {}  # iterates after 1, 5, 10 steps (tests: evidence upstream of the solve's end)
        for it in range(10):
            avp = self.fvp(obs, avail, pvec, rnn, msk)
            alpha = rdotr / torch.dot(pvec, avp)
            x += alpha * pvec
            r -= alpha * avp
            new_rdotr = torch.dot(r, r)
            pvec = r + (new_rdotr / rdotr) * pvec
            rdotr = new_rdotr
            if it + 1 in (1, 5, 10):
                cg_x[it + 1] = x.numpy().copy()
            if rdotr < 1e-10:
                break
        loss0 = loss.data.numpy()
        params = self.flat().clone()
        fv = self.fvp(obs, avail, x, rnn, msk)
        shs = 0.5 * (x * fv).sum(0, keepdim=True)
        step_size = 1 / torch.sqrt(shs / t.kl_threshold)[0]
        full_step = step_size * x
        old = {k: v.detach().clone() for k, v in self.p.items()}
        consume_policy_init_rng({k: tuple(v.shape) for k, v in self.p.items()})
        expected = (g * full_step).sum(0, keepdim=True).numpy()
        flag, fraction = False, 1
        info = dict(grad=g.numpy().copy(), step_dir=x.numpy().copy(), step_size=float(step_size), shs=float(shs), cg_x=cg_x)
        for _ in range(t.ls_step):
            self.set_flat(params + fraction * full_step)
            new_loss, ent, ratio = self.surrogate(obs, actions, avail, active, old_logp, adv, factor, rnn, msk)
            improve = new_loss.data.numpy() - loss0
            kl = trpo_kl(self.p, old, self.cfg, obs, avail, rnn, msk).mean()
            if kl < t.kl_threshold and (improve / expected) > t.accept_ratio and improve.item() > 0:
                flag = True
                break
            expected = expected * t.backtrack_coeff
            fraction *= t.backtrack_coeff
        if not flag:
            self.set_flat(params)
        info.update(kl=float(kl), loss_improve=float(improve), expected_improve=float(expected[0]),
                    dist_entropy=float(ent), ratio=float(ratio.mean()), accepted=flag, fraction=float(fraction),
                    loss=float(loss0))
        self.trace.append(info)
        return info

    def train(self, buf: "OracleActorBuffer", advantages: np.ndarray, keep_grad: bool = False,
              state_type: str = "EP") -> dict:  # hatrpo.py:196-247
        out = {"kl": 0.0, "dist_entropy": 0.0, "loss_improve": 0.0, "expected_improve": 0.0, "ratio": 0.0}
        if np.all(buf.active_masks[:-1] == 0.0):
            return out
        if state_type == "EP":
            advantages = normalize_advantages(advantages, buf.active_masks[:-1])
        cfg = self.cfg
        if cfg.use_recurrent_policy:  # hatrpo.py:222-231: ONE sample holding every chunk
            gen = buf.recurrent_generator(advantages, 1, cfg.data_chunk_length)
        elif cfg.use_naive_recurrent_policy:
            gen = buf.naive_recurrent_generator(advantages, 1)
        else:
            gen = buf.feed_forward_generator(advantages, 1)
        for sample, _ in gen:
            i = self.update(sample)
            for k in out:
                out[k] += i[k]
        return out
