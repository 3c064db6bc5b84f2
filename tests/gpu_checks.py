"""GPU parity checks: every HIP kernel (through the C ABI) against the CPU oracle on the same seeded inputs.

Each ``check_*`` function returns a dict of named error figures; ``tests/test_gpu_parity.py`` asserts on them and
``tests/gpu_diag.py`` prints them all in one run (one gpurun call = full picture, GPU minutes are scarce).
Tolerances: integer / index data and the GAE scan bit-exact; fp32 losses, grad-norms, gradients 1e-5 relative
(BASELINE.json north_star).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from harl_amd import _lib
from harl_amd._lib import call, ptr, stream
from harl_amd.synthetic import Shapes, actor_param_shapes, critic_param_shapes, make_buffers, synthetic_state_dict
from oracle import harl_oracle as O
from tests.helpers import NOISE_FACTOR, GoldenCase, excess, excess_at, load_noise, rel_err, vec_excess, vec_rel_err

DEV = torch.device("cuda:0")


class Box:
    def __init__(self, shape):
        self.shape = shape


class Discrete:
    def __init__(self, n):
        self.n = n
        self.shape = ()


class MultiDiscrete:
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = (len(nvec),)


def act_space_of(sh):
    if getattr(sh, "nvec", None) is not None:
        return MultiDiscrete(sh.nvec)
    return Discrete(sh.act_dim) if sh.discrete else Box((sh.act_dim,))


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(DEV).contiguous()


def default_args(hidden, **over):
    a = dict(hidden_sizes=list(hidden), activation_func="relu", use_feature_normalization=True,
             initialization_method="orthogonal_", gain=0.01, use_naive_recurrent_policy=False,
             use_recurrent_policy=False, recurrent_n=1, data_chunk_length=10, lr=5e-4, critic_lr=5e-4, opti_eps=1e-5,
             weight_decay=0, std_x_coef=1, std_y_coef=0.5, ppo_epoch=5, critic_epoch=5, use_clipped_value_loss=True,
             clip_param=0.2, actor_num_mini_batch=1, critic_num_mini_batch=1, entropy_coef=0.01, value_loss_coef=1,
             use_max_grad_norm=True, max_grad_norm=10.0, use_gae=True, gamma=0.99, gae_lambda=0.95, use_huber_loss=True,
             use_policy_active_masks=True, huber_delta=10.0, action_aggregation="prod", share_param=False,
             fixed_order=False)
    a.update(over)
    return a


# ------------------------------------------------------------------------------------------------
def check_gae() -> Dict[str, float]:
    """All 8 compute_returns branches, two shapes (incl. a ragged column count), vs the oracle: bit-exact."""
    out = {}
    for (T, N, seed) in [(16, 6, 11), (200, 333, 5)]:
        sh = Shapes(T=T, N=N, A=1, obs_dim=4, share_obs_dim=4, act_dim=1)
        d = make_buffers(sh, seed)
        nv = (d.value_preds[-1] * 0.5).copy()
        for use_gae in (True, False):
            for ptl in (True, False):
                for use_vn in (True, False):
                    vn = None
                    stats = None
                    if use_vn:
                        vn = O.OracleValueNorm()
                        vn.load_state(dict(running_mean=-0.2 * 0.25, running_mean_sq=2.3 * 0.25, debiasing_term=0.25))
                        stats = dev(np.array([-0.2 * 0.25, 2.3 * 0.25, 0.25], dtype=np.float32))
                    ret, vp = O.compute_returns(d.rewards, d.value_preds, d.critic_masks, d.bad_masks, nv, 0.99, 0.95,
                                                use_gae, ptl, vn)
                    adv = O.advantages_from_returns(ret, vp, vn).astype(np.float32)
                    r, v, m, b = dev(d.rewards), dev(d.value_preds), dev(d.critic_masks), dev(d.bad_masks)
                    d_nv = dev(nv)
                    g_ret = torch.zeros(T + 1, N, 1, device=DEV)
                    g_adv = torch.zeros(T, N, 1, device=DEV)
                    call("harl_gae_returns", ptr(r), ptr(v), ptr(m), ptr(b), ptr(d_nv), ptr(stats), ptr(g_ret),
                         ptr(g_adv), T, N, float(np.float32(0.99)), float(np.float32(0.99 * 0.95)), int(use_gae), int(ptl),
                         0, stream())
                    torch.cuda.synchronize()
                    key = f"T{T}N{N}_gae{int(use_gae)}_ptl{int(ptl)}_vn{int(use_vn)}"
                    got = g_ret.cpu().numpy()
                    rows = slice(0, T) if use_gae else slice(0, T + 1)
                    out[key + "_returns_mismatch"] = float(np.sum(got[rows] != ret[rows]))
                    out[key + "_adv_mismatch"] = float(np.sum(g_adv.cpu().numpy() != adv))
    return out


def check_wide_input() -> Dict[str, float]:
    """Wide-observation first layer (csrc/wide.hip): the x0n ATL image + input-LayerNorm statistics against torch, and the
    split-bf16 GEMM + ReLU/LayerNorm against a float64 evaluation, for widths around every tile / chunk boundary."""
    out = {}
    rng = np.random.default_rng(11)
    H = 128
    for D, M, use_ln, gather in ((5, 45, 1, False), (18, 300, 1, True), (32, 64, 0, False), (40, 70, 1, True), (54, 200, 1, False), (64, 33, 1, False), (65, 97, 1, False), (100, 257, 1, True), (128, 64, 0, False), (200, 1000, 1, False),
                                 (393, 300, 1, True), (449, 33, 1, False), (512, 130, 1, False)):
        KP, ns = (D + 31) // 32 * 32, (M + 31) // 32
        rows = M + 40
        X = (rng.standard_normal((rows, D)) * rng.uniform(0.2, 3.0, size=(1, D)) + rng.uniform(-2, 2, size=(1, D))).astype(np.float32)
        idx = rng.permutation(rows)[:M].astype(np.int64) if gather else None
        dX, didx = dev(X), (None if idx is None else torch.from_numpy(idx).to(DEV))
        x0n = torch.full((ns * 32 * KP,), float("nan"), device=DEV)
        mu0, rstd0 = torch.empty(ns * 32, device=DEV), torch.empty(ns * 32, device=DEV)
        call("harl_mlp_x0n_wide", ptr(dX), D, ptr(didx), M, D, use_ln, ptr(x0n), ptr(mu0), ptr(rstd0), stream())
        Xg = torch.from_numpy(X[:M] if idx is None else X[idx]).double()
        ref = torch.nn.functional.layer_norm(Xg, (D,), eps=1e-5) if use_ln else Xg
        img = x0n.cpu().reshape(ns, KP // 8, 2, 32, 4)  # [slab][piece q][half h][sample][c] -> feature 32(q>>2)+8(q&3)+4h+c
        dec = torch.empty(ns * 32, KP, dtype=torch.float32)
        for q in range(KP // 8):
            for hh in range(2):
                f0 = 32 * (q >> 2) + 8 * (q & 3) + 4 * hh
                dec[:, f0:f0 + 4] = img[:, q, hh].reshape(ns * 32, 4)
        tag = f"D{D}"
        out[f"x0n_{tag}_abs"] = float((dec[:M, :D].double() - ref).abs().max())
        if KP > D:  # pad columns: zeros, except a column of ones in the last one (db' of the fused first-layer gradient)
            out[f"x0n_{tag}_pad_abs"] = float(dec[:M, D:KP - 1].abs().max()) if KP - 1 > D else 0.0
            out[f"x0n_{tag}_ones_abs"] = float((dec[:M, KP - 1] - 1.0).abs().max())
        if use_ln:
            out[f"mu0_{tag}_abs"] = float((mu0.cpu()[:M].double() - Xg.mean(1)).abs().max())
            out[f"rstd0_{tag}_rel"] = float(((rstd0.cpu()[:M].double() * torch.sqrt(Xg.var(1, unbiased=False) + 1e-5)) - 1).abs().max())
        # GEMM + epilogue
        W = (rng.standard_normal((H, D)) / np.sqrt(D)).astype(np.float32)
        b = (rng.standard_normal(H) * 0.1).astype(np.float32)
        dW, db = dev(W), dev(b)
        wimg = torch.empty(3 * H * KP // 2, device=DEV)
        xo = torch.empty(ns * 32 * H, device=DEV)
        msk = torch.empty(ns * 2 * 64, dtype=torch.int32, device=DEV)
        rs = torch.empty(ns * 32, device=DEV)
        call("harl_mlp_fwd_wide", ptr(x0n), M, KP, ptr(dW), D, ptr(db), H, ptr(wimg), ptr(xo), ptr(msk), ptr(rs), stream())
        z = torch.relu(dec[:M, :D].double() @ torch.from_numpy(W).double().T + torch.from_numpy(b).double())
        refy = torch.nn.functional.layer_norm(z, (H,), eps=1e-5)
        yi = xo.cpu().reshape(ns, H // 8, 2, 32, 4)
        y = torch.empty(ns * 32, H)
        for q in range(H // 8):
            for hh in range(2):
                f0 = 32 * (q >> 2) + 8 * (q & 3) + 4 * hh
                y[:, f0:f0 + 4] = yi[:, q, hh].reshape(ns * 32, 4)
        out[f"fwd_wide_{tag}_abs"] = float((y[:M].double() - refy).abs().max())
    return out


def check_elementwise() -> Dict[str, float]:
    out = {}
    rng = np.random.default_rng(3)
    n = 100003
    adv = rng.standard_normal(n).astype(np.float32) * 2 + 0.3
    act = (rng.random(n) > 0.2).astype(np.float32)
    ref = O.normalize_advantages(adv.reshape(-1, 1, 1), act.reshape(-1, 1, 1)).reshape(-1)
    mom = torch.zeros(3, dtype=torch.float64, device=DEV)
    d_adv, d_act = dev(adv), dev(act)  # keep references: a temporary would be recycled by the caching allocator
    call("harl_masked_moments", ptr(d_adv), ptr(d_act), n, ptr(mom), _lib.scratch("mm"), stream())
    g = torch.empty(n, device=DEV)
    call("harl_adv_normalize", ptr(d_adv), ptr(mom), ptr(g), n, stream())
    out["adv_normalize_vec_rel"] = vec_rel_err(g.cpu().numpy(), ref)
    out["moments_count_err"] = abs(mom[2].item() - act.sum())
    # factor
    for agg in ("prod", "mean"):
        D = 5
        nl = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
        ol = (rng.standard_normal((n, D)) * 0.1).astype(np.float32)
        f0 = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
        ref = f0 * getattr(torch, agg)(torch.exp(torch.from_numpy(nl) - torch.from_numpy(ol)), dim=-1).numpy()
        f, d_nl, d_ol = dev(f0), dev(nl), dev(ol)
        call("harl_factor_update", ptr(f), ptr(d_nl), ptr(d_ol), n, D, int(agg == "mean"), stream())
        out[f"factor_{agg}_rel"] = rel_err(f.cpu().numpy(), ref)
    # valuenorm
    x = (rng.standard_normal(5000) * 3 + 1).astype(np.float32)
    vn = O.OracleValueNorm()
    vn.load_state(dict(running_mean=0.15, running_mean_sq=0.85, debiasing_term=0.5))
    from harl_amd.valuenorm import ValueNorm
    gvn = ValueNorm(1, device=DEV)
    gvn.stats.copy_(dev(np.array([0.15, 0.85, 0.5], dtype=np.float32)))
    for _ in range(3):
        vn.update(x.reshape(-1, 1))
        gvn.update(dev(x))
    s = vn.state()
    ref = np.array([s["running_mean"].item(), s["running_mean_sq"].item(), s["debiasing_term"].item()])
    out["valuenorm_rel"] = rel_err(gvn.stats.cpu().numpy(), ref)
    idx = torch.from_numpy(rng.permutation(5000)[:1234].astype(np.int64))
    vn.update(x[idx.numpy()].reshape(-1, 1))
    gvn.update(dev(x), idx.to(DEV))
    s = vn.state()
    ref = np.array([s["running_mean"].item(), s["running_mean_sq"].item(), s["debiasing_term"].item()])
    out["valuenorm_gather_rel"] = rel_err(gvn.stats.cpu().numpy(), ref)
    return out


def check_adam() -> Dict[str, float]:
    out = {}
    rng = np.random.default_rng(9)
    for n, clip, scale in [(20142, True, 1.0), (70001, True, 37.0), (501, False, 0.5)]:
        p0 = rng.standard_normal(n).astype(np.float32)
        pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
        opt = torch.optim.Adam([pt], lr=5e-4, eps=1e-5, weight_decay=0)
        gp, gm, gv = dev(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        info = torch.zeros(1, dtype=torch.float64, device=DEV)
        sc = dev(np.array([scale], dtype=np.float32))
        norms = []
        for step in range(1, 6):
            g = (rng.standard_normal(n) * (3.0 if step % 2 else 0.01)).astype(np.float32)
            pt.grad = torch.from_numpy(g.copy()) * scale
            if clip:
                norms.append(float(torch.nn.utils.clip_grad_norm_([pt], 10.0)))
            else:
                norms.append(float(pt.grad.norm()))
            opt.step()
            d_g = dev(g)
            call("harl_gradnorm_clip_adam", ptr(gp), ptr(d_g), ptr(gm), ptr(gv), n, ptr(sc), int(clip), 10.0, 5e-4, 0.9,
                 0.999, 1e-5, 0.0, 1.0 - 0.9 ** step, 1.0 - 0.999 ** step, ptr(info), stream())
        out[f"adam_n{n}_param_vec_rel"] = vec_rel_err(gp.cpu().numpy(), pt.detach().numpy())
        out[f"adam_n{n}_norm_rel"] = rel_err(info.item(), sum(norms))
    return out


# ------------------------------------------------------------------------------------------------
def _is_rnn(args) -> bool:
    return bool(args["use_recurrent_policy"] or args["use_naive_recurrent_policy"])


def _mk_actor(sh: Shapes, seed: int, **over):
    from harl_amd.happo import HAPPO
    args = default_args(sh.hidden_sizes, **over)
    a = HAPPO(args, Box((sh.obs_dim,)), act_space_of(sh), device=DEV)
    sd = synthetic_state_dict(actor_param_shapes(sh, args["use_feature_normalization"], _is_rnn(args)), seed, args["std_x_coef"])
    assert list(sd.keys()) == list(a.actor.state_dict().keys()), (list(sd.keys()), list(a.actor.state_dict().keys()))
    a.actor.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return a, sd, args


def _mk_critic(sh: Shapes, seed: int, **over):
    from harl_amd.v_critic import VCritic
    args = default_args(sh.hidden_sizes, **over)
    c = VCritic(args, Box((sh.share_obs_dim,)), device=DEV)
    sd = synthetic_state_dict(critic_param_shapes(sh, args["use_feature_normalization"], _is_rnn(args)), seed)
    assert list(sd.keys()) == list(c.critic.state_dict().keys())
    c.critic.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return c, sd, args


FWD_SHAPES = [
    dict(name="mpe_box", obs_dim=18, share_obs_dim=54, act_dim=5, discrete=False, hidden_sizes=[128, 128], M=1000),
    dict(name="mpe_disc_h64", obs_dim=18, share_obs_dim=54, act_dim=5, discrete=True, hidden_sizes=[64, 64], M=77),
    dict(name="cheetah_3x128", obs_dim=23, share_obs_dim=17, act_dim=1, discrete=False, hidden_sizes=[128, 128, 128], M=4100),
    dict(name="wide_nofn_h64", obs_dim=77, share_obs_dim=70, act_dim=2, discrete=False, hidden_sizes=[64], M=257,
         over=dict(use_feature_normalization=False)),
    dict(name="humanoid_393", obs_dim=393, share_obs_dim=376, act_dim=1, discrete=False, hidden_sizes=[128, 128, 128], M=300),
    dict(name="mixed_64_128_disc14", obs_dim=40, share_obs_dim=33, act_dim=14, discrete=True, hidden_sizes=[64, 128], M=513),
    dict(name="mixed_128_64_box20", obs_dim=31, share_obs_dim=65, act_dim=20, discrete=False, hidden_sizes=[128, 64], M=640),
    # hidden width 256 (csrc/panel.hip): the reference's dexhands shapes (obs 422 / 398, 20-26 Box actions, [256, 256, 256])
    dict(name="hands_256x3_box20", obs_dim=211, share_obs_dim=200, act_dim=20, discrete=False, hidden_sizes=[256, 256, 256], M=700),
    dict(name="narrow_256x2_disc6", obs_dim=24, share_obs_dim=40, act_dim=6, discrete=True, hidden_sizes=[256, 256], M=333),
    # 33..64 inputs into 128-wide layers: the actor's fused optimiser step does not fit the LDS (harl_update_supported says no
    # -> layer kernels), its forward-only passes and the critic's step do (154 / 160 KiB)
    dict(name="obs40_box6_h128", obs_dim=40, share_obs_dim=64, act_dim=6, discrete=False, hidden_sizes=[128, 128], M=900),
    # Categorical heads on networks whose last layer runs inside the loss launch (harl_update_last_*, DISCRETE instantiations)
    dict(name="disc6_3x64", obs_dim=30, share_obs_dim=45, act_dim=6, discrete=True, hidden_sizes=[64, 64, 64], M=1500),
    dict(name="disc3_wide_2x128", obs_dim=90, share_obs_dim=70, act_dim=3, discrete=True, hidden_sizes=[128, 128], M=700),
    # remaining instantiations of the fused forward + loss launch: 64-wide with 33..64 inputs, 128-wide with a <= 4-way Gaussian head
    dict(name="obs50_box4_h64", obs_dim=50, share_obs_dim=33, act_dim=4, discrete=False, hidden_sizes=[64, 64], M=1100),
    dict(name="obs20_box2_h128", obs_dim=20, share_obs_dim=28, act_dim=2, discrete=False, hidden_sizes=[128, 128], M=800),
]


def check_forward(spec) -> Dict[str, float]:
    """log-probs (actor) and values (critic) of the MFMA forward vs the oracle's torch forward."""
    out = {}
    over = spec.get("over", {})
    M = spec["M"]
    sh = Shapes(T=M, N=1, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"])
    d = make_buffers(sh, 21, unavailable_p=0.25 if sh.discrete else 0.0)
    actor, sd, args = _mk_actor(sh, 4242, **over)
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    obs = d.obs[0][:-1].reshape(M, -1)
    act = d.actions[0].reshape(M, -1)
    avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
    p = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        ref, ent_ref, _ = O.actor_evaluate_actions(p, cfg, torch.from_numpy(obs), torch.from_numpy(act),
                                                   None if avail is None else torch.from_numpy(avail), None)
        am = torch.from_numpy((np.random.default_rng(1).random((M, 1)) > 0.3).astype(np.float32))
        _, ent_ref_m, _ = O.actor_evaluate_actions(p, cfg, torch.from_numpy(obs), torch.from_numpy(act),
                                                   None if avail is None else torch.from_numpy(avail), am)
    got, ent, dist = actor.evaluate_actions(obs, None, act, None, avail, None)
    _, ent_m, _ = actor.evaluate_actions(obs, None, act, None, avail, am.numpy())
    torch.cuda.synchronize()
    out["logp_vec_rel"] = vec_rel_err(got.cpu().numpy(), ref.numpy())
    out["entropy_rel"] = rel_err(ent.item(), float(ent_ref))            # stochastic_policy.py:88-127 returns all three
    out["entropy_active_masked_rel"] = rel_err(ent_m.item(), float(ent_ref_m))
    out["dist_logp_vec_rel"] = vec_rel_err((dist.log_prob(torch.as_tensor(act, device=DEV).squeeze(-1)).reshape(M, -1) if sh.discrete
                                            else dist.log_prob(torch.as_tensor(act, device=DEV))).cpu().numpy(), ref.numpy())
    critic, csd, _ = _mk_critic(sh, 777, **over)
    so = d.share_obs[:-1].reshape(M, -1)
    with torch.no_grad():
        vref = O.critic_forward({k: torch.from_numpy(v) for k, v in csd.items()}, torch.from_numpy(so))
    vgot, _ = critic.get_values(so, None, None)
    torch.cuda.synchronize()
    out["values_vec_rel"] = vec_rel_err(vgot.cpu().numpy(), vref.numpy())
    return out


def check_get_actions(spec) -> Dict[str, float]:
    """Rollout-side sampling: deterministic actions == distribution mode; returned log-probs == oracle log-probs of the
    returned actions; stochastic samples have the right first/second moments (Gaussian) / respect availability."""
    out = {}
    M = 4096
    sh = Shapes(T=M, N=1, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"])
    d = make_buffers(sh, 5, unavailable_p=0.3 if sh.discrete else 0.0)
    actor, sd, args = _mk_actor(sh, 31, **spec.get("over", {}))
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    obs = d.obs[0][:-1].reshape(M, -1)
    avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
    p = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        kind, dp = O._dist_params(p, cfg, torch.from_numpy(obs), None if avail is None else torch.from_numpy(avail))
    torch.manual_seed(0)
    a_det, lp_det, _ = actor.get_actions(obs, None, None, avail, deterministic=True)
    a_smp, lp_smp, _ = actor.get_actions(obs, None, None, avail, deterministic=False)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref_lp, _, _ = O.actor_evaluate_actions(p, cfg, torch.from_numpy(obs), a_smp.cpu(),
                                                None if avail is None else torch.from_numpy(avail), None)
    out["sample_logp_vec_rel"] = vec_rel_err(lp_smp.cpu().numpy(), ref_lp.numpy())
    if sh.discrete:
        mode = dp[0].argmax(-1, keepdim=True).numpy()
        out["mode_mismatch"] = float(np.sum(a_det.cpu().numpy() != mode))
        taken = a_smp.cpu().numpy().astype(np.int64)
        out["unavailable_action_sampled_count"] = float(np.sum(np.take_along_axis(avail, taken, axis=1) == 0))
    else:
        out["mode_vec_rel"] = vec_rel_err(a_det.cpu().numpy(), dp[0].numpy())
        zs = (a_smp.cpu().numpy() - dp[0].numpy()) / dp[1].numpy()
        out["sample_zscore_mean_abs"] = float(abs(zs.mean()))        # ~ N(0, 1/sqrt(M*D))
        out["sample_zscore_std_err"] = float(abs(zs.std() - 1.0))
    return out


def check_gradients(spec, mini_batches: int = 1, agg: str = "prod", inactive_p: float = 0.0) -> Dict[str, float]:
    """ONE HAPPO.update and ONE VCritic.update: pre-clip gradient vector, loss scalars, grad-norm and the
    post-Adam parameters vs the oracle (autograd + torch.optim.Adam)."""
    out = {}
    over = dict(spec.get("over", {}))
    over.update(action_aggregation=agg)
    M = spec["M"]
    nvec = spec.get("nvec")  # MultiDiscrete heads (act_dim = sum(nvec))
    sh = Shapes(T=M, N=1, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"], **({"nvec": list(nvec)} if nvec else {}))
    d = make_buffers(sh, 33, inactive_p=inactive_p, unavailable_p=0.25 if sh.discrete else 0.0)
    actor, sd, args = _mk_actor(sh, 99, **over)
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    rng = np.random.default_rng(5)
    obs = d.obs[0][:-1].reshape(M, -1)
    avail = None if (not sh.discrete or nvec) else d.available_actions[0][:-1].reshape(M, -1)
    oracle = O.OracleHAPPO({k: torch.from_numpy(v) for k, v in sd.items()}, cfg)
    # policy-consistent actions / old log-probs so that ratios straddle the clip range
    with torch.no_grad():
        feat = O.mlp_base_forward(oracle.net.p, torch.from_numpy(obs))
        if nvec:
            act = np.stack([rng.integers(0, n, size=M) for n in nvec], -1).astype(np.float32)
        elif sh.discrete and M > 20000:  # (a per-row rng.choice loop is too slow at many-slab sizes: inverse-CDF draw)
            logits = torch.nn.functional.linear(feat, oracle.net.p["act.action_out.linear.weight"], oracle.net.p["act.action_out.linear.bias"])
            logits = torch.where(torch.from_numpy(avail) == 0, torch.full_like(logits, -1e10), logits)
            cdf = np.cumsum(torch.softmax(logits, -1).numpy().astype(np.float64), -1)
            act = (rng.random((M, 1)) * cdf[:, -1:] > cdf).sum(-1).clip(0, sh.act_dim - 1).astype(np.float32)[:, None]
        elif sh.discrete:
            logits = torch.nn.functional.linear(feat, oracle.net.p["act.action_out.linear.weight"], oracle.net.p["act.action_out.linear.bias"])
            logits = torch.where(torch.from_numpy(avail) == 0, torch.full_like(logits, -1e10), logits)
            pr = torch.softmax(logits, -1).numpy().astype(np.float64)
            pr /= pr.sum(-1, keepdims=True)
            act = np.array([rng.choice(sh.act_dim, p=q) for q in pr], dtype=np.float32)[:, None]
        else:
            mean = torch.nn.functional.linear(feat, oracle.net.p["act.action_out.fc_mean.weight"], oracle.net.p["act.action_out.fc_mean.bias"]).numpy()
            std = (torch.sigmoid(oracle.net.p["act.action_out.log_std"] / cfg.std_x_coef) * cfg.std_y_coef).detach().numpy()
            act = (mean + std * rng.standard_normal(mean.shape)).astype(np.float32)
        lp, _, _ = oracle.evaluate_actions(obs, act, avail, None)
    old_logp = (lp.numpy() + 0.15 * rng.standard_normal(lp.shape)).astype(np.float32)
    adv = rng.standard_normal((M, 1)).astype(np.float32)
    factor = (1 + 0.2 * rng.standard_normal((M, 1))).astype(np.float32)
    active = d.active_masks[0][:-1].reshape(M, 1)
    sample_o = (obs, act, active, old_logp, adv, avail, factor)
    pl, ent, gn, imp, g = oracle.update(sample_o, keep_grad=True)
    taps = []
    actor._grad_tap = lambda gr, sc: taps.append((gr.clone(), sc))
    rnn = np.zeros((M, 1, 1), dtype=np.float32)
    res = actor.update((obs, rnn, act, None, active, old_logp, adv, avail, factor))
    torch.cuda.synchronize()
    gg = taps[0][0].cpu().numpy()
    if nvec:  # the arena keeps the heads of a group contiguous: compare in the reference's parameter order
        net_ = actor.actor
        gg = torch.cat([taps[0][0][net_.offsets[nm][0]:net_.offsets[nm][0] + p_.numel()] for nm, p_ in net_.named_parameters()]).cpu().numpy()
    out["actor_grad_vec_rel"] = vec_rel_err(gg, g)
    # per-parameter-tensor breakdown (largest)
    worst, off = ("", 0.0), 0
    for name, shp in actor_param_shapes(sh, args["use_feature_normalization"]):
        n = int(np.prod(shp))
        e = vec_rel_err(gg[off:off + n], g[off:off + n]) if np.max(np.abs(g[off:off + n])) > 0 else 0.0
        if e > worst[1]:
            worst = (name, e)
        off += n
    out["actor_grad_worst_tensor_rel"] = worst[1]
    out["_actor_grad_worst_tensor"] = worst[0]
    out["actor_loss_rel"] = rel_err(res[0].item(), pl.item())
    out["actor_entropy_rel"] = rel_err(res[1].item(), ent.item())
    out["actor_gradnorm_rel"] = rel_err(res[2].item(), float(gn))
    out["actor_ratio_rel"] = rel_err(res[3].item(), float(imp.mean()))
    out["actor_param_after_vec_rel"] = vec_rel_err((actor.actor.flat_reference() if nvec else actor.actor.flat_param).cpu().numpy(),
                                                   oracle.net.flat())

    # ---- critic
    critic, csd, cargs = _mk_critic(sh, 55, **over)
    oc = O.OracleVCritic({k: torch.from_numpy(v) for k, v in csd.items()}, cfg)
    so = d.share_obs[:-1].reshape(M, -1)
    with torch.no_grad():
        v0 = O.critic_forward(oc.net.p, torch.from_numpy(so)).numpy()
    vp = (v0 + 0.3 * rng.standard_normal(v0.shape)).astype(np.float32)  # straddles the value-clip range
    ret = (3.0 * rng.standard_normal(v0.shape) + 1.0).astype(np.float32)
    ret[::17] += 40.0  # push some errors past huber_delta
    ovn = O.OracleValueNorm()
    ovn.load_state(dict(running_mean=0.15, running_mean_sq=0.85, debiasing_term=0.5))
    from harl_amd.valuenorm import ValueNorm
    gvn = ValueNorm(1, device=DEV)
    gvn.stats.copy_(dev(np.array([0.15, 0.85, 0.5], dtype=np.float32)))
    loss, cgn, cg = oc.update((so, vp, ret), ovn, keep_grad=True)
    ctaps = []
    critic._grad_tap = lambda gr, sc: ctaps.append(gr.clone())
    cres = critic.update((so, rnn, vp, ret, None), gvn)
    torch.cuda.synchronize()
    out["critic_grad_vec_rel"] = vec_rel_err(ctaps[0].cpu().numpy(), cg)
    out["critic_loss_rel"] = rel_err(cres[0].item(), loss.item())
    out["critic_gradnorm_rel"] = rel_err(cres[1].item(), float(cgn))
    out["critic_param_after_vec_rel"] = vec_rel_err(critic.critic.flat_param.cpu().numpy(), oc.net.flat())
    return out


def check_trpo_rnn(spec, agg: str = "prod") -> Dict[str, float]:
    """HATRPO with a GRU policy: surrogate gradient, one Fisher-vector product through the recurrence on a random vector and
    one full update on an [L*m] recurrent sample vs the oracle (autograd double backward through the explicit GRU)."""
    from harl_amd.hatrpo import HATRPO
    from harl_amd.nets import build_seq
    out = {}
    L, m = spec["L"], spec["m"]
    M = L * m
    rn = spec.get("recurrent_n", 1)  # stacked GRU layers (rnn.py:14)
    sh = Shapes(T=L, N=m, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"], recurrent_n=rn)
    args = default_args(sh.hidden_sizes, kl_threshold=0.01, ls_step=10, accept_ratio=0.5, backtrack_coeff=0.8,
                        action_aggregation=agg, use_recurrent_policy=True, recurrent_n=rn)
    space = Discrete(sh.act_dim) if sh.discrete else Box((sh.act_dim,))
    actor = HATRPO(args, Box((sh.obs_dim,)), space, device=DEV)
    sd = synthetic_state_dict(actor_param_shapes(sh, args["use_feature_normalization"], True), 17, args["std_x_coef"])
    assert list(sd.keys()) == list(actor.actor.state_dict().keys())
    actor.actor.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    oracle = O.OracleHATRPO({k: torch.from_numpy(v) for k, v in sd.items()}, cfg, O.TrpoConfig())
    d = make_buffers(sh, 61, inactive_p=0.2, unavailable_p=0.25 if sh.discrete else 0.0, rnn=True)
    rng = np.random.default_rng(8)
    obs = d.obs[0][:-1].reshape(M, -1)
    masks = d.masks[0][:-1].reshape(M, 1)
    h0 = d.rnn["actor"][0][0]  # [m, 1, H]
    act = d.actions[0].reshape(M, -1)
    avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
    active = d.active_masks[0][:-1].reshape(M, 1)
    lp, _, _ = oracle.evaluate_actions(obs, act, avail, None, h0, masks)
    old_logp = (lp.detach().numpy() + 0.05 * rng.standard_normal(lp.shape)).astype(np.float32)
    adv = rng.standard_normal((M, 1)).astype(np.float32)
    factor = (1 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32)
    t = lambda x: None if x is None else torch.from_numpy(x)  # noqa: E731

    # ---- surrogate gradient + FVP on the recurrent layout
    H = sh.hidden_sizes[-1]
    seq = build_seq(DEV, L, m, H * rn, h0=dev(h0).reshape(m, H * rn), masks_src=dev(masks))
    Mp = L * seq["m_pad"]
    d_obs, d_act, d_old = dev(obs), dev(act), dev(old_logp)
    d_avail = None if avail is None else dev(avail)
    d_adv, d_fac, d_actv = dev(adv.reshape(M)), dev(factor.reshape(M)), dev(active.reshape(M))
    actor.actor.fold()
    sc, g = actor._surrogate(d_obs, Mp, d_act, d_avail, d_old, d_adv, None, d_fac, d_actv, want_grad=True, seq=seq)
    loss, ent, ratio = oracle.surrogate(t(obs), t(act), t(avail), t(active), t(old_logp), t(adv), t(factor), t(h0), t(masks))
    og = torch.autograd.grad(loss, oracle.params(), allow_unused=True)
    og = torch.cat([x.reshape(-1) for x in og]).numpy()
    v = rng.standard_normal(og.shape).astype(np.float32)
    ofv = oracle.fvp(t(obs), t(avail), torch.from_numpy(v), t(h0), t(masks)).numpy()
    avail_rows = d_avail if (d_avail is None or seq["idx"] is None) else d_avail[seq["idx"]].contiguous()
    gfv = actor._fvp(d_obs, Mp, M, avail_rows, dev(v), seq=seq)
    torch.cuda.synchronize()
    # the same three figures in float64: the bar is max(1e-5, 2 x the fp32 oracle's own distance from float64), as everywhere
    # else (rounds 2-3 held the GRU checks to a flat 2e-5 instead)
    O.set_work_dtype(torch.float64)
    try:
        o64 = O.OracleHATRPO({k: torch.from_numpy(v_) for k, v_ in sd.items()}, cfg, O.TrpoConfig())
        t64 = lambda x: None if x is None else torch.from_numpy(x).to(torch.float64)  # noqa: E731
        loss64, _, _ = o64.surrogate(t64(obs), t64(act), t64(avail), t64(active), t64(old_logp), t64(adv), t64(factor), t64(h0),
                                     t64(masks))
        og64 = torch.cat([x.reshape(-1) for x in torch.autograd.grad(loss64, o64.params(), allow_unused=True)]).numpy()
        ofv64 = o64.fvp(t64(obs), t64(avail), torch.from_numpy(v).to(torch.float64), t64(h0), t64(masks)).numpy()
    finally:
        O.set_work_dtype(torch.float32)
    out["_surrogate_loss_rel"] = rel_err((sc[0] / sc[1]).item(), loss.item())
    out["surrogate_loss_excess"] = excess((sc[0] / sc[1]).item(), loss.item(), loss64.item())
    out["_surrogate_grad_vec_rel"] = vec_rel_err(g.cpu().numpy(), og)
    out["surrogate_grad_vec_excess"] = vec_excess(g.cpu().numpy(), og, og64)
    out["_fvp_vec_rel"] = vec_rel_err(gfv.cpu().numpy(), ofv)
    out["fvp_vec_excess"] = vec_excess(gfv.cpu().numpy(), ofv, ofv64)

    # ---- one full update through the API-compatible entry point
    info = oracle.update((obs, act, active, old_logp, adv, avail, factor, h0, masks))
    taps = []
    actor._grad_tap = lambda g_, x_, ss: taps.append((g_.cpu().numpy(), x_.cpu().numpy(), ss))
    kl, li, ei, ent_, ratio_ = actor.update((obs, h0, act, masks, active, old_logp, adv, avail, factor))
    torch.cuda.synchronize()
    _trpo_update_excess(out, sd, cfg, (obs, act, active, old_logp, adv, avail, factor, h0, masks), info,
                        dict(step_size=taps[0][2], kl=kl, loss_improve=li, expected_improve=ei, entropy=ent_, ratio=ratio_),
                        actor.actor.flat_param.cpu().numpy(), oracle.flat().numpy(), n_pert=_TRPO_N_PERT, step_dir=taps[0][1])
    return out


TRPO_RNN_WIDE_SHAPES = [  # HATRPO through the composed GRU (gru_wide.tangent): 128-wide, and two stacked 64-wide layers
    dict(name="trpo_rnn_box_h128", obs_dim=18, share_obs_dim=54, act_dim=5, discrete=False, hidden_sizes=[128, 128], L=6, m=40),
    dict(name="trpo_rnn2_disc_h64", obs_dim=30, share_obs_dim=20, act_dim=9, discrete=True, hidden_sizes=[64], L=5, m=64,
         recurrent_n=2),
]

RNN_SHAPES = [  # (L, m): m = 40 pads to 64 sequences, m = 64 is the identity layout, m = 7 a single ragged slab
    dict(name="rnn_box_L10_m40", obs_dim=18, share_obs_dim=54, act_dim=5, discrete=False, hidden_sizes=[64, 64], L=10, m=40),
    dict(name="rnn_disc_L5_m64", obs_dim=30, share_obs_dim=20, act_dim=9, discrete=True, hidden_sizes=[64], L=5, m=64),
    dict(name="rnn_box_L25_m7_h128_64", obs_dim=70, share_obs_dim=11, act_dim=2, discrete=False, hidden_sizes=[128, 64], L=25, m=7),
    dict(name="rnn_step_m50", obs_dim=18, share_obs_dim=54, act_dim=5, discrete=False, hidden_sizes=[64, 64], L=1, m=50),
]


def check_rnn_update(spec) -> Dict[str, float]:
    """GRU policies: evaluate_actions / get_values (L-step unroll with mask resets) and ONE HAPPO.update + ONE
    VCritic.update on an [L*m] recurrent minibatch vs the oracle (autograd through the explicit GRU)."""
    out = {}
    L, m = spec["L"], spec["m"]
    M = L * m
    H = spec["hidden_sizes"][-1]
    over = dict(use_recurrent_policy=True)
    sh = Shapes(T=L, N=m, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"])
    d = make_buffers(sh, 61, inactive_p=0.2, unavailable_p=0.25 if sh.discrete else 0.0, rnn=True)
    actor, sd, args = _mk_actor(sh, 17, **over)
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    rng = np.random.default_rng(8)
    obs = d.obs[0][:-1].reshape(M, -1)          # row = l*m + j, the recurrent minibatch order
    masks = d.masks[0][:-1].reshape(M, 1)
    h0 = d.rnn["actor"][0][0]                    # [m, 1, H]
    act = d.actions[0].reshape(M, -1)
    avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
    active = d.active_masks[0][:-1].reshape(M, 1)
    oracle = O.OracleHAPPO({k: torch.from_numpy(v) for k, v in sd.items()}, cfg)
    with torch.no_grad():
        lp, _, _ = oracle.evaluate_actions(obs, act, avail, None, h0, masks)
    got, _, _ = actor.evaluate_actions(obs, h0, act, masks, avail, None)
    torch.cuda.synchronize()
    out["logp_vec_rel"] = vec_rel_err(got.cpu().numpy(), lp.numpy())
    old_logp = (lp.numpy() + 0.15 * rng.standard_normal(lp.shape)).astype(np.float32)
    adv = rng.standard_normal((M, 1)).astype(np.float32)
    factor = (1 + 0.2 * rng.standard_normal((M, 1))).astype(np.float32)
    a_sample = (obs, act, active, old_logp, adv, avail, factor, h0, masks)
    pl, ent, gn, imp, g = oracle.update(a_sample, keep_grad=True)
    # the same update in float64: every figure below is held to max(1e-5, 2 x the fp32 oracle's own distance from float64)
    # (``*_excess``; rounds 2-3 held this check to a flat 2e-5)
    O.set_work_dtype(torch.float64)
    try:
        o64 = O.OracleHAPPO({k: torch.from_numpy(v_) for k, v_ in sd.items()}, cfg)
        lp64, _, _ = o64.evaluate_actions(obs, act, avail, None, h0, masks)
        lp64 = lp64.detach().numpy()
        pl64, ent64, gn64, _, g64 = o64.update(a_sample, keep_grad=True)
    finally:
        O.set_work_dtype(torch.float32)
    out["_logp_vec_rel"] = out.pop("logp_vec_rel")
    out["logp_vec_excess"] = vec_excess(got.cpu().numpy(), lp.numpy(), lp64)
    taps = []
    actor._grad_tap = lambda gr, sc: taps.append((gr.clone(), sc))
    res = actor.update((obs, h0, act, masks, active, old_logp, adv, avail, factor))
    torch.cuda.synchronize()
    gg = taps[0][0].cpu().numpy()
    out["_actor_grad_vec_rel"] = vec_rel_err(gg, g)
    out["actor_grad_vec_excess"] = vec_excess(gg, g, g64)
    worst, off = ("", 0.0), 0
    worst_ex = 0.0
    for name, shp in actor_param_shapes(sh, args["use_feature_normalization"], True):
        n = int(np.prod(shp))
        blk = slice(off, off + n)
        if np.max(np.abs(g[blk])) > 0:
            e = vec_rel_err(gg[blk], g[blk])
            # per tensor: within the measured bar of the fp32 oracle's figure OR of the float64 one (two correct fp32
            # evaluations of a 5-element bias gradient can sit 5e-6 either side of the exact sum)
            bar = max(1e-5, NOISE_FACTOR * vec_rel_err(g[blk], g64[blk]))
            worst_ex = max(worst_ex, min(vec_rel_err(gg[blk], g[blk]), vec_rel_err(gg[blk], g64[blk])) / bar)
            out["_actor_grad_worst_tensor_vs_f64"] = max(out.get("_actor_grad_worst_tensor_vs_f64", 0.0), vec_rel_err(gg[blk], g64[blk]))
            if e > worst[1]:
                worst = (name, e)
        off += n
    out["_actor_grad_worst_tensor_rel"] = worst[1]
    out["_actor_grad_worst_tensor"] = worst[0]
    out["actor_grad_worst_tensor_excess"] = worst_ex
    out["_actor_loss_rel"] = rel_err(res[0].item(), pl.item())
    out["actor_loss_excess"] = excess(res[0].item(), pl.item(), float(pl64))
    out["_actor_entropy_rel"] = rel_err(res[1].item(), ent.item())
    out["actor_entropy_excess"] = excess(res[1].item(), ent.item(), float(ent64))
    out["_actor_gradnorm_rel"] = rel_err(res[2].item(), float(gn))
    out["actor_gradnorm_excess"] = excess(res[2].item(), float(gn), float(gn64))
    # Adam's first step is lr*g/(|g|+eps): elements whose gradient is ~eps-sized amplify fp32 rounding of g by lr/eps = 50
    # (tools/rnn_diag.py), so the post-step comparison is restricted to |g| > 1e-4; the rest must stay within one lr.
    pa, po = actor.actor.flat_param.cpu().numpy(), oracle.net.flat()
    big = np.abs(g) > 1e-4
    out["_actor_param_after_vec_rel"] = vec_rel_err(pa[big], po[big])
    out["actor_param_after_vec_excess"] = vec_excess(pa[big], po[big], o64.net.flat()[big])
    out["actor_param_after_small_g_excess"] = float(max(0.0, np.max(np.abs(pa - po)) - 2 * args["lr"]))

    critic, csd, cargs = _mk_critic(sh, 23, **over)
    oc = O.OracleVCritic({k: torch.from_numpy(v) for k, v in csd.items()}, cfg)
    so = d.share_obs[:-1].reshape(M, -1)
    ch0 = d.rnn["critic"][0]
    cmask = d.critic_masks[:-1].reshape(M, 1)
    with torch.no_grad():
        v0 = O.critic_forward(oc.net.p, torch.from_numpy(so), torch.from_numpy(ch0), torch.from_numpy(cmask)).numpy()
    vgot, hnew = critic.get_values(so, ch0, cmask)
    torch.cuda.synchronize()
    O.set_work_dtype(torch.float64)
    try:
        oc64 = O.OracleVCritic({k: torch.from_numpy(v_) for k, v_ in csd.items()}, cfg)
        f64 = lambda x: torch.from_numpy(x).to(torch.float64)  # noqa: E731
        with torch.no_grad():
            v064 = O.critic_forward(oc64.net.p, f64(so), f64(ch0), f64(cmask)).numpy()
    finally:
        O.set_work_dtype(torch.float32)
    out["_values_vec_rel"] = vec_rel_err(vgot.cpu().numpy(), v0)
    out["values_vec_excess"] = vec_excess(vgot.cpu().numpy(), v0, v064)
    if L == 1:  # single step: the returned hidden state is the GRU output before rnn.norm
        with torch.no_grad():
            feat = O.mlp_base_forward(oc.net.p, torch.from_numpy(so))
            _, href = O.rnn_layer_forward(oc.net.p, feat, torch.from_numpy(ch0), torch.from_numpy(cmask))
        O.set_work_dtype(torch.float64)
        try:
            with torch.no_grad():
                feat64 = O.mlp_base_forward(oc64.net.p, f64(so))
                _, href64 = O.rnn_layer_forward(oc64.net.p, feat64, f64(ch0), f64(cmask))
        finally:
            O.set_work_dtype(torch.float32)
        out["_hidden_state_vec_rel"] = vec_rel_err(hnew.cpu().numpy(), href.numpy())
        out["hidden_state_vec_excess"] = vec_excess(hnew.cpu().numpy(), href.numpy(), href64.numpy())
    vp = (v0 + 0.3 * rng.standard_normal(v0.shape)).astype(np.float32)
    ret = (3.0 * rng.standard_normal(v0.shape) + 1.0).astype(np.float32)
    ovn = O.OracleValueNorm()
    ovn.load_state(dict(running_mean=0.15, running_mean_sq=0.85, debiasing_term=0.5))
    from harl_amd.valuenorm import ValueNorm
    gvn = ValueNorm(1, device=DEV)
    gvn.stats.copy_(dev(np.array([0.15, 0.85, 0.5], dtype=np.float32)))
    loss, cgn, cg = oc.update((so, vp, ret, ch0, cmask), ovn, keep_grad=True)
    O.set_work_dtype(torch.float64)
    try:
        ovn64 = O.OracleValueNorm()
        ovn64.load_state(dict(running_mean=0.15, running_mean_sq=0.85, debiasing_term=0.5))
        loss64, cgn64, cg64 = oc64.update((so, vp, ret, ch0, cmask), ovn64, keep_grad=True)
    finally:
        O.set_work_dtype(torch.float32)
    ctaps = []
    critic._grad_tap = lambda gr, sc: ctaps.append(gr.clone())
    cres = critic.update((so, ch0, vp, ret, cmask), gvn)
    torch.cuda.synchronize()
    out["_critic_grad_vec_rel"] = vec_rel_err(ctaps[0].cpu().numpy(), cg)
    out["critic_grad_vec_excess"] = vec_excess(ctaps[0].cpu().numpy(), cg, cg64)
    out["_critic_loss_rel"] = rel_err(cres[0].item(), loss.item())
    out["critic_loss_excess"] = excess(cres[0].item(), loss.item(), float(loss64))
    out["_critic_gradnorm_rel"] = rel_err(cres[1].item(), float(cgn))
    out["critic_gradnorm_excess"] = excess(cres[1].item(), float(cgn), float(cgn64))
    pa, po = critic.critic.flat_param.cpu().numpy(), oc.net.flat()
    big = np.abs(cg) > 1e-4
    out["_critic_param_after_vec_rel"] = vec_rel_err(pa[big], po[big])
    out["critic_param_after_vec_excess"] = vec_excess(pa[big], po[big], oc64.net.flat()[big])
    return out


TRUNK_SPECS = [
    # the SMAC 3s5z shapes (bench.py `smac3s5z`): obs 128 / state 216 -> [64, 64, 64] -> GRU 64, Discrete(14); 300 sequences = 9.4 slabs per step
    dict(name="smac_rnn_3x64", obs_dim=128, share_obs_dim=216, act_dim=14, discrete=True, hidden_sizes=[64, 64, 64], L=10, m=300, rnn=True),
    dict(name="rnn_2x64_box3", obs_dim=100, share_obs_dim=70, act_dim=3, discrete=False, hidden_sizes=[64, 64], L=4, m=45, rnn=True),
    # feed-forward: a 12-way Gaussian head keeps the last layer out of the loss launch (every layer through the trunk launch) ...
    dict(name="ff_3x64_box12", obs_dim=200, share_obs_dim=300, act_dim=12, discrete=False, hidden_sizes=[64, 64, 64], L=1, m=1500, rnn=False),
    # ... a 5-way Categorical head takes it (harl_update_last_*): trunk launch in the log-prob passes, one-launch backward behind it
    dict(name="ff_2x64_disc5", obs_dim=70, share_obs_dim=90, act_dim=5, discrete=True, hidden_sizes=[64, 64], L=1, m=777, rnn=False),
]


def check_trunk_fused(spec) -> Dict[str, float]:
    """csrc/trunk.hip (one launch per direction for a 64-wide trunk behind a wide first layer + one launch for every weight
    gradient, the default) against the layer-by-layer launches it replaces (HARL_TRUNK_FUSED=0) on the same data and weights:
    log-probs, values, the flat gradients of ONE HAPPO.update / VCritic.update and the parameters after the step must be the same
    BITS (every stage calls the device functions of the layer kernel it stands for); a second fused run repeats itself."""
    L, m = spec["L"], spec["m"]
    M = L * m
    rnn = spec["rnn"]
    over = dict(use_recurrent_policy=True) if rnn else {}
    sh = Shapes(T=L, N=m, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"])
    d = make_buffers(sh, 61, inactive_p=0.2, unavailable_p=0.25 if sh.discrete else 0.0, rnn=rnn)
    rng = np.random.default_rng(8)
    obs = d.obs[0][:-1].reshape(M, -1)
    masks = d.masks[0][:-1].reshape(M, 1)
    h0 = d.rnn["actor"][0][0] if rnn else None
    act = d.actions[0].reshape(M, -1)
    avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
    active = d.active_masks[0][:-1].reshape(M, 1)
    so = d.share_obs[:-1].reshape(M, -1)
    ch0 = d.rnn["critic"][0] if rnn else None
    cmask = d.critic_masks[:-1].reshape(M, 1)
    adv = rng.standard_normal((M, 1)).astype(np.float32)
    factor = (1 + 0.2 * rng.standard_normal((M, 1))).astype(np.float32)
    vp = rng.standard_normal((M, 1)).astype(np.float32)
    ret = (3.0 * rng.standard_normal((M, 1)) + 1.0).astype(np.float32)
    from harl_amd.valuenorm import ValueNorm
    prev = os.environ.get("HARL_TRUNK_FUSED")
    got = {}
    old_logp = None
    try:
        for tag, mode in (("layers", "0"), ("fused", "1"), ("again", "1")):
            os.environ["HARL_TRUNK_FUSED"] = mode
            actor, _, _ = _mk_actor(sh, 17, **over)
            critic, _, _ = _mk_critic(sh, 23, **over)
            assert actor.actor.trunk_fused() == (mode == "1") and critic.critic.trunk_fused() == (mode == "1")
            lp, _, _ = actor.evaluate_actions(obs, h0, act, masks, avail, None)
            if old_logp is None:
                old_logp = (lp.cpu().numpy() + 0.15 * rng.standard_normal(tuple(lp.shape))).astype(np.float32)
            taps = []
            actor._grad_tap = lambda gr, sc: taps.append(gr.clone())
            res = actor.update((obs, h0, act, masks, active, old_logp, adv, avail, factor))
            vals, hnew = critic.get_values(so, ch0, cmask)
            gvn = ValueNorm(1, device=DEV)
            gvn.stats.copy_(dev(np.array([0.15, 0.85, 0.5], dtype=np.float32)))
            ctaps = []
            critic._grad_tap = lambda gr, sc: ctaps.append(gr.clone())
            cres = critic.update((so, ch0, vp, ret, cmask), gvn)
            lp2, _, _ = actor.evaluate_actions(obs, h0, act, masks, avail, None)  # forward-only pass with the stepped weights
            torch.cuda.synchronize()
            got[tag] = dict(logp=lp.clone(), grad=taps[0], stats=torch.stack([r_.reshape(()).double() for r_ in res[:3]]),
                            param=actor.actor.flat_param.clone(), logp_after=lp2.clone(), values=vals.clone(),
                            cgrad=ctaps[0], cstats=torch.stack([r_.reshape(()).double() for r_ in cres[:2]]),
                            cparam=critic.critic.flat_param.clone())
    finally:
        if prev is None:
            os.environ.pop("HARL_TRUNK_FUSED", None)
        else:
            os.environ["HARL_TRUNK_FUSED"] = prev
    out = {}
    a, b, c = got["layers"], got["fused"], got["again"]
    for k in a:
        out[f"{k}_mismatch"] = float(not torch.equal(a[k], b[k]))
        out[f"{k}_rerun_mismatch"] = float(not torch.equal(b[k], c[k]))
        den = float(a[k].abs().max().clamp_min(1e-30))
        out[f"_{k}_vec_rel"] = float((a[k].double() - b[k].double()).abs().max()) / den
    return out


def _perturb_one_ulp(nets, seed: int) -> None:
    """Every parameter of the oracle networks moved to a neighbouring float32, up or down at random (the perturbation of
    oracle/gen_noise_floor.py): how far the reference's fp32 figures move under the smallest change fp32 can express."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in nets:
            for v in net.p.values():
                up = torch.rand(v.shape, generator=g) < 0.5
                v.copy_(torch.where(up, torch.nextafter(v, torch.full_like(v, float("inf"))),
                                    torch.nextafter(v, torch.full_like(v, float("-inf")))))


_TRPO_N_PERT = 8
_TRPO_UPDATE_KEYS = (("step_size", "step_size"), ("kl", "kl"), ("loss_improve", "loss_improve"),
                     ("expected_improve", "expected_improve"), ("entropy", "dist_entropy"), ("ratio", "ratio"))


def _trpo_update_excess(out, sd, cfg, sample, info, got, param_after, oracle_after, tol: float = 1e-5,
                        n_pert: int = 3, step_dir=None) -> None:
    """Everything HATRPO.update() reports after the conjugate-gradient solve (step size, KL, improvements, the parameters
    after the line search) is a function of the CG solution, i.e. carries CG's amplification of rounding differences.
    Those figures get the bar of the golden tests (tests/helpers.excess): max(1e-5, NOISE_FACTOR x the reference's OWN
    uncertainty), the uncertainty measured here as its distance from the same update in fp64 and from the same fp32 update
    with every initial parameter moved by one ulp (max over n_pert draws)."""
    t = lambda d_: {k: torch.from_numpy(v) for k, v in d_.items()}  # noqa: E731
    O.set_work_dtype(torch.float64)
    try:
        o64 = O.OracleHATRPO(t(sd), cfg, O.TrpoConfig())
        i64 = o64.update(sample)
        p64 = o64.flat().numpy().astype(np.float64)
    finally:
        O.set_work_dtype(torch.float32)
    f = lambda x: float(np.asarray(x).reshape(-1)[0])  # noqa: E731
    pa = oracle_after.astype(np.float64)
    sens = {name: 0.0 for name, _ in _TRPO_UPDATE_KEYS}
    psens, dsens = 0.0, 0.0
    for k in range(n_pert):  # the spread over a few independent one-ulp perturbations (a single draw can sit near 0)
        op = O.OracleHATRPO(t(sd), cfg, O.TrpoConfig())
        _perturb_one_ulp([op], 4242 + k)
        ip = op.update(sample)
        for name, key in _TRPO_UPDATE_KEYS:
            sens[name] = max(sens[name], abs(f(ip[key]) - f(info[key])) / (abs(f(info[key])) + 1e-12))
        psens = max(psens, vec_rel_err(op.flat().numpy(), pa))
        dsens = max(dsens, vec_rel_err(ip["step_dir"], info["step_dir"]))
    if step_dir is not None:
        # the conjugate-gradient direction itself, in the symmetric form of check_trpo_upstream (rounds 2-3
        # held it to a flat 1e-4 / 2e-4): the HIP path's distance from the float64 solve against the fp32 oracle's own -- its
        # distance from float64 and how far its solution moves when its parameters move by one ulp
        d64 = np.asarray(i64["step_dir"], dtype=np.float64)
        out["_cg_step_dir_vs_f64"] = vec_rel_err(step_dir, d64)
        out["_cg_step_dir_oracle_f32_vs_f64"] = vec_rel_err(info["step_dir"], d64)
        out["_cg_step_dir_oracle_one_ulp"] = dsens
        out["_cg_step_dir_vec_rel"] = vec_rel_err(step_dir, info["step_dir"])
        out["cg_step_dir_excess"] = out["_cg_step_dir_vs_f64"] / max(1e-5, 4.0 * max(out["_cg_step_dir_oracle_f32_vs_f64"], dsens))
    for name, key in _TRPO_UPDATE_KEYS:
        out[f"_{name}_rel"] = rel_err(got[name], f(info[key]))
        out[f"_{name}_sens"] = sens[name]
        out[f"{name}_excess"] = excess(got[name], f(info[key]), f(i64[key]), sens=sens[name], tol=tol)
    out["_param_after_vec_rel"] = vec_rel_err(param_after, pa)
    out["param_after_vec_excess"] = vec_excess(param_after, pa, p64, sens=psens, tol=tol)


def check_trpo(spec, agg: str = "prod") -> Dict[str, float]:
    """HATRPO: surrogate gradient, one Fisher-vector product on a random vector, and one full update (CG + line
    search) vs the oracle (autograd double backward), from identical parameters and data."""
    from harl_amd.hatrpo import HATRPO
    out = {}
    M = spec["M"]
    sh = Shapes(T=M, N=1, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"])
    over = dict(spec.get("over", {}))
    args = default_args(sh.hidden_sizes, kl_threshold=0.01, ls_step=10, accept_ratio=0.5, backtrack_coeff=0.8,
                        action_aggregation=agg, **over)
    space = Discrete(sh.act_dim) if sh.discrete else Box((sh.act_dim,))
    actor = HATRPO(args, Box((sh.obs_dim,)), space, device=DEV)
    sd = synthetic_state_dict(actor_param_shapes(sh, args["use_feature_normalization"]), 17, args["std_x_coef"])
    actor.actor.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    oracle = O.OracleHATRPO({k: torch.from_numpy(v) for k, v in sd.items()}, cfg, O.TrpoConfig())
    d = make_buffers(sh, 44, inactive_p=0.1, unavailable_p=0.25 if sh.discrete else 0.0)
    rng = np.random.default_rng(8)
    obs = d.obs[0][:-1].reshape(M, -1)
    avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
    with torch.no_grad():
        kind, dp = O._dist_params(oracle.p, cfg, torch.from_numpy(obs), None if avail is None else torch.from_numpy(avail))
    if sh.discrete:
        pr = torch.exp(dp[0]).numpy().astype(np.float64)
        pr /= pr.sum(-1, keepdims=True)
        act = np.array([rng.choice(sh.act_dim, p=q) for q in pr], dtype=np.float32)[:, None]
    else:
        act = (dp[0].numpy() + dp[1].numpy() * rng.standard_normal(dp[0].shape)).astype(np.float32)
    lp, _, _ = oracle.evaluate_actions(obs, act, avail, None)
    old_logp = (lp.detach().numpy() + 0.05 * rng.standard_normal(lp.shape)).astype(np.float32)
    adv = rng.standard_normal((M, 1)).astype(np.float32)
    factor = (1 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32)
    active = d.active_masks[0][:-1].reshape(M, 1)

    # ---- Fisher-vector product on a random direction (primal activations come from a surrogate evaluation)
    d_obs, d_act, d_old = dev(obs), dev(act), dev(old_logp)
    d_avail = None if avail is None else dev(avail)
    d_adv, d_fac, d_actv = dev(adv.reshape(M)), dev(factor.reshape(M)), dev(active.reshape(M))
    actor.actor.fold()
    sc, g = actor._surrogate(d_obs, M, d_act, d_avail, d_old, d_adv, None, d_fac, d_actv, want_grad=True)
    t = lambda x: None if x is None else torch.from_numpy(x)  # noqa: E731
    loss, ent, ratio = oracle.surrogate(t(obs), t(act), t(avail), t(active), t(old_logp), t(adv), t(factor))
    og = torch.autograd.grad(loss, oracle.params(), allow_unused=True)
    og = torch.cat([x.reshape(-1) for x in og]).numpy()
    out["surrogate_loss_rel"] = rel_err((sc[0] / sc[1]).item(), loss.item())
    out["surrogate_grad_vec_rel"] = vec_rel_err(g.cpu().numpy(), og)
    v = rng.standard_normal(og.shape).astype(np.float32)
    ofv = oracle.fvp(t(obs), t(avail), torch.from_numpy(v)).numpy()
    gfv = actor._fvp(d_obs, M, M, d_avail, dev(v))
    torch.cuda.synchronize()
    out["fvp_vec_rel"] = vec_rel_err(gfv.cpu().numpy(), ofv)

    # ---- one full update
    info = oracle.update((obs, act, active, old_logp, adv, avail, factor))
    taps = []
    actor._grad_tap = lambda g_, x_, ss: taps.append((g_.cpu().numpy(), x_.cpu().numpy(), ss))
    rnn = np.zeros((M, 1, 1), dtype=np.float32)
    kl, li, ei, ent_, ratio_ = actor.update((obs, rnn, act, None, active, old_logp, adv, avail, factor))
    torch.cuda.synchronize()
    _trpo_update_excess(out, sd, cfg, (obs, act, active, old_logp, adv, avail, factor), info,
                        dict(step_size=taps[0][2], kl=kl, loss_improve=li, expected_improve=ei, entropy=ent_, ratio=ratio_),
                        actor.actor.flat_param.cpu().numpy(), oracle.flat().numpy(), step_dir=taps[0][1])
    return out


def check_multidiscrete_rollout(nvec, hidden) -> Dict[str, float]:
    """MultiDiscrete rollout side: evaluate_actions (summed log-probs, entropy) and get_actions (deterministic = per-head
    argmax; sampled actions in range with the log-probs of the drawn actions) against the oracle's heads."""
    from harl_amd.happo import HAPPO
    out = {}
    M = 203
    sh = Shapes(T=M, N=1, A=1, obs_dim=19, share_obs_dim=7, act_dim=sum(nvec), hidden_sizes=hidden, nvec=list(nvec))
    args = default_args(sh.hidden_sizes)
    actor = HAPPO(args, Box((sh.obs_dim,)), MultiDiscrete(nvec), device=DEV)
    sd = synthetic_state_dict(actor_param_shapes(sh, args["use_feature_normalization"]), 23, args["std_x_coef"])
    assert list(sd.keys()) == list(actor.actor.state_dict().keys())
    actor.actor.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    d = make_buffers(sh, 9, inactive_p=0.2)
    obs = d.obs[0][:-1].reshape(M, -1)
    act = d.actions[0].reshape(M, -1)
    active = d.active_masks[0][:-1].reshape(M, 1)
    p = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        olp, oent, odist = O.actor_evaluate_actions(p, cfg, torch.from_numpy(obs), torch.from_numpy(act), None,
                                                    torch.from_numpy(active))
    lp, ent, dist = actor.evaluate_actions(obs, None, act, None, None, active)
    out["logp_vec_rel"] = vec_rel_err(lp.cpu().numpy(), olp.numpy())
    out["entropy_rel"] = rel_err(ent.item(), oent.item())
    out["dist_is_none_mismatch"] = float(dist is not None)
    ologits = odist["logits"].numpy()  # normalised logits of every head, concatenated
    a_det, lp_det, _ = actor.get_actions(obs, None, None, None, deterministic=True)
    lo, want, wlp = 0, [], 0.0
    for n in nvec:
        blk = ologits[:, lo:lo + n]
        want.append(blk.argmax(-1))
        wlp = wlp + blk.max(-1)
        lo += n
    # (ties between fp32 logits of two implementations are not expected on random weights)
    out["deterministic_action_mismatch"] = float(np.sum(a_det.cpu().numpy() != np.stack(want, -1).astype(np.float32)))
    out["deterministic_logp_vec_rel"] = vec_rel_err(lp_det.cpu().numpy().reshape(-1), wlp)
    torch.manual_seed(5)
    a_s, lp_s, _ = actor.get_actions(obs, None, None, None)
    a_np = a_s.cpu().numpy()
    out["sample_range_mismatch"] = float(np.sum((a_np < 0) | (a_np >= np.asarray(nvec)[None, :]) | (a_np != np.round(a_np))))
    lo, wlp = 0, 0.0
    for k, n in enumerate(nvec):
        wlp = wlp + np.take_along_axis(ologits[:, lo:lo + n], a_np[:, k:k + 1].astype(np.int64), -1)[:, 0]
        lo += n
    out["sample_logp_vec_rel"] = vec_rel_err(lp_s.cpu().numpy().reshape(-1), wlp)
    out["shape_mismatch"] = float(tuple(a_s.shape) != (M, len(nvec)) or tuple(lp_s.shape) != (M, 1))
    return out


# ------------------------------------------------------------------------------------------------
def build_runner(case: GoldenCase):
    from harl_amd.runner import RUNNER_REGISTRY
    OnPolicyHARunner = RUNNER_REGISTRY[case.algo_name]
    train, model, algo = case.reference_dicts()
    sh, d = case.shapes, case.data
    space = act_space_of(sh)
    algo_args = dict(train=train, model=model, algo=algo)
    r = OnPolicyHARunner(dict(algo=case.algo_name), algo_args, dict(state_type=case.state_type),
                         obs_spaces=[Box((sh.obs_dim,))] * sh.A, share_obs_space=Box((sh.share_obs_dim,)),
                         act_spaces=[space] * sh.A, device=DEV)
    for a in range(sh.A):
        if a == 0 or not getattr(case, "share_param", False):
            r.actor[a].actor.load_state_dict({k: torch.from_numpy(v) for k, v in case.actor_sd[a].items()})
        b = r.actor_buffer[a]
        b.obs.copy_(dev(d.obs[a]))
        b.actions.copy_(dev(d.actions[a]))
        b.action_log_probs.copy_(dev(d.action_log_probs[a]))
        b.masks.copy_(dev(d.masks[a]))
        b.active_masks.copy_(dev(d.active_masks[a]))
        if sh.discrete:
            b.available_actions.copy_(dev(d.available_actions[a]))
        if d.rnn is not None:
            b.rnn_states.copy_(dev(d.rnn["actor"][a]))
    if d.rnn is not None:
        r.critic_buffer.rnn_states_critic.copy_(dev(d.rnn["critic_fp" if case.state_type == "FP" else "critic"]))
    r.critic.critic.load_state_dict({k: torch.from_numpy(v) for k, v in case.critic_sd.items()})
    cb = r.critic_buffer
    if case.state_type == "FP":
        for k in ("share_obs", "rewards", "value_preds", "masks", "bad_masks"):
            getattr(cb, k).copy_(dev(d.fp[k]))
    else:
        cb.share_obs.copy_(dev(d.share_obs))
        cb.rewards.copy_(dev(d.rewards))
        cb.value_preds.copy_(dev(d.value_preds))
        cb.masks.copy_(dev(d.critic_masks))
        cb.bad_masks.copy_(dev(d.bad_masks))
    if r.value_normalizer is not None:
        vi = case.vn_init
        r.value_normalizer.stats.copy_(dev(np.array([vi["running_mean"], vi["running_mean_sq"], vi["debiasing_term"]],
                                                    dtype=np.float32)))
    return r


def check_train_golden(name: str) -> Dict[str, float]:
    """compute_returns + OnPolicyHARunner.train() against the golden vectors recorded from the REAL reference."""
    case = GoldenCase(name)
    z = case.z
    out = {}
    torch.manual_seed(case.seed)
    np.random.seed(case.seed)
    r = build_runner(case)
    from harl_amd.buffers import _advance_matches_randperm
    assert _advance_matches_randperm()  # run the one-time RNG self-check outside the recording window
    perms: List[np.ndarray] = []
    real = torch.randperm

    def rec(n, *a, **k):
        p = real(n, *a, **k)
        perms.append(p.numpy().copy())
        return p

    torch.manual_seed(case.seed + 12345)
    torch.randperm = rec
    try:
        cb = r.critic_buffer
        cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
        torch.cuda.synchronize()
        use_gae = case.algo["use_gae"]
        T = case.shapes.T
        out["returns_mismatch"] = float(np.sum(cb.returns.cpu().numpy()[:T] != z["returns"][:T]))
        out["advantages_mismatch"] = float(np.sum(cb.advantages.cpu().numpy() != z["advantages"]))
        r.prep_training()
        traced = case.algo_name != "hatrpo"
        if traced:  # per-update statistics (the goldens carry the reference's actor_trace / critic_trace)
            uniq = []
            for a_ in r.actor:
                if not any(a_ is u for u in uniq):
                    uniq.append(a_)
                    a_._trace = []
            r.critic._trace = []
        infos, cinfo = r.train()
        torch.cuda.synchronize()
    finally:
        torch.randperm = real
    # Integer side.  (1) The CPU generator must end in exactly the state the reference's draws leave it in
    # (replay the recorded reference permutations from the same seed).  (2) Every permutation this implementation
    # materialised (agent order; minibatch permutations when num_mini_batch > 1 -- with a single minibatch only the
    # generator state is replayed, see buffers.consume_randperm) must be bit-identical to the reference's draw at the
    # same position of the stream.
    gp = case.perms()
    state_after = torch.get_rng_state()
    torch.manual_seed(case.seed + 12345)
    replay_bad = 0
    n_actor_draws = 0
    for gi, g in enumerate(gp):
        replay_bad += int(not np.array_equal(torch.randperm(len(g)).numpy(), g))
        if case.algo_name == "hatrpo":
            # HATRPO.update constructs a fresh policy ("old actor") per agent, which draws from the same generator
            # (hatrpo.py:127-130): one such construction after each of the A per-agent minibatch permutations
            is_order = (gi == 0 and not case.algo["fixed_order"])
            if not is_order and n_actor_draws < case.shapes.A:
                n_actor_draws += 1
                from harl_amd.nets import consume_policy_init_rng
                sp = Discrete(case.shapes.act_dim) if case.shapes.discrete else Box((case.shapes.act_dim,))
                consume_policy_init_rng({**case.model, **case.algo}, Box((case.shapes.obs_dim,)), sp)
    out["golden_replay_mismatch"] = float(replay_bad)
    out["rng_state_mismatch"] = float(not torch.equal(state_after, torch.get_rng_state()))
    # (each materialised permutation must BE one of the reference's draws, each of those used at most once.  Not "in stream
    # order": since round 6 the critic's samplers draw -- from the generator state the reference's critic finds, the actors'
    # draws fast-forwarded -- before the actors' do, so the critic's permutations are materialised first.  A permutation of a
    # few thousand elements equals the reference's draw of some position only if it was drawn from that position's state.)
    used, bad = [False] * len(gp), 0
    for pm in perms:
        k = next((i for i in range(len(gp)) if not used[i] and len(gp[i]) == len(pm) and np.array_equal(gp[i], pm)), None)
        if k is None:
            bad += 1
        else:
            used[k] = True
    out["perm_mismatch"] = float(bad)
    gold = z["actor_infos"]
    nz = load_noise(name)  # the same update in float64 (oracle/gen_noise_floor.py): how exact the reference's own figures are
    assert nz is not None, f"tests/golden/noise/{name}.npz missing: run oracle/gen_noise_floor.py"
    if case.algo_name == "hatrpo":
        got = np.array([[i["kl"], i["loss_improve"], i["expected_improve"], i["dist_entropy"], i["ratio"]] for i in infos])
        for c, nm in enumerate(("kl", "loss_improve", "expected_improve", "entropy", "ratio")):
            out[f"_actor_{nm}_rel"] = rel_err(got[:, c], gold[:, c])
            out[f"actor_{nm}_excess"] = excess(got[:, c], gold[:, c], nz["actor_infos"][:, c], nz["sens_actor_infos"][:, c])
    else:
        got = np.array([[i["policy_loss"], i["dist_entropy"], i["actor_grad_norm"], i["ratio"]] for i in infos])
        for c, nm in enumerate(("policy_loss", "entropy", "gradnorm", "ratio")):
            out[f"_actor_{nm}_rel"] = rel_err(got[:, c], gold[:, c])
            out[f"actor_{nm}_excess"] = excess(got[:, c], gold[:, c], nz["actor_infos"][:, c], nz["sens_actor_infos"][:, c])
    if traced and "actor_trace" in z.files and not getattr(case, "share_param", False):
        # per-update parity: the k-th optimiser step of every agent against the reference's k-th step of that agent
        # (policy_loss, dist_entropy, grad_norm, ratio), and the critic's steps (value_loss, grad_norm)
        gt, nt, st = z["actor_trace"], nz["actor_trace"], nz["sens_actor_trace"]
        first, worst, wkey, exc, exc_at = 0.0, 0.0, "", 0.0, ""
        for a in range(case.shapes.A):
            ga, na, sa = gt[gt[:, 0] == a][:, 1:], nt[nt[:, 0] == a][:, 1:], st[nt[:, 0] == a][:, 1:]
            tr = r.actor[a]._trace
            if not tr or len(ga) == 0:
                continue
            cum = torch.stack(tr).double().cpu().numpy()
            per = np.diff(np.concatenate([np.zeros((1, cum.shape[1])), cum]), axis=0)[:, :4]
            n = min(len(per), len(ga))
            e = np.abs(per[:n] - ga[:n]) / (np.abs(ga[:n]) + 1e-12)
            first = max(first, float(e[0].max()))
            out.setdefault("_trace_policy_loss_err", {})[a] = " ".join(f"{x:.1e}" for x in e[:, 0])
            out.setdefault("_trace_gradnorm_err", {})[a] = " ".join(f"{x:.1e}" for x in e[:, 2])
            out.setdefault("_trace_policy_loss_ref", {})[a] = " ".join(f"{x:.4g}" for x in ga[:n, 0])
            ex_a = excess(per[:n], ga[:n], na[:n], sa[:n])
            if ex_a > exc:
                exc, exc_at = ex_a, f"agent{a} " + excess_at(per[:n], ga[:n], na[:n], sa[:n])
            if float(e.max()) > worst:
                k, c = np.unravel_index(np.argmax(e), e.shape)
                worst, wkey = float(e.max()), f"agent{a}/update{k}/{('policy_loss', 'entropy', 'grad_norm', 'ratio')[c]}"
        out["_actor_trace_first_update_rel"] = first
        out["_actor_trace_max_rel"] = worst
        out["_actor_trace_worst"] = wkey
        out["actor_trace_excess"] = exc
        out["_actor_trace_excess_at"] = exc_at
        gc = z["critic_trace"]
        cum = torch.stack(r.critic._trace).double().cpu().numpy()
        per = np.diff(np.concatenate([np.zeros((1, cum.shape[1])), cum]), axis=0)[:, :2]
        n = min(len(per), len(gc))
        e = np.abs(per[:n] - gc[:n]) / (np.abs(gc[:n]) + 1e-12)
        out["_critic_trace_first_update_rel"] = float(e[0].max())
        out["_critic_trace_max_rel"] = float(e.max())
        out["critic_trace_excess"] = excess(per[:n], gc[:n], nz["critic_trace"][:n], nz["sens_critic_trace"][:n])
    cg = [cinfo["value_loss"], cinfo["critic_grad_norm"]]
    out["_critic_value_loss_rel"] = rel_err(cg[0], z["critic_info"][0])
    out["_critic_gradnorm_rel"] = rel_err(cg[1], z["critic_info"][1])
    out["critic_info_excess"] = excess(cg, z["critic_info"], nz["critic_info"], nz["sens_critic_info"])
    for a in range(case.shapes.A):
        fp = r.actor[a].actor.flat_reference().cpu().numpy()
        out[f"_actor{a}_final_param_vec_rel"] = vec_rel_err(fp, z[f"actor_final_{a}"])
        out[f"actor{a}_final_param_excess"] = vec_excess(fp, z[f"actor_final_{a}"], nz[f"actor_final_{a}"], nz[f"sens_actor_final_{a}"])
    fp = r.critic.critic.flat_param.cpu().numpy()
    out["_critic_final_param_vec_rel"] = vec_rel_err(fp, z["critic_final"])
    out["critic_final_param_excess"] = vec_excess(fp, z["critic_final"], nz["critic_final"], nz["sens_critic_final"])
    if r.value_normalizer is not None:
        out["vn_final_rel"] = rel_err(r.value_normalizer.stats.cpu().numpy(), z["vn_final"])
    return out


def check_rollout_learning(discrete: bool, recurrent: bool = False) -> Dict[str, float]:
    """OnPolicyHARunner.run() end to end on a toy vectorised environment: collect (device-side sampling) -> env.step ->
    insert -> compute -> train -> after_update.  The policy must actually learn (mean step reward improves), buffer
    bookkeeping must follow the reference's insert() rules (masks / active_masks / bad_masks / hidden-state resets)."""
    from harl_amd.runner import OnPolicyHARunner
    from tests.fake_env import FakeVecEnv
    torch.manual_seed(3)
    np.random.seed(3)
    N, T = 256, 50
    env = FakeVecEnv(N, n_agents=3, state_dim=6, act_dim=4 if discrete else 2, discrete=discrete, horizon=25, seed=5)
    a = default_args([64, 64], lr=3e-3, critic_lr=3e-3, use_recurrent_policy=recurrent, data_chunk_length=10,
                     actor_num_mini_batch=1, critic_num_mini_batch=1)
    train = dict(n_rollout_threads=N, episode_length=T, use_valuenorm=True, use_proper_time_limits=True,
                 use_linear_lr_decay=False, num_env_steps=N * T * 14, log_interval=1)
    model = {k: a[k] for k in ("hidden_sizes", "activation_func", "use_feature_normalization", "initialization_method",
                                "gain", "use_naive_recurrent_policy", "use_recurrent_policy", "recurrent_n",
                                "data_chunk_length", "lr", "critic_lr", "opti_eps", "weight_decay", "std_x_coef", "std_y_coef")}
    algo = {k: v for k, v in a.items() if k not in model}
    r = OnPolicyHARunner(dict(algo="happo"), dict(train=train, model=model, algo=algo), dict(state_type="EP"), envs=env,
                         device=DEV)
    hist = r.run()
    torch.cuda.synchronize()
    rew = [h[2] for h in hist]
    out = {"reward_first": float(np.mean(rew[:2])), "reward_last": float(np.mean(rew[-2:]))}
    out["_improvement"] = out["reward_last"] - out["reward_first"]
    finite = all(np.isfinite(list(i.values())).all() for h in hist for i in h[0]) and all(np.isfinite(list(h[1].values())).all() for h in hist)
    out["nonfinite_count"] = 0.0 if finite else 1.0
    # bookkeeping of the last episode's buffers (before after_update rolled slot T into slot 0 they were checked in run)
    b1, cb = r.actor_buffer[1], r.critic_buffer
    out["mask_zero_frac"] = float((cb.masks[1:] == 0).float().mean().item())          # one reset per 25 steps
    out["active_zero_frac_agent1"] = float((b1.active_masks[1:] == 0).float().mean().item())  # 3 dead steps per 25
    out["bad_zero_frac"] = float((cb.bad_masks[1:] == 0).float().mean().item())
    return out


def check_checkpoint_compat(tmpdir: str) -> Dict[str, float]:
    """restore() must load the files the REFERENCE's save() wrote (tests/golden/ref_ckpt, produced by oracle/gen_golden.py
    from on_policy_base_runner.py:724-740): per-agent actor (GRU + Categorical head), critic, CPU ValueNorm (3 keys);
    save() must write files with the same names / keys / shapes that plain torch (and hence the reference) loads back."""
    import os
    from harl_amd.runner import OnPolicyHARunner
    from tests.helpers import GOLDEN_DIR
    out = {}
    sh = Shapes(T=4, N=4, A=2, obs_dim=9, share_obs_dim=12, act_dim=4, discrete=True, hidden_sizes=[64, 64])
    a = default_args([64, 64], use_recurrent_policy=True)
    model_keys = ("hidden_sizes", "activation_func", "use_feature_normalization", "initialization_method", "gain",
                  "use_naive_recurrent_policy", "use_recurrent_policy", "recurrent_n", "data_chunk_length", "lr", "critic_lr",
                  "opti_eps", "weight_decay", "std_x_coef", "std_y_coef")
    model = {k: a[k] for k in model_keys}
    algo = {k: v for k, v in a.items() if k not in model}
    train = dict(n_rollout_threads=sh.N, episode_length=sh.T, use_valuenorm=True, use_proper_time_limits=True)
    r = OnPolicyHARunner(dict(algo="happo"), dict(train=train, model=model, algo=algo), dict(state_type="EP"),
                         obs_spaces=[Box((sh.obs_dim,))] * sh.A, share_obs_space=Box((sh.share_obs_dim,)),
                         act_spaces=[Discrete(sh.act_dim)] * sh.A, device=DEV)
    ref_dir = os.path.join(GOLDEN_DIR, "ref_ckpt")
    r.restore(ref_dir)
    for ag in range(sh.A):
        want = synthetic_state_dict(actor_param_shapes(sh, True, True), 500 + ag, a["std_x_coef"])
        got = r.actor[ag].actor.state_dict()
        out[f"actor{ag}_key_mismatch"] = float(list(got.keys()) != list(want.keys()))
        out[f"actor{ag}_restore_max_abs"] = float(max(np.max(np.abs(got[k].cpu().numpy() - v)) for k, v in want.items()))
    want = synthetic_state_dict(critic_param_shapes(sh, True, True), 599)
    got = r.critic.critic.state_dict()
    out["critic_key_mismatch"] = float(list(got.keys()) != list(want.keys()))
    out["critic_restore_max_abs"] = float(max(np.max(np.abs(got[k].cpu().numpy() - v)) for k, v in want.items()))
    out["vn_restore_max_abs"] = float(np.max(np.abs(r.value_normalizer.stats.cpu().numpy() - np.array([0.25, 1.5, 0.75]))))
    # the restored parameters must be live in the kernels: folded weights refreshed -> a forward pass differs from init
    obs = np.random.default_rng(0).standard_normal((8, sh.obs_dim)).astype(np.float32)
    v1, _ = r.critic.get_values(np.random.default_rng(1).standard_normal((8, sh.share_obs_dim)).astype(np.float32),
                                np.zeros((8, 1, 64), np.float32), np.ones((8, 1), np.float32))
    p = {k: torch.from_numpy(v) for k, v in want.items()}
    O.set_activation(a.get("activation_func", "relu"))  # (the oracle's functional code reads a module global: not the previous test's)
    vref = O.critic_forward(p, torch.from_numpy(np.random.default_rng(1).standard_normal((8, sh.share_obs_dim)).astype(np.float32)),
                            torch.zeros(8, 1, 64), torch.ones(8, 1))
    out["values_after_restore_vec_rel"] = vec_rel_err(v1.cpu().numpy(), vref.detach().numpy())
    # ---- save(): same file names, keys, shapes, values; loadable by plain torch on the CPU
    r.save(tmpdir)
    for name in ("actor_agent0.pt", "actor_agent1.pt", "critic_agent.pt", "value_normalizer.pt"):
        mine = torch.load(os.path.join(tmpdir, name), map_location="cpu")
        ref = torch.load(os.path.join(ref_dir, name), map_location="cpu")
        out[f"{name}_key_mismatch"] = float(list(mine.keys()) != list(ref.keys()))
        out[f"{name}_shape_mismatch"] = float(any(tuple(mine[k].shape) != tuple(ref[k].shape) for k in ref))
        out[f"{name}_roundtrip_max_abs"] = float(max(float((mine[k].float() - ref[k].float()).abs().max()) for k in ref))
    return out


def check_full_size_properties() -> Dict[str, float]:
    """BASELINE.json configs[1] sizes (T=200, N=4096, 3 agents, obs 18 / share_obs 54 / Box 5, MLP [128,128]) -- too big for
    the oracle as a whole (51 s per update on the CPU), so parity is checked through size-independent properties:
      * column independence: GAE returns / advantages and per-row log-probs of randomly chosen columns / rows equal the
        oracle run on just those columns / rows (bit-exact for the scan, 1e-5 for the fp32 MLP);
      * linearity: the unscaled folded gradients and loss sums of the full batch equal the sum over two column halves;
      * determinism: two full train() calls from identical state give bit-identical parameters and statistics."""
    import copy
    import bench
    out = {}
    T, N, A = bench.T, bench.N_PER_GPU, bench.A
    r = bench.build_gpu_runner(bench.WORKLOADS["mpe"], N, 0, 1, torch.device(DEV), "onpolicy")
    rng = np.random.default_rng(0)
    # ---- 1. GAE scan: 48 random columns vs the oracle on those columns (bit-exact)
    cb = r.critic_buffer
    snap_vp = cb.value_preds.clone()
    nv = cb.value_preds[-1].clone()
    cb.compute_returns(nv, r.value_normalizer)
    torch.cuda.synchronize()
    cols = np.sort(rng.choice(N, 48, replace=False))
    g = lambda t: t[:, cols].cpu().numpy()  # noqa: E731
    ovn = O.OracleValueNorm()
    st = r.value_normalizer.stats.cpu().numpy()
    ovn.load_state(dict(running_mean=float(st[0]), running_mean_sq=float(st[1]), debiasing_term=float(st[2])))
    ret, _ = O.compute_returns(g(cb.rewards), g(snap_vp), g(cb.masks), g(cb.bad_masks), nv[cols].cpu().numpy(), 0.99, 0.95, True,
                               True, ovn)
    out["gae_columns_mismatch"] = float(np.sum(ret[:T] != g(cb.returns)[:T]))
    adv = O.advantages_from_returns(ret, g(cb.value_preds), ovn)
    out["adv_columns_mismatch"] = float(np.sum(adv.astype(np.float32) != g(cb.advantages)))
    # ---- 2. per-row log-probs of 4096 random rows vs the oracle on those rows
    a0, b0 = r.actor[0], r.actor_buffer[0]
    rows = np.sort(rng.choice(T * N, 4096, replace=False))
    lp, _, _ = a0.evaluate_actions(b0.flat("obs"), None, b0.flat("actions"), None)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in a0.actor.state_dict().items()}
    args = {**r.algo_args["model"], **r.algo_args["algo"]}
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    with torch.no_grad():
        ref, _, _ = O.actor_evaluate_actions(sd, cfg, b0.flat("obs")[rows].cpu(), b0.flat("actions")[rows].cpu(), None, None)
    out["logp_rows_vec_rel"] = vec_rel_err(lp[rows].cpu().numpy(), ref.numpy())
    # ---- 3. linearity of the unscaled sums under a column split (what the data-parallel path relies on)
    B = T * N
    advf = cb.advantages.reshape(B).contiguous()
    mom = torch.zeros(3, dtype=torch.float64, device=DEV)
    a0.masked_moments(b0, advf, mom)
    factor = torch.ones(B, device=DEV)
    a0.actor.fold()
    full_idx = None
    def fb(idx, m):
        nb = a0._forward_backward(b0.flat("obs"), idx, m, b0.flat("actions"), None, b0.flat("action_log_probs"), advf, mom,
                                  factor, b0.flat("active_masks").reshape(B))
        sc = torch.zeros(48, dtype=torch.float64, device=DEV)
        from harl_amd._lib import call, ptr, stream
        call("harl_reduce_scalars", ptr(a0.actor.part_scalars), nb, ptr(sc), stream())
        return a0.actor.dwp.clone(), sc
    gfull, sfull = fb(full_idx, B)
    allrows = torch.arange(B, device=DEV)
    left = allrows[(allrows % N) < N // 2].contiguous()
    right = allrows[(allrows % N) >= N // 2].contiguous()
    gl, sl = fb(left, left.numel())
    gr, sr = fb(right, right.numel())
    torch.cuda.synchronize()
    out["grad_split_sum_vec_rel"] = vec_rel_err((gl + gr).cpu().numpy(), gfull.cpu().numpy())
    out["scalars_split_sum_rel"] = rel_err((sl + sr)[:5].cpu().numpy(), sfull[:5].cpu().numpy())
    # ---- 4. determinism of a whole train() at full size
    def snapshot():
        return ([a.actor.flat_param.clone() for a in r.actor], r.critic.critic.flat_param.clone(),
                [(a.actor_optimizer.exp_avg.clone(), a.actor_optimizer.exp_avg_sq.clone(), a.actor_optimizer.step_count) for a in r.actor],
                (r.critic.critic_optimizer.exp_avg.clone(), r.critic.critic_optimizer.exp_avg_sq.clone(), r.critic.critic_optimizer.step_count),
                r.value_normalizer.stats.clone(), torch.get_rng_state())
    def restore(s):
        for a, p, o in zip(r.actor, s[0], s[2]):
            a.actor.flat_param.copy_(p); a.actor_optimizer.exp_avg.copy_(o[0]); a.actor_optimizer.exp_avg_sq.copy_(o[1])
            a.actor_optimizer.step_count = o[2]
        r.critic.critic.flat_param.copy_(s[1])
        r.critic.critic_optimizer.exp_avg.copy_(s[3][0]); r.critic.critic_optimizer.exp_avg_sq.copy_(s[3][1])
        r.critic.critic_optimizer.step_count = s[3][2]
        r.value_normalizer.stats.copy_(s[4]); torch.set_rng_state(s[5])
    s0 = snapshot()
    i1, c1 = r.train()
    p1 = [a.actor.flat_param.clone() for a in r.actor] + [r.critic.critic.flat_param.clone()]
    restore(s0)
    i2, c2 = r.train()
    p2 = [a.actor.flat_param.clone() for a in r.actor] + [r.critic.critic.flat_param.clone()]
    torch.cuda.synchronize()
    out["determinism_param_mismatch"] = float(sum(int((x != y).sum().item()) for x, y in zip(p1, p2)))
    out["determinism_info_mismatch"] = float(i1 != i2 or c1 != c2)
    out["train_nonfinite_count"] = float(not all(np.isfinite(list(i.values())).all() for i in i1))
    return out


TRPO_TRACE_KEYS = ("accepted", "fraction", "kl", "loss", "loss_improve", "expected_improve", "dist_entropy", "ratio", "step_size", "shs")


def _forcing_hook(taps_of: dict):
    """TEACHER FORCING (round 6), as an ``oracle.GRAD_HOOK``: before every optimiser step of the objects in ``taps_of`` {id(oracle
    actor / critic): [(flat parameters, exp_avg, exp_avg_sq, step count), ...]} their parameters and Adam moments are overwritten
    with the HIP path's state in front of the same step -- every update is compared from identical inputs, nothing a previous
    update did differently is carried along (the free-running comparison is chaotic on the bench buffers: the same HIP step
    against three evidence runs of the same oracle gave pooled excess ratios between 0.15 and 2.7,
    profiles/r06_free_running_spread.md)."""
    seen = {}

    def force(stage, obj, sample, _vn):
        if stage != "pre" or id(obj) not in taps_of:
            return
        n_ = seen.get(id(obj), 0)
        seen[id(obj)] = n_ + 1
        flat, m_, v_, step = taps_of[id(obj)][n_]
        off = 0
        with torch.no_grad():
            for prm in obj.net.params():
                n = prm.numel()
                prm.copy_(torch.from_numpy(flat[off:off + n]).view(prm.shape).to(prm.dtype))
                if step > 0:
                    stt = obj.net.opt.state[prm]
                    if "exp_avg" not in stt:
                        stt["exp_avg"], stt["exp_avg_sq"] = torch.zeros_like(prm), torch.zeros_like(prm)
                    stt["step"] = torch.tensor(float(step))
                    stt["exp_avg"].copy_(torch.from_numpy(m_[off:off + n]).view(prm.shape).to(prm.dtype))
                    stt["exp_avg_sq"].copy_(torch.from_numpy(v_[off:off + n]).view(prm.shape).to(prm.dtype))
                off += n
    return force


def _oracle_bench_run(payload: dict, tag: str, dt_name: str, pert_seed, keep_grad: bool) -> dict:
    """One oracle compute() + ha_train() on the host copies in ``payload`` (see _bench_config_runs).  Runs either in this
    process or -- on hosts with enough cores -- in a worker process of its own (tests/oracle_worker.py), so that the fp32 /
    float64 / one-ulp runs of a full-size comparison take the wall time of the slowest instead of their sum."""
    import time as _time
    import bench
    w = bench.WORKLOADS[payload["workload"]]
    T, n_threads = w["T"], payload["n_threads"]
    args = bench.algo_args(n_threads, T, w)
    cfg = O.PathConfig.from_reference_dicts(args["train"], args["model"], args["algo"])
    torch.set_num_threads(int(os.environ.get("HARL_ORACLE_THREADS") or min(16, os.cpu_count() or 1)))
    dt = dict(f32=torch.float32, f64=torch.float64)[dt_name]
    if payload.get("mode") in ("agent", "critic"):  # teacher-forced single-agent / critic-only runs of the HATRPO workloads
        return _oracle_trpo_piece(payload, w, args, cfg, dt, pert_seed)
    actor_sd, critic_sd, abuf_np, cbuf_np, st0 = (payload[k] for k in ("actor_sd", "critic_sd", "abuf", "cbuf", "st0"))
    O.set_work_dtype(dt)
    try:
        t0 = _time.perf_counter()
        torch.set_rng_state(payload["rng0"])
        trpo = w["algo"] == "hatrpo"
        if trpo:
            tc = O.TrpoConfig(**{k: args["algo"][k] for k in ("kl_threshold", "ls_step", "accept_ratio", "backtrack_coeff")})
            actors = [O.OracleHATRPO({k: v.clone() for k, v in sd.items()}, cfg, tc) for sd in actor_sd]
        else:
            actors = [O.OracleHAPPO({k: v.clone() for k, v in sd.items()}, cfg) for sd in actor_sd]
        critic = O.OracleVCritic({k: v.clone() for k, v in critic_sd.items()}, cfg)
        if pert_seed is not None:
            _perturb_one_ulp([a_ if trpo else a_.net for a_ in actors] + [critic.net], pert_seed)
        abufs = [O.OracleActorBuffer(d["obs"].copy(), d["actions"].copy(), d["logp"].copy(), d["masks"].copy(), d["active"].copy(),
                                     None if d.get("avail") is None else d["avail"].copy(),
                                     rnn_states=None if d.get("rnn") is None else d["rnn"].copy()) for d in abuf_np]
        cbuf = O.OracleCriticBufferEP(cbuf_np["share_obs"].copy(), cbuf_np["rewards"].copy(), cbuf_np["value_preds"].copy(),
                                      cbuf_np["masks"].copy(), cbuf_np["bad_masks"].copy())
        if cbuf_np.get("rnn") is not None:
            cbuf.rnn_states_critic = cbuf_np["rnn"].copy()
        vn = O.OracleValueNorm()
        vn.load_state(dict(running_mean=float(st0[0]), running_mean_sq=float(st0[1]), debiasing_term=float(st0[2])))
        if payload.get("forced") is not None and not trpo:
            who = {id(a_): payload["forced"]["actor"][k] for k, a_ in enumerate(actors)}
            who[id(critic)] = payload["forced"]["critic"]
            O.GRAD_HOOK = _forcing_hook(who)
        with torch.no_grad():  # compute(): the critic's value of slot T (on_policy_base_runner.py:462-484)
            if cbuf_np.get("rnn") is not None:
                nv = critic.get_values(cbuf_np["share_obs"][-1], cbuf_np["rnn"][-1], cbuf_np["masks"][-1])
            else:
                nv = critic.get_values(cbuf_np["share_obs"][-1])
            nv = nv.detach().double().numpy().reshape(-1, 1)
        # identical scan inputs on both sides: the fp32 oracle's compute_returns is fed the HIP value of slot T (which is
        # compared with the oracle's own first)
        cbuf.compute_returns(payload["next_value_hip"].copy() if tag == "f32" else nv.astype(np.float64 if dt == torch.float64 else np.float32),
                             vn, cfg)
        infos, cinfo, extra_ = O.ha_train(actors, critic, abufs, cbuf, vn, cfg, keep_grad=keep_grad)
        if trpo:  # one update per agent: the line search's decisions and the five statistics (no parameter-sized arrays)
            atr = [[{k: (bool(v) if k == "accepted" else float(v)) for k, v in u.items() if k in TRPO_TRACE_KEYS} for u in a_.trace]
                   for a_ in actors]
            fin = [a_.flat().numpy().astype(np.float64) for a_ in actors]
        else:
            atr = [np.array([[u["policy_loss"], u["dist_entropy"], u["grad_norm"], u["ratio"]] for u in a_.trace]) for a_ in actors]
            fin = [np.asarray(a_.net.flat(), dtype=np.float64) for a_ in actors]
        return dict(nv=nv, adv=np.asarray(extra_["advantages"]), returns=np.asarray(cbuf.returns).copy(), infos=infos, cinfo=cinfo,
                    atr=atr,
                    ctr=np.array([[u["value_loss"], u["grad_norm"]] for u in critic.trace]),
                    fin=fin,
                    grads=[[np.asarray(u["grad"], dtype=np.float64) for u in a_.trace] for a_ in actors] if keep_grad else None,
                    cfin=np.asarray(critic.net.flat(), dtype=np.float64), vn=vn.state(), rng=torch.get_rng_state(),
                    seconds=_time.perf_counter() - t0)
    finally:
        O.GRAD_HOOK = None
        O.set_work_dtype(torch.float32)


def _oracle_trpo_piece(payload: dict, w: dict, args: dict, cfg, dt, pert_seed) -> dict:
    """One piece of a HATRPO workload's full-size comparison (check_bench_config_parity_trpo), so that the pieces run side by
    side on the host's cores instead of one 17-agent chain (> 20 minutes of double backward at 204 800 rows x 393 inputs):
      mode "agent"  : ONE agent's sequential-update step of on_policy_ha_runner.py:47-124 from the HIP path's inputs to that step
                      (the factor the HIP path handed this agent, its pre-update parameters, the shared returns): pre-update
                      log-probs, HATRPO.train (gradient, 10 CG steps, line search), post-update log-probs, the factor it hands
                      on.  Teacher forcing: every link of the 17-step chain is checked from identical inputs, errors of earlier
                      links do not pile up in later ones.
      mode "critic" : compute() + compute_returns + VCritic.train, and the REPLAY of every draw train() takes from the global
                      CPU generator (per agent: the single-batch sampler's randperm and the old-actor construction of
                      hatrpo.py:127-130; then the critic's real samplers) -> the generator state the HIP path must end in."""
    import time as _time
    T, n_threads, A = w["T"], payload["n_threads"], w["A"]
    rnn = bool(w.get("rnn"))
    O.set_work_dtype(dt)
    try:
        t0 = _time.perf_counter()
        cb = payload["cbuf"]
        vn = O.OracleValueNorm()
        st0 = payload["st0"]
        vn.load_state(dict(running_mean=float(st0[0]), running_mean_sq=float(st0[1]), debiasing_term=float(st0[2])))
        so = cb.get("share_obs")
        cbuf = O.OracleCriticBufferEP(np.zeros((T + 1, n_threads, 1), dtype=np.float32) if so is None else so.copy(), cb["rewards"].copy(),
                                      cb["value_preds"].copy(), cb["masks"].copy(), cb["bad_masks"].copy())
        if cb.get("rnn") is not None:
            cbuf.rnn_states_critic = cb["rnn"].copy()
        if payload["mode"] == "critic":
            critic = O.OracleVCritic({k: v.clone() for k, v in payload["critic_sd"].items()}, cfg)
            if pert_seed is not None:
                _perturb_one_ulp([critic.net], pert_seed)
            with torch.no_grad():
                nv = (critic.get_values(so[-1], cb["rnn"][-1], cb["masks"][-1]) if rnn else critic.get_values(so[-1]))
                nv = nv.detach().double().numpy().reshape(-1, 1)
            cbuf.compute_returns(payload["next_value_hip"].copy(), vn, cfg)
            # ---- the generator: replay the actors' draws, then the critic's real ones
            torch.set_rng_state(payload["rng0"])
            B = T * n_threads
            shapes = {k: tuple(v) for k, v in payload["actor_shapes"]}
            for _ in range(A):
                torch.randperm(B // cfg.data_chunk_length if cfg.use_recurrent_policy else (n_threads if cfg.use_naive_recurrent_policy else B))
                O.consume_policy_init_rng(shapes)
            if payload.get("forced_critic") is not None and pert_seed is None:  # (a one-ulp twin runs free: it IS the perturbation)
                O.GRAD_HOOK = _forcing_hook({id(critic): payload["forced_critic"]})
            cinfo = critic.train(cbuf, vn)
            return dict(nv=nv, returns=np.asarray(cbuf.returns).copy(), cinfo=cinfo,
                        ctr=np.array([[u["value_loss"], u["grad_norm"]] for u in critic.trace]),
                        cfin=np.asarray(critic.net.flat(), dtype=np.float64), vn=vn.state(), rng=torch.get_rng_state(),
                        seconds=_time.perf_counter() - t0)
        # ---- one agent
        a = payload["agent"]
        tc = O.TrpoConfig(**{k: args["algo"][k] for k in ("kl_threshold", "ls_step", "accept_ratio", "backtrack_coeff")})
        actor = O.OracleHATRPO({k: v.clone() for k, v in payload["actor_sd"].items()}, cfg, tc)
        if pert_seed is not None:
            _perturb_one_ulp([actor], pert_seed)
        d = payload["abuf"]
        buf = O.OracleActorBuffer(d["obs"].copy(), d["actions"].copy(), d["logp"].copy(), d["masks"].copy(), d["active"].copy(),
                                  None if d.get("avail") is None else d["avail"].copy(),
                                  rnn_states=None if d.get("rnn") is None else d["rnn"].copy())
        cbuf.compute_returns(payload["next_value_hip"].copy(), vn, cfg)
        advantages = O.advantages_from_returns(cbuf.returns, cbuf.value_preds, vn)
        np_work = np.float64 if dt == torch.float64 else np.float32
        factor_in = payload["factor_in"].astype(np_work)
        buf.update_factor(factor_in)
        flat = lambda v: v.reshape(T * n_threads, -1)  # noqa: E731
        avail = None if buf.available_actions is None else flat(buf.available_actions[:-1])
        ev = (flat(buf.obs[:-1]), flat(buf.actions), avail, flat(buf.active_masks[:-1]))
        if cfg.recurrent:
            ev = ev + (buf.rnn_states[0], flat(buf.masks[:-1]))
        torch.manual_seed(4242 + a)  # (the sampler's row order only changes the order of the sums)
        old_logp, _, _ = actor.evaluate_actions(*ev)
        info = actor.train(buf, advantages.copy())
        new_logp, _, _ = actor.evaluate_actions(*ev)
        agg = getattr(torch, cfg.action_aggregation)
        factor_out = factor_in * agg(torch.exp(new_logp - old_logp), dim=-1).reshape(T, n_threads, 1).detach().numpy()
        u = actor.trace[-1] if actor.trace else {}
        return dict(agent=a, info=info, trace={k: (bool(v) if k == "accepted" else float(v)) for k, v in u.items() if k in TRPO_TRACE_KEYS},
                    fin=actor.flat().numpy().astype(np.float64), factor_out=np.asarray(factor_out, dtype=np.float64),
                    seconds=_time.perf_counter() - t0)
    finally:
        O.GRAD_HOOK = None
        O.set_work_dtype(torch.float32)


def _oracle_launch(payload: dict, plan, keep_grad: bool, threads: int = 0) -> dict:
    """One worker process per entry of ``plan`` [(tag, dtype name, one-ulp seed | None)] (tests/oracle_worker.py; the payload
    travels through one file in a temporary directory); returns the handle ``_oracle_collect`` waits on.  With the core-slot
    scheduler active (tests/oracle_sched.py, prefetch_full_size) the workers are queued on it -- 16 threads on 16 logical CPUs of
    their own each -- otherwise they start right away."""
    import subprocess
    import sys
    import tempfile
    from tests import oracle_sched
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    td = tempfile.TemporaryDirectory(prefix="harl_oracle_")
    pin = os.path.join(td.name, "payload.pt")
    torch.save(payload, pin)
    nthr = str(threads or int(os.environ.get("HARL_ORACLE_THREADS") or 16))
    procs = []
    for tag, dtn, seed in plan:
        pout = os.path.join(td.name, f"{tag}.pt")
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS=nthr,
                   HARL_ORACLE_THREADS=nthr, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_WAIT_POLICY="PASSIVE")
        cmd = [sys.executable, "-m", "tests.oracle_worker", pin, pout, tag, dtn, "none" if seed is None else str(seed), str(int(keep_grad))]
        if oracle_sched.ACTIVE is not None:
            procs.append((tag, pout, oracle_sched.ACTIVE.submit(cmd, env, root)))
        else:
            procs.append((tag, pout, subprocess.Popen(cmd, cwd=root, env=env)))
    return dict(td=td, procs=procs)


def _oracle_collect(handle: dict) -> dict:
    runs = {}
    limit = float(os.environ.get("HARL_ORACLE_TIMEOUT_S", "1500"))
    try:
        for tag, pout, pr in handle["procs"]:
            rc = pr.wait(timeout=limit)
            if rc != 0:
                raise RuntimeError(f"oracle worker {tag} failed with exit code {rc}")
            runs[tag] = torch.load(pout, weights_only=False)
    finally:
        for _, _, pr in handle["procs"]:
            p_ = getattr(pr, "proc", pr)
            if p_ is not None and p_.poll() is None:
                p_.kill()
        handle["td"].cleanup()
    return runs


def _oracle_bench_runs(payload: dict, plan, keep_grad: bool) -> dict:
    """All oracle runs of ``plan`` [(tag, dtype name, one-ulp seed | None)].  With >= 64 host CPUs (the GPU boxes have 256) each
    run gets a process of its own (16 torch threads each); otherwise they run one after the other in this process.  Same code
    either way (_oracle_bench_run)."""
    mode = os.environ.get("HARL_ORACLE_PARALLEL", "auto")  # "0": in-process, "force": worker processes whatever the host
    if mode != "force" and ((os.cpu_count() or 1) < 64 or len(plan) == 1 or mode == "0"):
        return {tag: _oracle_bench_run(payload, tag, dtn, seed, keep_grad) for tag, dtn, seed in plan}
    return _oracle_collect(_oracle_launch(payload, plan, keep_grad))


def _bench_hip_step(n_threads: int, keep_grad: bool = False, logp: str = "recipe", workload: str = "mpe", agents=None):
    """The HIP half of a full-size comparison: build the bench runner of ``workload``, copy everything train() reads to the
    host, run ONE bench step with the per-update traces switched on.  Returns (hip dict, oracle payload, parameter names /
    shapes, meta)."""
    import bench
    w = bench.WORKLOADS[workload]
    T, A = w["T"], w["A"]
    trpo = w["algo"] == "hatrpo"
    torch.manual_seed(1)
    r = bench.build_gpu_runner(w, n_threads, 0, 1, DEV, logp)
    # ---- host copies of everything train() reads, taken BEFORE the HIP step
    npy = lambda t: t.detach().cpu().numpy().copy()  # noqa: E731
    actor_sd = [{k: v.detach().cpu().clone() for k, v in a.actor.state_dict().items()} for a in r.actor]
    critic_sd = {k: v.detach().cpu().clone() for k, v in r.critic.critic.state_dict().items()}
    # (``agents``: the HATRPO checks re-run a few agents only -- 17 x 323 MB of observations need not cross PCIe)
    abuf_np = [None if (agents is not None and k not in agents) else
               dict(obs=npy(b.obs), actions=npy(b.actions), logp=npy(b.action_log_probs), masks=npy(b.masks),
                    active=npy(b.active_masks), avail=None if b.available_actions is None else npy(b.available_actions),
                    rnn=npy(b.rnn_states) if w.get("rnn") else None) for k, b in enumerate(r.actor_buffer)]
    cb = r.critic_buffer
    cbuf_np = dict(share_obs=npy(cb.share_obs), rewards=npy(cb.rewards), value_preds=npy(cb.value_preds), masks=npy(cb.masks),
                   bad_masks=npy(cb.bad_masks), rnn=npy(cb.rnn_states_critic) if w.get("rnn") else None)
    st0 = npy(r.value_normalizer.stats)
    rng0 = torch.get_rng_state()
    # ---- HIP path: one bench step (bench.one_step) with the per-update traces switched on
    gtaps = [[] for _ in r.actor]
    forced = not keep_grad and os.environ.get("HARL_FULLSIZE_FORCED", "1") != "0"
    if forced:  # pre-update state of every optimiser step: the oracle re-runs each update from exactly this state
        if not trpo:  # (HATRPO actors take ONE step per train(): their pieces start from the initial parameters anyway)
            for a_ in r.actor:
                assert not a_.actor.md
                a_._state_tap = []
        r.critic._state_tap = []
    for a_, tp in zip(r.actor, gtaps):
        a_._trace = []
        if keep_grad:  # (host sync per update: diagnostics only)
            a_._grad_tap = (lambda tp_: (lambda gr, sc: tp_.append(gr.cpu().numpy().astype(np.float64))))(tp)
    r.critic._trace = []
    for a_ in r.actor:
        a_.actor.invalidate_caches()
    r.critic.critic.invalidate_caches()
    r.compute()
    torch.cuda.synchronize()
    next_value_hip = npy(cb.value_preds[-1])
    returns_hip = npy(cb.returns)
    ginfos, gcinfo = r.train()
    torch.cuda.synchronize()
    rng_hip = torch.get_rng_state()
    gtr = []
    for a_ in r.actor:
        if trpo:  # one dict per update (harl_amd/hatrpo.py)
            gtr.append([dict(u) for u in a_._trace])
            continue
        cum = torch.stack(a_._trace).double().cpu().numpy()
        gtr.append(np.diff(np.concatenate([np.zeros((1, cum.shape[1])), cum]), axis=0)[:, :4])
    cum = torch.stack(r.critic._trace).double().cpu().numpy()
    gctr = np.diff(np.concatenate([np.zeros((1, cum.shape[1])), cum]), axis=0)[:, :2]
    forced_actor = [[(p.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy(), int(k)) for p, m, v, k in a_._state_tap]
                    for a_ in r.actor] if (forced and not trpo) else None
    forced_critic = [(p.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy(), int(k)) for p, m, v, k in r.critic._state_tap] if forced else None
    gfin = [npy(a_.actor.flat_reference()) for a_ in r.actor]
    # the factor every agent was handed (on_policy_ha_runner.py:56: actor_buffer[agent].update_factor(factor)) stays in its buffer
    gfactor = [npy(b.factor) for b in r.actor_buffer] if trpo else None
    gcfin = npy(r.critic.critic.flat_param)
    gvn = npy(r.value_normalizer.stats)
    shapes = dict(actor=[(k, tuple(v.shape)) for k, v in actor_sd[0].items()], critic=[(k, tuple(v.shape)) for k, v in critic_sd.items()])
    del r
    torch.cuda.empty_cache()
    args = bench.algo_args(n_threads, T, w)
    cfg = O.PathConfig.from_reference_dicts(args["train"], args["model"], args["algo"])
    payload = dict(workload=workload, n_threads=n_threads, actor_sd=actor_sd, critic_sd=critic_sd, abuf=abuf_np, cbuf=cbuf_np,
                   st0=st0, rng0=rng0, next_value_hip=next_value_hip)
    if forced:
        payload["forced"] = dict(actor=forced_actor, critic=forced_critic)
    hip = dict(next_value=next_value_hip, returns=returns_hip, rng=rng_hip, atr=gtr, ctr=gctr, infos=ginfos, cinfo=gcinfo,
               fin=gfin, cfin=gcfin, vn=gvn, grads=gtaps if keep_grad else None, factor=gfactor)
    return hip, payload, shapes, dict(T=T, A=A, actor_sd=actor_sd, abuf=abuf_np, cfg=cfg)


def _oracle_plan(with_f64: bool, n_pert: int):
    return [("f32", "f32", None)] + ([("f64", "f64", None)] if with_f64 else []) + \
           [(f"pert{k}", "f32", 977 + k) for k in range(n_pert)]


# ---- full-size comparisons started ahead of the tests that assert on them -----------------------------------------------------
# The oracle's side of a full-size check is minutes of host-CPU work (f32 + float64 + one-ulp twins of compute() + train() at
# 819 200 rows per agent) during which the GPU idles; run one after the other inside their tests they were 383 s of the round-5
# suite's 859 s.  `prefetch_full_size` (called by a session fixture, tests/conftest.py) runs the HIP step of every selected
# full-size check right away -- seconds each -- and starts all their oracle workers at once; the tests collect the results
# at the end of the file, by which time the ~190 other tests have run next to the workers.
FULL_SIZE = {  # key -> (workload, logp, n_threads, spec)   [test name -> key: tests/conftest.py]; longest jobs first
    # HATRPO workloads: teacher-forced pieces (_oracle_trpo_piece) -- spec = the agents checked at the measured size (each in fp32
    # and from parameters one ulp away: 5-6 minutes of 8 host cores per run) + the critic; the other agents' links of the chain
    # are the HIP path's own
    "humanoid17": ("humanoid17", "recipe", 1024, (0, 8, 16)),
    "hatrpo_gru128": ("hatrpo_gru128", "recipe", 512, (0, 7)),
    # HAPPO workloads: spec = number of one-ulp twins next to the fp32 and float64 runs (0: the runs are teacher-forced, every
    # update starts from the HIP path's state -- a twin from perturbed INITIAL parameters would be overwritten at once)
    "cheetah6": ("cheetah6", "recipe", 4096, 0),
    "mpe_onpolicy": ("mpe", "onpolicy", 4096, 0),
    "mpe": ("mpe", "recipe", 4096, 0),
    "smac3s5z": ("smac3s5z", "recipe", 512, 0),
}
# bar of the full-size checks: max(1e-5, FULL_SIZE_NOISE x the fp32 oracle's own distance from its twin(s)).  The goldens' factor
# is 2 against noise files of many perturbation runs; here the yardstick is ONE or two twin runs, i.e. a max over few samples of a
# heavy-tailed quantity (a ReLU decision on one heavy sample moves a gradient row by 1e-3 of its norm), and the HIP path's
# distance is one more draw from the same distribution: with 2 a correct implementation failed one figure in three evidence
# runs of this round (gpurun_out/r06/gpu_tests_call{2,3}.txt), hence 4 -- next to flat ceilings on the raw figures in the tests.
FULL_SIZE_NOISE = 4.0
MAIN_PROCESS_CORES = 64  # logical CPUs (= 32 cores) the test process keeps for itself while oracle workers run
_PREFETCH: Dict[tuple, dict] = {}


# twins of the teacher-forced HATRPO pieces: an agent's step from parameters one ulp away (the conjugate-gradient solve amplifies
# input perturbations by the condition number of F: a float64 run from the SAME parameters says nothing about that); the critic --
# cheap -- in float64 and from one-ulp parameters
TRPO_AGENT_PLAN = [("f32", "f32", None), ("pert0", "f32", 977)]
TRPO_CRITIC_PLAN = [("f32", "f32", None), ("f64", "f64", None), ("pert0", "f32", 978)]


def _trpo_launch(hip: dict, payload: dict, shapes: dict, agents) -> dict:
    """Queue the pieces of a HATRPO workload's comparison: per checked agent one payload (its buffers and parameters, the factor
    the HIP path handed it, the small critic-buffer arrays) run in fp32 and float64; one critic payload (fp32 + float64)."""
    small = {k: payload["cbuf"][k] for k in ("rewards", "value_preds", "masks", "bad_masks")}
    small["rnn"] = payload["cbuf"].get("rnn")
    common = dict(workload=payload["workload"], n_threads=payload["n_threads"], st0=payload["st0"], rng0=payload["rng0"],
                  next_value_hip=payload["next_value_hip"])
    handles = {}
    for a in agents:
        pl = dict(common, mode="agent", agent=a, actor_sd=payload["actor_sd"][a], abuf=payload["abuf"][a], cbuf=small,
                  factor_in=hip["factor"][a])
        handles[a] = _oracle_launch(pl, TRPO_AGENT_PLAN, False)
    pl = dict(common, mode="critic", critic_sd=payload["critic_sd"], cbuf=payload["cbuf"], actor_shapes=shapes["actor"],
              forced_critic=(payload.get("forced") or {}).get("critic"))
    handles["critic"] = _oracle_launch(pl, TRPO_CRITIC_PLAN, False)
    return handles


def prefetch_full_size(keys) -> None:
    """HIP step + oracle worker launch for every key of FULL_SIZE in ``keys``.  Failures are kept and raised by the test that asks
    for the result, not here.  Hosts with < 64 CPUs (no room for worker processes) do nothing: the checks then run inline.
    The host's logical CPUs are PARTITIONED: the test process pins itself to the first MAIN_PROCESS_CORES of its affinity set
    (and runs that many torch threads); the rest are slots of 16 for the oracle workers (tests/oracle_sched.py: 16 torch
    threads each -- the thread count the measured bars were taken at; the oracle's fp32 figures depend on it -- queued in
    submission order).  The first attempt let twenty 16-thread workers and the test process's 128 OpenMP threads share all cores:
    the suite ran ~20x slower, every OpenMP barrier waiting for descheduled threads."""
    if (os.cpu_count() or 1) < 64 and os.environ.get("HARL_ORACLE_PARALLEL") != "force":
        return
    todo = [k for k in FULL_SIZE if k in set(keys)]
    if not todo:
        return
    from tests import oracle_sched
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    main, slots = oracle_sched.partition(avail, min(MAIN_PROCESS_CORES, max(2, len(avail) // 4)))
    try:
        os.sched_setaffinity(0, main)
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(max(1, len(main) // 2))  # (one thread per physical core of the test process's share)
    oracle_sched.ACTIVE = oracle_sched.Scheduler([], slots=slots)
    for k in todo:
        workload, logp, n_threads, spec = FULL_SIZE[k]
        slot = (workload, logp, n_threads, spec)
        try:
            hip, payload, shapes, meta = _bench_hip_step(n_threads, False, logp, workload,
                                                         agents=spec if isinstance(spec, tuple) else None)
            if isinstance(spec, tuple):  # HATRPO: teacher-forced pieces
                handle = _trpo_launch(hip, payload, shapes, spec)
            else:
                handle = _oracle_launch(payload, _oracle_plan(True, spec), False)
            del payload
            _PREFETCH[slot] = dict(hip=hip, shapes=shapes, meta=meta, handle=handle)
        except Exception as e:  # noqa: BLE001 -- re-raised by the test of this check
            _PREFETCH[slot] = dict(error=e)
        torch.cuda.empty_cache()


def _trpo_config_runs(workload: str, n_threads: int, agents):
    """(hip dict, {agent: {"f32": .., "f64": ..}}, {"f32": .., "f64": ..} critic runs, meta) of a HATRPO workload: collected from
    the session's prefetch, or computed here (worker processes on big hosts, in-process otherwise)."""
    slot = (workload, "recipe", n_threads, tuple(agents))
    pre = _PREFETCH.pop(slot, None)
    if pre is not None:
        if "error" in pre:
            raise pre["error"]
        hip, handles, meta = pre["hip"], pre["handle"], pre["meta"]
    else:
        hip, payload, shapes, meta = _bench_hip_step(n_threads, False, "recipe", workload, agents=tuple(agents))
        mode = os.environ.get("HARL_ORACLE_PARALLEL", "auto")
        if mode != "force" and ((os.cpu_count() or 1) < 64 or mode == "0"):
            handles = None
            small = {k: payload["cbuf"][k] for k in ("rewards", "value_preds", "masks", "bad_masks")}
            small["rnn"] = payload["cbuf"].get("rnn")
            common = dict(workload=workload, n_threads=n_threads, st0=payload["st0"], rng0=payload["rng0"],
                          next_value_hip=payload["next_value_hip"])
            runs = {}
            for a in agents:
                pl = dict(common, mode="agent", agent=a, actor_sd=payload["actor_sd"][a], abuf=payload["abuf"][a], cbuf=small,
                          factor_in=hip["factor"][a])
                runs[a] = {t: _oracle_bench_run(pl, t, dtn, sd_, False) for t, dtn, sd_ in TRPO_AGENT_PLAN}
            pl = dict(common, mode="critic", critic_sd=payload["critic_sd"], cbuf=payload["cbuf"], actor_shapes=shapes["actor"],
                      forced_critic=(payload.get("forced") or {}).get("critic"))
            crit = {t: _oracle_bench_run(pl, t, dtn, sd_, False) for t, dtn, sd_ in TRPO_CRITIC_PLAN}
            return hip, runs, crit, meta
        handles = _trpo_launch(hip, payload, shapes, agents)
        del payload
    runs = {a: _oracle_collect(handles[a]) for a in agents}
    crit = _oracle_collect(handles["critic"])
    return hip, runs, crit, meta


def _bench_config_runs(n_threads: int, with_f64: bool, keep_grad: bool = False, logp: str = "recipe", n_pert: int = 0,
                       workload: str = "mpe"):
    """HIP step + oracle run(s) of a BENCH configuration on identical contents (see check_bench_config_parity).  Returns
    (hip dict, {"f32": .., "f64": .., "pert0": ..} oracle dicts, names/shapes of the actor / critic parameter tensors, meta).
    ``n_pert`` further fp32 oracle runs start from parameters moved by one ulp (``_perturb_one_ulp``): how far the reference's
    own fp32 figures move under the smallest change fp32 can express.  ``workload``: an entry of bench.WORKLOADS.  A result
    started by ``prefetch_full_size`` is collected instead of recomputed."""
    slot = (workload, logp, n_threads, n_pert)
    pre = _PREFETCH.pop(slot, None) if not keep_grad else None
    if pre is not None:
        if "error" in pre:
            raise pre["error"]
        return pre["hip"], _oracle_collect(pre["handle"]), pre["shapes"], pre["meta"]
    hip, payload, shapes, meta = _bench_hip_step(n_threads, keep_grad, logp, workload)
    runs = _oracle_bench_runs(payload, _oracle_plan(with_f64, n_pert), keep_grad)
    return hip, runs, shapes, meta


def dump_parity(name: str, out: dict) -> None:
    """Keep a parity check's result dict as JSON under gpurun_out/parity/ (HARL_PARITY_DIR overrides): `pytest -q` hides the
    printed figures, and gpurun merges that directory back from the GPU box; the round's copies live in profiles/."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.environ.get("HARL_PARITY_DIR") or os.path.join(root, "gpurun_out", "parity")
    try:
        os.makedirs(d, exist_ok=True)
        sha = None
        if os.path.exists(os.path.join(root, ".git_sha")):
            sha = open(os.path.join(root, ".git_sha")).read().strip()
        rec = dict(check=name, git_sha=sha, device=torch.cuda.get_device_name(0) if torch.cuda.is_available() else None,
                   figures={k: v for k, v in out.items() if isinstance(v, (int, float, str))})
        with open(os.path.join(d, name + ".json"), "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)
    except OSError:
        pass


def check_bench_config_parity(logp: str = "recipe", n_threads: int = 4096, n_pert: int = 0, workload: str = "mpe") -> Dict[str, float]:
    """The BENCH configuration itself against the oracle: exactly what ``bench.py`` times -- BASELINE.json
    configs[1], T = 200, n_rollout_threads = 4096, 3 agents, obs 18 / share_obs 54 / Box 5, MLP [128, 128], 5 + 5 epochs, fixed
    agent order, recipe log-probs -- built by ``bench.build_gpu_runner``; buffers, weights and ValueNorm state are copied to the
    host and the fp32 oracle's ``compute() + ha_train`` (on_policy_ha_runner.py:11-130; ~27 s on 16 host threads) runs on the
    same contents.  Compared: the critic's value of slot T, returns (bit-exact: the oracle's scan is fed the HIP value of slot
    T so that both scans see identical inputs), the generator state, EVERY optimiser step's policy_loss / dist_entropy /
    grad_norm / ratio (3 x 5) and value_loss / critic_grad_norm (5), the averaged infos, the final parameter vectors and the
    ValueNorm statistics.

    Round 6: the comparison is TEACHER-FORCED (``_forcing_hook``).  The synthetic buffers carry no signal (advantages independent
    of the actions), so a gradient is a sum of 819 200 random-sign terms, and with the recipe's stored log-probs (-1 + 0.1 N(0,1),
    unrelated to the policy) the importance ratios span 1e-22 .. 3e+2: ONE ReLU decision that two fp32 implementations take
    differently on a heavy sample moves a row of a weight gradient by 1e-3 of its norm, and Adam's m / sqrt(v) carries that into
    every later update -- two free-running fp32 trajectories separate from the second update on, the oracle against its own
    one-ulp twin as much as the HIP path against the oracle, and a bar built from a few such twins is a lottery
    (profiles/r06_free_running_spread.md: the SAME HIP step scored pooled excess ratios between 0.15 and 2.7 against three runs
    of the same oracle).  So the HIP path records its parameters and Adam moments in front of EVERY optimiser step and the oracle
    -- in fp32 and in float64 -- re-runs each step from exactly that state: every update is a first update.  Asserted here:
      * per kind of figure (all 15 grad-norms, ...) max error / max(1e-5, FULL_SIZE_NOISE x the fp32 oracle's own distance from
        its float64 twin on the same steps) <= 1 (``*_excess``);
    and by the tests on top: 1e-5 FLAT on every update's figures except the cancelling sums of the on-policy variant.
    ``HARL_FULLSIZE_FORCED=0`` runs the free form (``n_pert`` one-ulp twins make sense only there)."""
    out: Dict[str, float] = {}
    hip, runs, _shapes, meta = _bench_config_runs(n_threads, True, logp=logp, n_pert=n_pert, workload=workload)
    T, A = meta["T"], meta["A"]
    import bench as _bench
    recurrent = bool(_bench.WORKLOADS[workload].get("rnn"))  # GRU chains: no flat 1e-5 on anything downstream of them
    o, o64 = runs["f32"], runs["f64"]
    perts = [runs[k] for k in sorted(runs) if k.startswith("pert")]
    out["_oracle_seconds"] = float(sum(v["seconds"] for v in runs.values()))
    out["next_value_vec_rel"] = vec_rel_err(hip["next_value"], o["nv"])
    out["returns_mismatch"] = float(np.sum(hip["returns"][:T] != o["returns"][:T].astype(np.float32)))
    out["rng_state_mismatch"] = float(not torch.equal(hip["rng"], o["rng"]))

    def rel(a_, b_):
        a_, b_ = np.asarray(a_, dtype=np.float64), np.asarray(b_, dtype=np.float64)
        return np.abs(a_ - b_) / (np.abs(b_) + 1e-12)

    def pooled(get, key):
        """max error of the HIP figures / max(1e-5, FULL_SIZE_NOISE x the oracle's pooled own uncertainty on this kind of figure)"""
        err = float(rel(get(hip), get(o)).max())
        floor = float(rel(get(o), get(o64)).max())
        for pr in perts:
            floor = max(floor, float(rel(get(pr), get(o)).max()))
        out[f"_{key}_rel"] = err
        out[f"_{key}_oracle_own_uncertainty"] = floor
        out[f"{key}_excess"] = err / max(1e-5, FULL_SIZE_NOISE * floor)

    names = ("policy_loss", "dist_entropy", "grad_norm", "ratio")
    for c, nm in enumerate(names):
        get = lambda run, c=c: np.stack([t[:, c] for t in run["atr"]])  # noqa: E731  [agent, epoch]
        first = float(rel(get(hip)[0, 0], get(o)[0, 0]))
        if logp == "recipe" and not recurrent:
            out[f"first_update_{nm}_rel"] = first  # flat 1e-5
        else:
            # on-policy buffers: the policy loss is a masked mean of advantage-normalised surrogates with ratios ~ 1, i.e. ~ 0 by
            # construction -- its RELATIVE error has no flat bar even on the first update (measured: 1.1e-5 here, the fp32 oracle
            # 1e-5 from float64); raw figure reported, the measured bar asserted
            ffloor = max([float(rel(get(o)[0, 0], get(o64)[0, 0]))] + [float(rel(get(pr)[0, 0], get(o)[0, 0])) for pr in perts])
            out[f"_first_update_{nm}_rel"] = first
            out[f"_first_update_{nm}_oracle_own_uncertainty"] = ffloor
            if nm == "policy_loss" and not recurrent:
                # |loss| ~ 2e-3 here (mean of 819 200 unit-scale terms that cancel): the figure with a meaning is the ABSOLUTE
                # difference -- in units of the advantage-normalised terms' scale (~1) -- held to 1e-7 by the test
                out["_first_update_policy_loss_value"] = float(get(o)[0, 0])
                out["_first_update_policy_loss_abs"] = float(abs(get(hip)[0, 0] - get(o)[0, 0]))
                out["_first_update_policy_loss_excess"] = first / max(1e-5, FULL_SIZE_NOISE * ffloor)
            else:
                out[f"first_update_{nm}_excess"] = first / max(1e-5, FULL_SIZE_NOISE * ffloor)
        pooled(get, f"actor_update_{nm}")
    for c, nm in enumerate(("value_loss", "grad_norm")):
        if recurrent:  # the critic's GRU chunks: per-update figures on the pooled measured bar, like the actors'
            pooled(lambda run, c=c: run["ctr"][:, c], f"critic_update_{nm}")
        else:
            out[f"critic_update_{nm}_rel"] = float(rel(hip["ctr"][:, c], o["ctr"][:, c]).max())
    keys = ("policy_loss", "dist_entropy", "actor_grad_norm", "ratio")
    pooled(lambda run: np.array([[float(i[k]) for k in keys] for i in run["infos"]], dtype=np.float64), "actor_infos")
    out["critic_info_rel"] = rel_err([hip["cinfo"]["value_loss"], hip["cinfo"]["critic_grad_norm"]],
                                     [o["cinfo"]["value_loss"], o["cinfo"]["critic_grad_norm"]])
    ovn = o["vn"]
    out["vn_final_rel"] = rel_err(hip["vn"], [float(np.asarray(ovn[k]).reshape(-1)[0]) for k in ("running_mean", "running_mean_sq", "debiasing_term")])
    worst_raw, floor = 0.0, 0.0
    for a in range(A):
        raw = vec_rel_err(hip["fin"][a], o["fin"][a])
        worst_raw = max(worst_raw, raw)
        out[f"_actor{a}_final_param_vec_rel"] = raw
        out[f"_actor{a}_oracle_f32_vs_f64_vec_rel"] = vec_rel_err(o["fin"][a], o64["fin"][a])
        floor = max(floor, out[f"_actor{a}_oracle_f32_vs_f64_vec_rel"])
        for k, pr in enumerate(perts):
            out[f"_actor{a}_oracle_one_ulp_run{k}_vec_rel"] = vec_rel_err(pr["fin"][a], o["fin"][a])
            floor = max(floor, out[f"_actor{a}_oracle_one_ulp_run{k}_vec_rel"])
    out["_actor_final_param_vec_rel_max"] = worst_raw
    out["_actor_final_param_oracle_own_uncertainty"] = floor
    out["actor_final_param_excess"] = worst_raw / max(1e-5, FULL_SIZE_NOISE * floor)
    out["_critic_final_param_vec_rel"] = vec_rel_err(hip["cfin"], o["cfin"])
    out["_critic_oracle_f32_vs_f64_vec_rel"] = vec_rel_err(o["cfin"], o64["cfin"])
    cfl = max([out["_critic_oracle_f32_vs_f64_vec_rel"]] + [vec_rel_err(pr["cfin"], o["cfin"]) for pr in perts])
    out["critic_final_param_excess"] = out["_critic_final_param_vec_rel"] / max(1e-5, FULL_SIZE_NOISE * cfl)
    out["_teacher_forced"] = float(os.environ.get("HARL_FULLSIZE_FORCED", "1") != "0")
    out["_oracle_wall_seconds_max"] = float(max(v["seconds"] for v in runs.values()))
    dump_parity(f"bench_config_parity_{workload}_{logp}", out)
    return out


def check_bench_config_parity_trpo(workload: str = "humanoid17", n_threads: int = 1024, agents=(0, 1, 8, 16)) -> Dict[str, float]:
    """A HATRPO bench workload at its MEASURED size against the oracle: `humanoid17` (17 agents x
    204 800 rows, obs 393, MLP [128]x3, hatrpo.yaml defaults) or `hatrpo_gru128` (8 agents, 128-wide GRU, Discrete(14) with
    unavailable actions, chunks of 10).  The HIP path runs the whole compute() + train(); the oracle (hatrpo.py:37-194,
    trpo_util.py:96-158: double backward, 10 CG steps, backtracking line search) re-runs the sequential-update step of the
    agents in ``agents`` -- each from the inputs the HIP path gave that step: the factor it was handed, its pre-update
    parameters, the shared returns -- in fp32 and in float64, and the critic, as separate worker processes (the 17-agent chain
    in one process is > 20 minutes of host time; the pieces take ~1-2 minutes side by side).  Teacher forcing: every checked link
    of the factor chain starts from identical inputs, so nothing upstream piles up in it.  Asserted:
      * returns bit-exact; the CPU generator's final state (the oracle side replays every draw of train()); the value of slot T;
      * per checked agent the SAME accept / reject decision and the SAME number of backtracks as the fp32 oracle -- integers,
        no tolerance -- unless the oracle's own float64 twin decides differently for that agent (reported, excluded);
      * kl, loss (surrogate at theta_old), loss_improve, expected_improve, dist_entropy, ratio, step_size, the agent's final
        parameters and the FACTOR IT HANDS ON (819 200-entry array against the HIP path's input of the next agent): pooled over
        the checked agents on the measured bar max(1e-5, 2 x the fp32 oracle's own distance from float64), flat ceilings on the
        raw figures in the test;
      * the critic: 1e-5 flat (feed-forward) / measured bar (recurrent)."""
    import bench as _bench
    out: Dict[str, float] = {}
    hip, runs, crit, meta = _trpo_config_runs(workload, n_threads, agents)
    T, A = meta["T"], meta["A"]
    recurrent = bool(_bench.WORKLOADS[workload].get("rnn"))
    o = crit["f32"]
    ctw = [crit[k] for k in sorted(crit) if k != "f32"]  # the critic's twins (float64, one-ulp)
    twin_of = lambda a: [runs[a][k] for k in sorted(runs[a]) if k != "f32"]  # noqa: E731  (an agent's twins)
    secs = [v["seconds"] for r_ in runs.values() for v in r_.values()] + [v["seconds"] for v in crit.values()]
    out["_oracle_seconds"] = float(sum(secs))
    out["_oracle_wall_seconds_max"] = float(max(secs))
    out["_oracle_run_seconds"] = " ".join(f"a{a}:" + "/".join(f"{r_[k]['seconds']:.0f}" for k in sorted(r_)) for a, r_ in runs.items()) + \
        " critic:" + "/".join(f"{crit[k]['seconds']:.0f}" for k in sorted(crit))
    out["next_value_vec_rel"] = vec_rel_err(hip["next_value"], o["nv"])
    out["returns_mismatch"] = float(np.sum(hip["returns"][:T] != o["returns"][:T].astype(np.float32)))
    out["rng_state_mismatch"] = float(not torch.equal(hip["rng"], o["rng"]))

    def rel(a_, b_):
        a_, b_ = np.asarray(a_, dtype=np.float64), np.asarray(b_, dtype=np.float64)
        return np.abs(a_ - b_) / (np.abs(b_) + 1e-12)

    def bt(u):  # number of backtracks from the oracle's `fraction` = backtrack_coeff ** k
        if "backtracks" in u:
            return int(u["backtracks"])
        return int(round(math.log(max(u["fraction"], 1e-300)) / math.log(0.8)))

    dec = lambda u: ("A" if u["accepted"] else "R") + str(bt(u))  # noqa: E731
    g_ = {a: hip["atr"][a][0] for a in agents}
    o_ = {a: runs[a]["f32"]["trace"] for a in agents}
    t_ = {a: twin_of(a)[0]["trace"] for a in agents}
    out["_agents_checked"] = " ".join(str(a) for a in agents)
    out["_decisions_hip_all_agents"] = " ".join(dec(hip["atr"][a][0]) for a in range(A))
    out["_decisions_hip"] = " ".join(dec(g_[a]) for a in agents)
    out["_decisions_oracle"] = " ".join(dec(o_[a]) for a in agents)
    out["_decisions_oracle_twin"] = " ".join(dec(t_[a]) for a in agents)
    stable = [a for a in agents if all(dec(o_[a]) == dec(tw["trace"]) for tw in twin_of(a))]
    out["_agents_with_oracle_own_disagreement"] = float(len(agents) - len(stable))
    out["linesearch_decision_mismatch"] = float(sum(g_[a]["accepted"] != o_[a]["accepted"] for a in stable))
    out["linesearch_backtracks_mismatch"] = float(sum(bt(g_[a]) != bt(o_[a]) for a in stable))
    cmp_agents = [a for a in stable if dec(g_[a]) == dec(o_[a])]  # figures are comparable where both sides walked the same path
    out["_agents_compared"] = float(len(cmp_agents))
    for nm in ("kl", "loss", "loss_improve", "expected_improve", "dist_entropy", "ratio", "step_size"):
        if not cmp_agents:
            break
        get = lambda us, nm=nm: np.array([us[a][nm] for a in cmp_agents], dtype=np.float64)  # noqa: E731
        err = float(rel(get(g_), get(o_)).max())
        floor = max(float(rel(np.array([tw["trace"][nm] for tw in twin_of(a)]), o_[a][nm]).max()) for a in cmp_agents)
        out[f"_trpo_{nm}_rel"] = err
        out[f"_trpo_{nm}_oracle_own_uncertainty"] = floor
        out[f"trpo_{nm}_excess"] = err / max(1e-5, FULL_SIZE_NOISE * floor)
    # final parameters of the checked agents; the factor each one hands on = the HIP path's input of the next agent
    worst, floor, fworst, ffloor, links = 0.0, 0.0, 0.0, 0.0, 0
    for a in cmp_agents:
        e_a = vec_rel_err(hip["fin"][a], runs[a]["f32"]["fin"])
        f_a = max(vec_rel_err(tw["fin"], runs[a]["f32"]["fin"]) for tw in twin_of(a))
        # ... and of the STEP (theta_new - theta_old: the conjugate-gradient solve scaled by the line search), which the
        # parameter vector's norm hides
        th0 = np.concatenate([v.numpy().reshape(-1) for v in meta["actor_sd"][a].values()]).astype(np.float64)
        st_o = runs[a]["f32"]["fin"] - th0
        out[f"_agent{a}_final_param_vec_rel"] = e_a
        out[f"_agent{a}_final_param_oracle_own_uncertainty"] = f_a
        out[f"_agent{a}_step_norm_over_param_norm"] = float(np.linalg.norm(st_o) / np.linalg.norm(th0))
        out[f"_agent{a}_step_vec_rel"] = vec_rel_err(np.asarray(hip["fin"][a], dtype=np.float64) - th0, st_o)
        out[f"_agent{a}_step_oracle_own_uncertainty"] = max(vec_rel_err(tw["fin"] - th0, st_o) for tw in twin_of(a))
        worst = max(worst, e_a)
        floor = max(floor, f_a)
        if a + 1 < A:
            links += 1
            want = runs[a]["f32"]["factor_out"].reshape(-1)
            fworst = max(fworst, vec_rel_err(hip["factor"][a + 1].reshape(-1), want))
            ffloor = max([ffloor] + [vec_rel_err(tw["factor_out"].reshape(-1), want) for tw in twin_of(a)])
            out[f"_factor_after_agent{a}_max_abs_rel"] = float(np.max(np.abs(hip["factor"][a + 1].reshape(-1) - want) / (np.abs(want) + 1e-30)))
    out["_actor_final_param_vec_rel_max"] = worst
    out["_actor_final_param_oracle_own_uncertainty"] = floor
    out["actor_final_param_excess"] = worst / max(1e-5, FULL_SIZE_NOISE * floor)
    out["_factor_links_checked"] = float(links)
    out["_factor_vec_rel_max"] = fworst
    out["_factor_oracle_own_uncertainty"] = ffloor
    out["factor_excess"] = fworst / max(1e-5, FULL_SIZE_NOISE * ffloor)
    for c, nm in enumerate(("value_loss", "grad_norm")):
        err = float(rel(hip["ctr"][:, c], o["ctr"][:, c]).max())
        if recurrent:
            fl = max(float(rel(tw["ctr"][:, c], o["ctr"][:, c]).max()) for tw in ctw)
            out[f"_critic_update_{nm}_rel"] = err
            out[f"critic_update_{nm}_excess"] = err / max(1e-5, FULL_SIZE_NOISE * fl)
        else:
            out[f"critic_update_{nm}_rel"] = err
    ovn = o["vn"]
    out["vn_final_rel"] = rel_err(hip["vn"], [float(np.asarray(ovn[k]).reshape(-1)[0]) for k in ("running_mean", "running_mean_sq", "debiasing_term")])
    out["_critic_final_param_vec_rel"] = vec_rel_err(hip["cfin"], o["cfin"])
    cfloor = max(vec_rel_err(tw["cfin"], o["cfin"]) for tw in ctw)
    out["_critic_final_param_oracle_own_uncertainty"] = cfloor
    out["critic_final_param_excess"] = out["_critic_final_param_vec_rel"] / max(1e-5, FULL_SIZE_NOISE * cfloor)
    dump_parity(f"bench_config_parity_{workload}_full_size", out)
    return out


def check_generator_api(name: str) -> Dict[str, float]:
    """The buffers' public generators (feed-forward / naive / chunked recurrent, actor and critic) yield, for the
    reference's recorded permutations, exactly the rows the oracle's generators yield (bit-exact gathers)."""
    case = GoldenCase(name)
    out = {}
    r = build_runner(case)
    train, model, algo = case.reference_dicts()
    T, N = case.shapes.T, case.shapes.N
    d = case.data
    ob = O.OracleActorBuffer(d.obs[0].copy(), d.actions[0].copy(), d.action_log_probs[0].copy(), d.masks[0].copy(),
                             d.active_masks[0].copy(),
                             None if d.available_actions[0] is None else d.available_actions[0].copy(),
                             rnn_states=None if d.rnn is None else d.rnn["actor"][0].copy())
    adv = np.random.default_rng(1).standard_normal((T, N, 1)).astype(np.float32)
    fac = (1 + 0.1 * np.random.default_rng(2).standard_normal((T, N, 1))).astype(np.float32)
    ob.update_factor(fac)
    buf = r.actor_buffer[0]
    buf.update_factor(fac)
    k, L = 2, model["data_chunk_length"]
    def run(gen_o, gen_p, order):
        bad = 0
        torch.manual_seed(77)
        want = [s for s, _ in gen_o()]
        torch.manual_seed(77)
        got = list(gen_p())
        bad += int(len(want) != len(got))
        for w, g in zip(want, got):
            for wi, gi in order:
                a, b = w[wi], g[gi]
                if a is None or b is None:
                    bad += int((a is None) != (b is None))
                else:
                    bad += int(not np.array_equal(np.asarray(a).reshape(-1), b.cpu().numpy().reshape(-1)))
        return float(bad)
    # oracle tuple: (obs, actions, active, logp, adv, avail, factor[, rnn, masks]) ; product: reference order
    ff = [(0, 0), (1, 2), (2, 4), (3, 5), (4, 6), (5, 7), (6, 8)]
    out["ff_actor_mismatch"] = run(lambda: ob.feed_forward_generator(adv, k), lambda: buf.feed_forward_generator_actor(adv, k), ff)
    if case.recurrent:
        rec = ff + [(7, 1), (8, 3)]
        out["chunk_actor_mismatch"] = run(lambda: ob.recurrent_generator(adv, k, L),
                                          lambda: buf.recurrent_generator_actor(adv, k, L), rec)
        out["naive_actor_mismatch"] = run(lambda: ob.naive_recurrent_generator(adv, k),
                                          lambda: buf.naive_recurrent_generator_actor(adv, k), rec)
    # critic
    cb = r.critic_buffer
    cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
    torch.cuda.synchronize()
    if case.state_type == "FP":
        f = d.fp
        oc = O.OracleCriticBufferFP(f["share_obs"].copy(), f["rewards"].copy(), cb.value_preds.cpu().numpy(), f["masks"].copy(),
                                    f["bad_masks"].copy())
    else:
        oc = O.OracleCriticBufferEP(d.share_obs.copy(), d.rewards.copy(), cb.value_preds.cpu().numpy(), d.critic_masks.copy(),
                                    d.bad_masks.copy())
    oc.returns = cb.returns.cpu().numpy()
    if d.rnn is not None:
        oc.rnn_states_critic = d.rnn["critic_fp" if case.state_type == "FP" else "critic"].copy()
    cff = [(0, 0), (1, 2), (2, 3)]  # oracle (share_obs, value_preds, returns[, rnn, masks]); product reference order
    out["ff_critic_mismatch"] = run(lambda: oc.feed_forward_generator(k), lambda: cb.feed_forward_generator_critic(k), cff)
    if case.recurrent:
        crec = cff + [(3, 1), (4, 4)]
        out["chunk_critic_mismatch"] = run(lambda: oc.recurrent_generator(k, L), lambda: cb.recurrent_generator_critic(k, L), crec)
        out["naive_critic_mismatch"] = run(lambda: oc.naive_recurrent_generator(k), lambda: cb.naive_recurrent_generator_critic(k), crec)
    return out


# the three ways the wide GEMMs get their weight fragments: streamed from L2 (the default), first eight k-steps resident in LDS
# (round 6, opt-in: measured neutral), 32-column panels shared through LDS (round 5, opt-in: measured slower)
WIDE_MODES = (("stream", {"HARL_WIDE_SHARED": "0", "HARL_WIDE_RESIDENT": "0"}),
              ("resident", {"HARL_WIDE_SHARED": "0", "HARL_WIDE_RESIDENT": "1"}),
              ("shared", {"HARL_WIDE_SHARED": "1", "HARL_WIDE_RESIDENT": "0"}))


def check_wide_shared(M: int = 70000) -> Dict[str, float]:
    """The wide GEMMs with their weight fragments shared through LDS (k_fwd_wide_sh, round 5) against the streaming kernel they
    replace from 512 slabs on (k_fwd_wide): same MFMA sequence per slab, so the outputs must agree BIT FOR BIT -- forward of a
    393-wide first layer, its tangent, the raw mode, and the one-launch hidden tangent (K = 2 H), at a size where every workgroup
    walks several 8-slab iterations and the last one is ragged."""
    out: Dict[str, float] = {}
    ns = (M + 31) // 32
    mp = ns * 32
    g = torch.Generator(device=DEV).manual_seed(3)
    rn = lambda *sh: torch.randn(*sh, device=DEV, generator=g)  # noqa: E731
    H = 128
    for D in (393, 100):
        KP = (D + 31) // 32 * 32
        x0n = rn(mp * KP)
        W, b = rn(H * D) * 0.1, rn(H) * 0.1
        wimg = torch.empty(3 * H * KP // 2, device=DEV)
        xh1 = rn(mp * H)
        mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (ns * 2 * 64,), device=DEV, dtype=torch.int32, generator=g)
        rstd = torch.rand(mp, device=DEV, generator=g) + 0.5
        res = {}
        for mode, env in WIDE_MODES:
            os.environ.update(env)
            xo, mo, ro = torch.zeros(mp * H, device=DEV), torch.zeros(ns * 2 * 64, dtype=torch.int32, device=DEV), torch.zeros(mp, device=DEV)
            call("harl_mlp_fwd_wide", ptr(x0n), M, KP, ptr(W), D, ptr(b), H, ptr(wimg), ptr(xo), ptr(mo), ptr(ro), stream())
            xd = torch.zeros(mp * H, device=DEV)
            call("harl_mlp_tangent_wide", ptr(x0n), M, KP, ptr(W), D, ptr(b), H, ptr(wimg), ptr(xh1), ptr(mask), ptr(rstd), ptr(xd), stream())
            zr = torch.zeros(mp * H, device=DEV)
            call("harl_mlp_linear_wide", ptr(x0n), M, KP, ptr(W), D, ptr(b), H, ptr(wimg), ptr(zr), stream())
            torch.cuda.synchronize()
            res[mode] = (xo, mo, ro, xd, zr)
        for k, nm in enumerate(("fwd_x", "fwd_mask", "fwd_rstd", "tangent", "raw")):  # (whole images: the padding rows see the same inputs)
            out[f"D{D}_{nm}_mismatch"] = float((res["stream"][k] != res["shared"][k]).sum().item())
            out[f"D{D}_{nm}_resident_mismatch"] = float((res["stream"][k] != res["resident"][k]).sum().item())
        out[f"D{D}_fwd_all_zero_count"] = float((res["shared"][0] != 0).sum().item() == 0)
    # one-launch hidden tangent
    xin, xdot = rn(mp * H), rn(mp * H)
    Wp, Wd, bd = rn(H * H) * 0.1, rn(H * H) * 0.1, rn(H) * 0.1
    wimg = torch.empty(3 * H * 2 * H // 2, device=DEV)
    xh1 = rn(mp * H)
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (ns * 2 * 64,), device=DEV, dtype=torch.int32, generator=g)
    rstd = torch.rand(mp, device=DEV, generator=g) + 0.5
    res = {}
    for mode, env in WIDE_MODES:
        os.environ.update(env)
        o = torch.zeros(mp * H, device=DEV)
        call("harl_mlp_tangent_hidden2", ptr(xdot), ptr(xin), M, H, H, ptr(Wp), ptr(Wd), ptr(bd), ptr(wimg), ptr(xh1), ptr(mask), ptr(rstd),
             ptr(o), stream())
        torch.cuda.synchronize()
        res[mode] = o
    os.environ.pop("HARL_WIDE_SHARED", None)
    os.environ.pop("HARL_WIDE_RESIDENT", None)
    out["tangent_hidden2_mismatch"] = float((res["stream"] != res["shared"]).sum().item())
    out["tangent_hidden2_resident_mismatch"] = float((res["stream"] != res["resident"]).sum().item())
    out["tangent_hidden2_nonzero_count"] = float((res["shared"] != 0).sum().item() == 0)
    return out


def _atl_rows(t: torch.Tensor, ns: int, H: int) -> torch.Tensor:
    """ATL(H) image [ns slabs] -> rows [ns * 32, H] (common.h: piece q, lane half h, sample s, element c <-> feature
    32 (q >> 2) + 8 (q & 3) + 4 h + c)."""
    img = t.detach().cpu().double().reshape(ns, H // 8, 2, 32, 4)
    q = torch.arange(H // 8)
    feat = (32 * (q // 4) + 8 * (q % 4)).reshape(-1, 1, 1) + 4 * torch.arange(2).reshape(1, -1, 1) + torch.arange(4).reshape(1, 1, -1)
    rows = torch.empty(ns, 32, H, dtype=torch.float64)
    rows[:, :, feat.reshape(-1)] = img.permute(0, 3, 1, 2, 4).reshape(ns, 32, -1)
    return rows.reshape(ns * 32, H)


def _relu_mask_rows(mask: torch.Tensor, ns: int, H: int) -> torch.Tensor:
    """ReLU bit masks (common.h: lane (s, h) keeps register R of its H/2 in word R >> 5, MSB first) -> bool rows [ns * 32, H]."""
    NW = (H // 2 + 31) // 32
    m = mask.detach().cpu().to(torch.int64).reshape(ns, NW, 64) & 0xFFFFFFFF
    out = torch.zeros(ns, 32, H, dtype=torch.bool)
    for R in range(H // 2):
        bit = (m[:, R >> 5, :] >> (31 - (R & 31))) & 1          # [ns, 64 lanes]
        f0 = 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2)
        out[:, :, f0] = bit[:, :32].bool()
        out[:, :, f0 + 4] = bit[:, 32:].bool()
    return out.reshape(ns * 32, H)


def check_bwd_fused(M: int, first: bool, fill: int = 1, seed: int = 0) -> Dict[str, float]:
    """harl_mlp_bwd_dx_dw (the whole backward of a 128 x 128 hidden layer in one launch, round 5) on random ATL operands:
    (i) against a float64 restatement of the math it replaces -- autograd through Linear + ReLU + LayerNorm,
    harl/models/base/mlp.py:25-38: dz_prev, dW', db' and (first-layer variant) dW_1' | db_1', each as the error relative to the
    largest entry; (ii) against the layer kernels it replaces (harl_mlp_dw_partials + harl_mlp_bwd_dx): dz_prev bit for bit
    (same instruction sequence per slab), the weight gradients to fp32 summation-order noise."""
    H, KP = 128, 32
    ns = (M + 31) // 32
    mp = ns * 32
    g = torch.Generator(device=DEV).manual_seed(seed)
    rn = lambda *sh: torch.randn(*sh, device=DEV, generator=g)  # noqa: E731
    dz, xh, x0n = rn(mp * H), rn(mp * H), rn(mp * KP)
    mask = torch.randint(-2 ** 31, 2 ** 31 - 1, (ns * 2 * 64,), device=DEV, dtype=torch.int32, generator=g)
    rstd = torch.rand(mp, device=DEV, generator=g) + 0.5
    W = rn(H * H) * 0.1
    n_wg = max(1, min(512, (ns + 1) // 2))
    ROW2, ROW1 = H * H + H, H * KP + H
    out: Dict[str, float] = {}
    # ---- new kernel
    dzp_new = torch.full((mp * H,), float("nan"), device=DEV)
    part2_new = torch.full((n_wg * ROW2,), float("nan"), device=DEV)
    part1_new = torch.full((n_wg * ROW1,), float("nan"), device=DEV)
    call("harl_mlp_bwd_dx_dw", ptr(dz), ptr(xh), ptr(mask), ptr(rstd), M, H, H, ptr(W), ptr(dzp_new), ptr(x0n) if first else None,
         KP if first else 0, ptr(part1_new) if first else None, ptr(part2_new), n_wg, fill, stream())
    # ---- the layer kernels it replaces
    dzp_old = torch.full((mp * H,), float("nan"), device=DEV)
    part2_old = torch.full((n_wg * ROW2,), float("nan"), device=DEV)
    part1_old = torch.full((n_wg * ROW1,), float("nan"), device=DEV)
    call("harl_mlp_dw_partials", ptr(dz), 0, 0, H, ptr(xh), 0, 0, None, None, None, H, M, ptr(part2_old), n_wg, stream())
    call("harl_mlp_bwd_dx", ptr(dz), ptr(xh), ptr(mask), ptr(rstd), M, H, H, ptr(W), ptr(dzp_old), ptr(x0n) if first else None,
         KP if first else 0, ptr(part1_old) if first else None, n_wg if first else 0, stream())
    torch.cuda.synchronize()
    out["dz_prev_vs_layer_kernels_mismatch"] = float((dzp_new != dzp_old).sum().item())
    s2n, s2o = part2_new.reshape(n_wg, ROW2).double().sum(0).cpu(), part2_old.reshape(n_wg, ROW2).double().sum(0).cpu()
    out["dw2_vs_layer_kernels_vec_rel"] = vec_rel_err(s2n.numpy(), s2o.numpy())
    # ---- float64 restatement
    dzr, xr = _atl_rows(dz, ns, H), _atl_rows(xh, ns, H)
    Wm = W.detach().cpu().double().reshape(H, H)
    dxh = dzr @ Wm
    rs = rstd.detach().cpu().double().reshape(-1, 1)
    da = rs * (dxh - dxh.mean(-1, keepdim=True) - xr * (dxh * xr).mean(-1, keepdim=True))
    dzp_ref = torch.where(_relu_mask_rows(mask, ns, H), da, torch.zeros_like(da))
    out["dz_prev_vec_rel"] = vec_rel_err(_atl_rows(dzp_new, ns, H).numpy(), dzp_ref.numpy())
    dW2 = dzr.t() @ xr
    out["dw2_vec_rel"] = vec_rel_err(s2n[:H * H].reshape(H, H).numpy(), dW2.numpy())
    out["db2_vec_rel"] = vec_rel_err(s2n[H * H:].numpy(), dzr.sum(0).numpy())
    if first:
        s1n, s1o = part1_new.reshape(n_wg, ROW1).double().sum(0).cpu(), part1_old.reshape(n_wg, ROW1).double().sum(0).cpu()
        out["dw1_vs_layer_kernels_vec_rel"] = vec_rel_err(s1n.numpy(), s1o.numpy())
        x0r = _atl_rows(x0n, ns, KP)
        out["dw1_vec_rel"] = vec_rel_err(s1n[:H * KP].reshape(H, KP).numpy(), (dzp_ref.t() @ x0r).numpy())
        out["db1_vec_rel"] = vec_rel_err(s1n[H * KP:].numpy(), dzp_ref.sum(0).numpy())
    # ---- determinism: a second launch gives the same bits
    part2_b = torch.empty_like(part2_new)
    dzp_b = torch.empty_like(dzp_new)
    part1_b = torch.empty_like(part1_new)
    call("harl_mlp_bwd_dx_dw", ptr(dz), ptr(xh), ptr(mask), ptr(rstd), M, H, H, ptr(W), ptr(dzp_b), ptr(x0n) if first else None,
         KP if first else 0, ptr(part1_b) if first else None, ptr(part2_b), n_wg, fill, stream())
    torch.cuda.synchronize()
    out["rerun_mismatch"] = float((part2_b != part2_new).sum().item() + (dzp_b != dzp_new).sum().item()
                                  + ((part1_b != part1_new).sum().item() if first else 0))
    return out


def check_fused_vs_layered(rows: int, mode: str = "1", hidden=(128, 128), obs_dim: int = 18) -> Dict[str, float]:
    """csrc/update.hip (HARL_FUSED_UPDATE=``mode``: 1 or hybrid) against the layer-by-layer kernels (=0) on the same data: unscaled folded
    gradients, loss sums, first-epoch log-probs, log-prob pass + factor product; second fused run bit-identical."""
    import os
    sh = Shapes(T=rows, N=1, A=1, obs_dim=obs_dim, share_obs_dim=54, act_dim=5, discrete=False, hidden_sizes=list(hidden))
    d = make_buffers(sh, 3)
    actor, _, _ = _mk_actor(sh, 1)
    critic, _, _ = _mk_critic(sh, 2)
    obs = dev(d.obs[0][:-1].reshape(rows, -1))
    act = dev(d.actions[0].reshape(rows, -1))
    rng = np.random.default_rng(0)
    adv = dev(rng.standard_normal(rows).astype(np.float32))
    factor = dev((1 + 0.1 * rng.standard_normal(rows)).astype(np.float32))
    active = dev((rng.random(rows) > 0.1).astype(np.float32))
    so = dev(d.share_obs[:-1].reshape(rows, -1))
    vp = dev(rng.standard_normal(rows).astype(np.float32))
    ret = dev((vp.cpu().numpy() + rng.standard_normal(rows)).astype(np.float32))
    prev = os.environ.get("HARL_FUSED_UPDATE")
    got = {}
    try:
        os.environ["HARL_FUSED_UPDATE"] = "0"
        actor.actor.fold()
        lp0 = torch.empty(rows, actor.actor.act_w, device=DEV)
        actor._logp_pass(obs, act, None, rows, lp0)
        old_logp = (lp0 + dev(0.1 * rng.standard_normal((rows, actor.actor.act_w)).astype(np.float32))).contiguous()
        # Two fp32 paths compute the importance ratio with different roundings; a sample whose ratio sits within rounding
        # distance of a clip edge gets its gradient switched on in one and off in the other, and ONE such sample of 131 299
        # moves these noise-sum gradients (random-sign advantages: cancellation ~ sqrt(M)) by 2.5e-3 of their inf-norm
        # (round 4, tools/diag_fused_last.py: slab 1748, one row, everything else equal to 5e-7).  Those rows are taken out
        # of BOTH runs (active = 0), as the BASELINE-shape tests do (_mask_relu_kinks).
        imp0 = torch.exp((lp0 - old_logp).sum(-1))
        # (2e-5: a log-prob of -24.46 carries 3 ulps = 5.7e-6 of difference between the two paths: 0.7999964 vs 0.8000012)
        near = ((imp0 / 0.8 - 1).abs() < 2e-5) | ((imp0 / 1.2 - 1).abs() < 2e-5)
        active = torch.where(near, torch.zeros_like(active), active).contiguous()
        n_edge = int(near.sum().item())
        for tag, mode_ in (("old", "0"), ("new", mode), ("again", mode)):
            os.environ["HARL_FUSED_UPDATE"] = mode_
            actor.actor.invalidate_caches()
            critic.critic.invalidate_caches()
            lp = torch.zeros(rows, actor.actor.act_w, device=DEV)
            nblk = actor._forward_backward(obs, None, rows, act, None, old_logp, adv, None, factor, active, logp_out=lp)
            sc = torch.zeros(_lib.PS_STRIDE, dtype=torch.float64, device=DEV)
            call("harl_reduce_scalars", ptr(actor.actor.part_scalars), nblk, ptr(sc), stream())
            lp2 = torch.empty(rows, actor.actor.act_w, device=DEV)
            fac = factor.clone()
            actor._logp_pass(obs, act, None, rows, lp2, old_logp=old_logp, factor=fac)
            from harl_amd.valuenorm import ValueNorm
            gvn = ValueNorm(1, device=DEV)
            gvn.stats.copy_(dev(np.array([0.15, 0.85, 0.5], dtype=np.float32)))
            taps = []
            critic._grad_tap = lambda gr, s_: taps.append((gr.clone(), s_.clone()))
            critic.critic.fold()
            critic._update_core(so, None, rows, rows, vp, ret, None)
            vals, _ = critic.get_values(so, None, None)
            got[tag] = dict(dwp=actor.actor.dwp.clone(), sc=sc, lp=lp, lp2=lp2, fac=fac, cg=taps[0][0], csc=taps[0][1],
                            vals=vals.clone())
            # undo the critic's optimiser step so that every mode starts from the same weights
            critic, _, _ = _mk_critic(sh, 2)
    finally:
        if prev is None:
            os.environ.pop("HARL_FUSED_UPDATE", None)
        else:
            os.environ["HARL_FUSED_UPDATE"] = prev
    torch.cuda.synchronize()

    def vrel(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    o, n, a = got["old"], got["new"], got["again"]
    if os.environ.get("HARL_DEBUG_BLOCKS"):  # per layer block of the folded-gradient arena (debugging aid)
        net = actor.actor
        offs = sorted(set([0, net.dwp.numel()] + [int(x) for x in net._dwp_offs]))
        for a_, b_ in zip(offs[:-1], offs[1:]):
            d_ = (n["dwp"][a_:b_] - o["dwp"][a_:b_]).abs()
            print(f"block [{a_},{b_}): max|old| {float(o['dwp'][a_:b_].abs().max()):.3e} max|new-old| {float(d_.max()):.3e} at {int(d_.argmax())}"
                  f" ; n > 1e-5 max: {int((d_ > 1e-5 * o['dwp'][a_:b_].abs().max()).sum())}")
        print("scalars old", o["sc"][:14].cpu().numpy(), "new", n["sc"][:14].cpu().numpy())
    out = dict(_clip_edge_rows_masked=float(n_edge), actor_dwp_vec_rel=vrel(n["dwp"], o["dwp"]), actor_logp_epoch0_vec_rel=vrel(n["lp"], o["lp"]),
               actor_logp_pass_vec_rel=vrel(n["lp2"], o["lp2"]), actor_factor_vec_rel=vrel(n["fac"], o["fac"]),
               actor_loss_sums_rel=float(((n["sc"][:5] - o["sc"][:5]).abs() / o["sc"][:5].abs().clamp_min(1e-30)).max()),
               critic_grad_vec_rel=vrel(n["cg"], o["cg"]), critic_values_vec_rel=vrel(n["vals"], o["vals"]),
               critic_loss_sums_rel=float(((n["csc"][:2] - o["csc"][:2]).abs() / o["csc"][:2].abs().clamp_min(1e-30)).max()),
               actor_rerun_bitwise_equal=float(torch.equal(n["dwp"], a["dwp"]) and torch.equal(n["sc"], a["sc"])),
               critic_rerun_bitwise_equal=float(torch.equal(n["cg"], a["cg"])))
    return out


def check_many_slabs_linearity(spec, rows: int, chunk: int = 8192) -> Dict[str, float]:
    """Size-independent property instead of a CPU oracle at a size the oracle cannot hold (a float64 autograd pass over
    5e5 rows exhausts the GPU box's host memory; torch-fp32 itself is only good to ~1e-3 of a gradient tensor's inf-norm
    there: sums with cancellation ~ sqrt(M)): the UNSCALED folded gradients and loss sums of one forward + loss + backward
    over `rows` rows -- where every wave of the persistent kernels walks eight or more slabs -- must equal the float64 sum of
    the same quantities over disjoint chunks of `chunk` rows, each of which is a size the oracle-checked tests cover (a
    slab or two per wave).  A slab-indexing mistake that only shows when a wave iterates breaks this identity."""
    nvec = spec.get("nvec")
    sh = Shapes(T=rows, N=1, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"], **({"nvec": list(nvec)} if nvec else {}))
    actor, _, _ = _mk_actor(sh, 31)
    net = actor.actor
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    obs = torch.randn(rows, sh.obs_dim, generator=g, device=DEV)
    if nvec:
        act = torch.stack([torch.randint(0, n, (rows,), generator=g, device=DEV) for n in nvec], -1).float()
    elif sh.discrete:
        act = torch.randint(0, sh.act_dim, (rows, 1), generator=g, device=DEV).float()
    else:
        act = torch.randn(rows, sh.act_dim, generator=g, device=DEV)
    adv = torch.randn(rows, generator=g, device=DEV)
    factor = (1 + 0.1 * torch.randn(rows, generator=g, device=DEV)).contiguous()
    net.fold()
    lp0 = torch.empty(rows, net.act_w, device=DEV)
    actor._logp_pass(obs, act, None, rows, lp0)
    old_logp = (lp0 + 0.1 * torch.randn(rows, net.act_w, generator=g, device=DEV)).contiguous()

    def run(lo, hi):
        m = hi - lo
        net.invalidate_caches()
        nblk = actor._forward_backward(obs[lo:hi].contiguous(), None, m, act[lo:hi].contiguous(), None,
                                       old_logp[lo:hi].contiguous(), adv[lo:hi].contiguous(), None, factor[lo:hi].contiguous(), None)
        sc = torch.zeros(_lib.PS_STRIDE, dtype=torch.float64, device=DEV)
        call("harl_reduce_scalars", ptr(net.part_scalars), nblk, ptr(sc), stream())
        return net.dwp.double().clone(), sc[:5].clone()

    full_dwp, full_sc = run(0, rows)
    acc_dwp, acc_sc = torch.zeros_like(full_dwp), torch.zeros_like(full_sc)
    for lo in range(0, rows, chunk):
        d_, s_ = run(lo, min(rows, lo + chunk))
        acc_dwp += d_
        acc_sc += s_
    torch.cuda.synchronize()
    out = {"loss_sums_rel": float(((full_sc - acc_sc).abs() / acc_sc.abs().clamp_min(1e-30)).max())}
    # per layer block of the folded-gradient arena (the blocks differ by orders of magnitude)
    worst = 0.0
    offs = sorted(set([0, full_dwp.numel()] + [int(o) for o in net._dwp_offs]))
    for a_, b_ in zip(offs[:-1], offs[1:]):
        ref = acc_dwp[a_:b_]
        if float(ref.abs().max()) > 0:
            worst = max(worst, float((full_dwp[a_:b_] - ref).abs().max() / ref.abs().max()))
    out["folded_grad_block_vec_rel"] = worst
    return out


def check_gradient_noise(spec, agg: str = "prod") -> Dict[str, float]:
    """Where does ONE update of the HIP path sit relative to exact arithmetic, compared with the reference's fp32 arithmetic?
    Pre-clip gradient of the same HAPPO.update computed three ways -- HIP kernels, oracle in float32 (= the reference's
    arithmetic), oracle in float64 -- and reported per parameter tensor as |g - g64|_inf / |g64|_inf for the first two
    (diagnostic: `_t32/<tensor>` and `_gpu/<tensor>`; `gpu_over_ref32_worst` = the largest ratio of the two)."""
    out = {}
    over = dict(spec.get("over", {}))
    over.update(action_aggregation=agg)
    M = spec["M"]
    sh = Shapes(T=M, N=1, A=1, obs_dim=spec["obs_dim"], share_obs_dim=spec["share_obs_dim"], act_dim=spec["act_dim"],
                discrete=spec["discrete"], hidden_sizes=spec["hidden_sizes"])
    d = make_buffers(sh, 33, unavailable_p=0.25 if sh.discrete else 0.0)
    actor, sd, args = _mk_actor(sh, 99, **over)
    cfg = O.PathConfig.from_reference_dicts({}, args, args)
    rng = np.random.default_rng(5)
    obs = d.obs[0][:-1].reshape(M, -1)
    avail = None if not sh.discrete else d.available_actions[0][:-1].reshape(M, -1)
    act = d.actions[0].reshape(M, -1)
    o32 = O.OracleHAPPO({k: torch.from_numpy(v) for k, v in sd.items()}, cfg)
    with torch.no_grad():
        lp, _, _ = o32.evaluate_actions(obs, act, avail, None)
    old_logp = (lp.numpy() + 0.15 * rng.standard_normal(lp.shape)).astype(np.float32)
    adv = rng.standard_normal((M, 1)).astype(np.float32)
    factor = (1 + 0.2 * rng.standard_normal((M, 1))).astype(np.float32)
    active = d.active_masks[0][:-1].reshape(M, 1)
    sample_o = (obs, act, active, old_logp, adv, avail, factor)
    pl32, _, gn32, _, g32 = o32.update(sample_o, keep_grad=True)
    O.set_work_dtype(torch.float64)
    try:
        o64 = O.OracleHAPPO({k: torch.from_numpy(v) for k, v in sd.items()}, cfg)
        pl64, _, gn64, _, g64 = o64.update(sample_o, keep_grad=True)
    finally:
        O.set_work_dtype(torch.float32)
    taps = []
    actor._grad_tap = lambda gr, sc: taps.append((gr.clone(), sc))
    rnn = np.zeros((M, 1, 1), dtype=np.float32)
    res = actor.update((obs, rnn, act, None, active, old_logp, adv, avail, factor))
    torch.cuda.synchronize()
    gg = taps[0][0].double().cpu().numpy()
    worst, off = 0.0, 0
    for name, shp in actor_param_shapes(sh, args["use_feature_normalization"]):
        n = int(np.prod(shp))
        ref = np.max(np.abs(g64[off:off + n])) + 1e-30
        e32 = float(np.max(np.abs(g32[off:off + n] - g64[off:off + n])) / ref)
        eg = float(np.max(np.abs(gg[off:off + n] - g64[off:off + n])) / ref)
        out[f"_t32/{name}"] = e32
        out[f"_gpu/{name}"] = eg
        rms = lambda x: float(np.sqrt(np.mean(x * x)))  # noqa: E731
        out[f"_t32rms/{name}"] = rms(g32[off:off + n] - g64[off:off + n]) / ref
        out[f"_gpurms/{name}"] = rms(gg[off:off + n] - g64[off:off + n]) / ref
        worst = max(worst, eg / max(e32, 1e-9))
        off += n
    out["gpu_over_ref32_worst"] = worst
    out["_loss_t32_vs_64"] = rel_err(pl32.item(), pl64.item())
    out["_loss_gpu_vs_64"] = rel_err(res[0].item(), pl64.item())
    out["_gradnorm_t32_vs_64"] = rel_err(float(gn32), float(gn64))
    out["_gradnorm_gpu_vs_64"] = rel_err(res[2].item(), float(gn64))
    return out


# BASELINE.json configurations at their real network / agent shapes with the thread count cut so that the oracle (one CPU
# socket) finishes each in well under a minute: rows = T * N <= 32 000, FULL agent count, one optimiser epoch per network.
BASELINE_SHAPES = {
    # configs[1]: MPE simple_spread, 3 agents, obs 18 / share 54, Box(5), hidden [128, 128]
    "mpe3": dict(shapes=dict(T=200, N=160, A=3, obs_dim=18, share_obs_dim=54, act_dim=5, discrete=False,
                             hidden_sizes=[128, 128]), seed=3, overrides=dict(ppo_epoch=1, critic_epoch=1)),
    # configs[2]: MAMuJoCo HalfCheetah-6x1, 6 agents, obs 23 / share 17, Box(1), hidden [128, 128, 128]
    "cheetah6": dict(shapes=dict(T=200, N=160, A=6, obs_dim=23, share_obs_dim=17, act_dim=1, discrete=False,
                                 hidden_sizes=[128, 128, 128]), seed=4, overrides=dict(ppo_epoch=1, critic_epoch=1)),
    # configs[3]: SMAC 3s5z, 8 agents, obs 128 / state 216, Discrete(14) with unavailable actions, GRU policy, T = 160,
    # chunks of 10 (tuned_configs/smac/3s5z/happo)
    "smac3s5z": dict(shapes=dict(T=160, N=96, A=8, obs_dim=128, share_obs_dim=216, act_dim=14, discrete=True,
                                 hidden_sizes=[64, 64, 64]), seed=5, unavailable_p=0.3,
                     overrides=dict(ppo_epoch=1, critic_epoch=1, use_recurrent_policy=True, data_chunk_length=10)),
    # configs[4]: MAMuJoCo Humanoid-17x1, 17 agents, obs 393 / share 376, Box(1), HATRPO (CG + FVP + line search)
    "humanoid17": dict(shapes=dict(T=200, N=40, A=17, obs_dim=393, share_obs_dim=376, act_dim=1, discrete=False,
                                   hidden_sizes=[128, 128, 128]), seed=6, algo="hatrpo",
                       overrides=dict(critic_epoch=1, fixed_order=True)),
}


def _mask_relu_kinks(case, margin: float = 2e-5, replace=None) -> int:
    """The update has two kinds of kinks: ReLU (derivative 0 / 1 at z = 0) and the PPO clip (gradient on / off where the
    importance ratio crosses 1 +- clip_param).  A sample sitting within rounding distance of one gets the derivative switched
    on in one fp32 implementation and off in another.  Measured at T = 200, N = 160: ONE sample out of 32 000 with a
    pre-activation of ~1e-8 moved the lower layers' weight gradients by up to 1.3 % of their inf-norm, another one whose
    ratio was 4e-7 from 0.8 moved EVERY gradient tensor by 0.5-3 %, and Adam's first step amplified either into 1e-3
    differences in the following agents' losses -- while the same figures agree to 1e-6 once those samples are left out.
    With ~8 million pre-activations per update a few always are that close, so they are taken out of the comparison: every
    (t, n) whose actor pre-activations (float64 forward of the initial weights) come within ``margin`` of zero in any MLP
    layer, or whose ratio comes within ``margin`` (relative) of a clip boundary, gets active_mask = 0, which removes it from
    the loss and the gradient in the reference and here alike (happo.py:77-81).  Returns the number of masked samples.
    HATRPO (``replace`` defaults to True there): the Fisher-vector product differentiates kl.mean() over EVERY row of the batch,
    active or not (trpo_util.py:132-158), so an inactive kink-adjacent row still flips a ReLU derivative inside F -- one such
    row moves F v by 1/M of its norm (measured at M = 8000: 1.2e-4 in the lower layers' blocks, while the same product agrees
    with float64 to 2e-7 at M = 4000 and 6000 where no row flips; tools/diag_fvp.py).  Those rows are therefore REPLACED by
    copies of a kink-free row's observation instead of masked."""
    if replace is None:
        replace = case.algo_name == "hatrpo"
    T, N = case.shapes.T, case.shapes.N
    train, model, algo = case.reference_dicts()
    cfg = O.PathConfig.from_reference_dicts(train, model, algo)
    n_masked = 0
    for a in range(case.shapes.A):
        p = {k: torch.from_numpy(v).double() for k, v in case.actor_sd[a].items()}
        obs = torch.from_numpy(case.data.obs[a][:-1].reshape(T * N, -1)).double()
        x = obs
        near = torch.zeros(T * N, dtype=torch.bool)
        with torch.no_grad():
            if "base.feature_norm.weight" in p:
                x = torch.nn.functional.layer_norm(x, (x.shape[-1],), p["base.feature_norm.weight"], p["base.feature_norm.bias"], 1e-5)
            i = 0
            while f"base.mlp.fc.{i}.weight" in p:
                z = torch.nn.functional.linear(x, p[f"base.mlp.fc.{i}.weight"], p[f"base.mlp.fc.{i}.bias"])
                near |= (z.abs().min(dim=-1).values < margin)
                x = torch.nn.functional.layer_norm(torch.relu(z), (z.shape[-1],), p[f"base.mlp.fc.{i+2}.weight"],
                                                   p[f"base.mlp.fc.{i+2}.bias"], 1e-5)
                i += 3
            if case.algo_name != "hatrpo":  # clipped surrogate: ratio near 1 +- clip_param
                d = case.data
                avail = None if d.available_actions[a] is None else torch.from_numpy(d.available_actions[a][:-1].reshape(T * N, -1)).double()
                rnn = masks = None
                if case.recurrent:
                    rnn = torch.from_numpy(d.rnn["actor"][a][0]).double()
                    masks = torch.from_numpy(d.masks[a][:-1].reshape(T * N, 1)).double()
                kind, dp = O._dist_params(p, cfg, obs, avail, rnn, masks)
                act = torch.from_numpy(d.actions[a].reshape(T * N, -1)).double()
                if kind == "categorical":
                    lp = dp[0].gather(-1, act.long())
                else:
                    lp = -((act - dp[0]) ** 2) / (2 * dp[1] * dp[1]) - torch.log(dp[1]) - 0.9189385332046727
                r_ = torch.exp(lp - torch.from_numpy(d.action_log_probs[a].reshape(T * N, -1)).double())
                imp = r_.prod(-1) if cfg.action_aggregation == "prod" else r_.mean(-1)
                for edge in (1.0 - cfg.clip_param, 1.0 + cfg.clip_param):
                    near |= ((imp - edge).abs() < margin * edge)
        nn_ = near.numpy()
        if replace and nn_.any():
            safe = np.flatnonzero(~nn_)
            ob = case.data.obs[a][:-1].reshape(T * N, -1)   # a view of the buffer: rows are overwritten in place
            assert np.shares_memory(ob, case.data.obs[a])
            bad = np.flatnonzero(nn_)
            ob[bad] = ob[safe[np.arange(len(bad)) % len(safe)]]
        else:
            am = case.data.active_masks[a]
            am[:-1].reshape(T * N, 1)[nn_] = 0.0
        n_masked += int(near.sum())
    return n_masked


_BASELINE_N_PERT = 3


def check_baseline_shape(name: str) -> Dict[str, float]:
    """One whole train() (one epoch per network) at a BASELINE.json configuration's real shapes and agent count against the
    oracle on the same seeded buffers: every agent's first update (policy loss / entropy / grad-norm / ratio, or HATRPO's
    kl / improvement figures), the sequential factor chain across all agents, the critic's update, final parameters.
    Bar per entry: max(1e-5, NOISE_FACTOR x the oracle's own fp32-vs-fp64 distance) (tests/helpers.excess)."""
    from tests.helpers import SyntheticCase
    from tests.test_oracle_golden import build_oracle
    spec = BASELINE_SHAPES[name]
    case = SyntheticCase(name, Shapes(**spec["shapes"]), spec["seed"], algo_name=spec.get("algo", "happo"),
                         overrides=spec.get("overrides"), unavailable_p=spec.get("unavailable_p", 0.0))
    out = {}
    out["_kink_adjacent_samples_masked"] = float(_mask_relu_kinks(case))
    # ---- oracle, fp32 (= the reference's arithmetic) and fp64 (the yardstick)
    runs = {}
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        O.set_work_dtype(dt)
        try:
            torch.manual_seed(case.seed)
            np.random.seed(case.seed)
            cfg, actors, critic, abufs, cbuf, vn = build_oracle(case)
            torch.manual_seed(case.seed + 12345)
            cbuf.compute_returns(cbuf.value_preds[-1].copy(), vn, cfg)
            infos, cinfo, extra = O.ha_train(actors, critic, abufs, cbuf, vn, cfg)
        finally:
            O.set_work_dtype(torch.float32)
        fin = [np.asarray(a_.flat().numpy() if case.algo_name == "hatrpo" else a_.net.flat(), dtype=np.float64) for a_ in actors]
        runs[tag] = dict(infos=infos, cinfo=cinfo, fin=fin, cfin=np.asarray(critic.net.flat(), dtype=np.float64),
                         returns=cbuf.returns.copy())
    # ---- HATRPO: the reported figures are functions of the CG solution, whose amplification of rounding differences a
    # single fp32-vs-fp64 distance samples poorly (17 agents x 5 figures: some land near 0 by chance).  The bar also takes
    # the reference's sensitivity: the same fp32 train() from initial parameters moved by one ulp (as for the goldens).
    pert = []
    if case.algo_name == "hatrpo":
        for k in range(_BASELINE_N_PERT):
            torch.manual_seed(case.seed)
            np.random.seed(case.seed)
            cfg, actors, critic, abufs, cbuf, vn = build_oracle(case)
            _perturb_one_ulp(list(actors), 977 + k)
            torch.manual_seed(case.seed + 12345)
            cbuf.compute_returns(cbuf.value_preds[-1].copy(), vn, cfg)
            infos, _, _ = O.ha_train(actors, critic, abufs, cbuf, vn, cfg)
            pert.append(dict(infos=infos, fin=[np.asarray(a_.flat().numpy(), dtype=np.float64) for a_ in actors]))
    # ---- HIP path
    torch.manual_seed(case.seed)
    np.random.seed(case.seed)
    r = build_runner(case)
    torch.manual_seed(case.seed + 12345)
    cb = r.critic_buffer
    cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
    r.prep_training()
    ginfos, gcinfo = r.train()
    torch.cuda.synchronize()
    T = case.shapes.T
    out["returns_mismatch"] = float(np.sum(cb.returns.cpu().numpy()[:T] != runs["f32"]["returns"][:T]))
    keys = (("kl", "loss_improve", "expected_improve", "dist_entropy", "ratio") if case.algo_name == "hatrpo"
            else ("policy_loss", "dist_entropy", "actor_grad_norm", "ratio"))
    tab = lambda infos_: np.array([[float(np.asarray(i[k]).reshape(-1)[0]) for k in keys] for i in infos_], dtype=np.float64)  # noqa: E731
    g, o32, o64 = tab(ginfos), tab(runs["f32"]["infos"]), tab(runs["f64"]["infos"])
    sens = None
    for pr in pert:
        d_ = np.abs(tab(pr["infos"]) - o32) / (np.abs(o32) + 1e-12)
        sens = d_ if sens is None else np.maximum(sens, d_)
    for c, k in enumerate(keys):
        out[f"_actor_{k}_rel"] = rel_err(g[:, c], o32[:, c])
        out[f"actor_{k}_excess"] = excess(g[:, c], o32[:, c], o64[:, c], sens=None if sens is None else sens[:, c])
        out[f"_actor_{k}_at"] = excess_at(g[:, c], o32[:, c], o64[:, c], sens=None if sens is None else sens[:, c])
    gc = [gcinfo["value_loss"], gcinfo["critic_grad_norm"]]
    c32 = [runs["f32"]["cinfo"]["value_loss"], runs["f32"]["cinfo"]["critic_grad_norm"]]
    c64 = [runs["f64"]["cinfo"]["value_loss"], runs["f64"]["cinfo"]["critic_grad_norm"]]
    out["_critic_rel"] = rel_err(gc, c32)
    out["critic_info_excess"] = excess(gc, c32, c64)
    worst = 0.0
    for a in range(case.shapes.A):
        fp = r.actor[a].actor.flat_param.cpu().numpy()
        ps = max([vec_rel_err(pr["fin"][a], runs["f32"]["fin"][a]) for pr in pert], default=None)
        worst = max(worst, vec_excess(fp, runs["f32"]["fin"][a], runs["f64"]["fin"][a], sens=ps))
        out["_actor_final_param_vec_rel_max"] = max(out.get("_actor_final_param_vec_rel_max", 0.0), vec_rel_err(fp, runs["f32"]["fin"][a]))
    out["actor_final_param_excess"] = worst
    out["critic_final_param_excess"] = vec_excess(r.critic.critic.flat_param.cpu().numpy(), runs["f32"]["cfin"], runs["f64"]["cfin"])
    return out


def check_trpo_upstream(name: str = "humanoid17", agents=(0, 8, 16)) -> Dict[str, float]:
    """HATRPO at a BASELINE shape, UPSTREAM of the end of the conjugate-gradient solve: for a few agents, from
    identical parameters and buffer contents, the surrogate gradient vector, the Fisher-vector product on three random vectors
    and the CG iterate after 1, 5 and 10 steps -- each as the HIP path's distance from the oracle in float64 next to the fp32
    oracle's own distance from it (symmetric yardstick: no perturbation bars).  Asserted (keys without "_"): gradient and
    F.v within 1e-5 of the vector's inf-norm of the fp64 value; the CG iterates within 4 x the fp32 oracle's own distance
    (they amplify rounding by the condition number of F; the distances are reported).  The samples an fp64 forward puts within
    2e-5 of a ReLU kink are masked as in check_baseline_shape; the same figures WITHOUT the masking are reported under
    `_unmasked_*` (not asserted: one such sample moves a gradient tensor by up to a few % in any fp32 implementation)."""
    from tests.helpers import SyntheticCase
    from tests.test_oracle_golden import build_oracle
    spec = BASELINE_SHAPES[name]
    out: Dict[str, float] = {}
    rng = np.random.default_rng(123)
    for masked in (True, False):
        case = SyntheticCase(name, Shapes(**spec["shapes"]), spec["seed"], algo_name=spec.get("algo", "happo"),
                             overrides=spec.get("overrides"), unavailable_p=spec.get("unavailable_p", 0.0))
        if masked:
            out["_kink_adjacent_samples_masked"] = float(_mask_relu_kinks(case))
        sh = case.shapes
        M = sh.T * sh.N
        torch.manual_seed(case.seed)
        np.random.seed(case.seed)
        r = build_runner(case)
        r.prep_training()
        pre = "" if masked else "_unmasked_"
        worst = dict(grad=0.0, fvp=0.0, cg1=0.0, cg5=0.0, cg10=0.0)
        ref32 = dict(worst)
        for a in agents:
            d = case.data
            obs = d.obs[a][:-1].reshape(M, -1)
            act = d.actions[a].reshape(M, -1)
            old_logp = d.action_log_probs[a].reshape(M, -1)
            active = d.active_masks[a][:-1].reshape(M, 1)
            adv = rng.standard_normal((M, 1)).astype(np.float32)
            factor = (1 + 0.1 * rng.standard_normal((M, 1))).astype(np.float32)
            vs = [rng.standard_normal(sum(int(np.prod(v.shape)) for v in case.actor_sd[a].values())).astype(np.float32)
                  for _ in range(3)]
            sample = (obs, act, active, old_logp, adv, None, factor)
            ora = {}
            for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
                O.set_work_dtype(dt)
                try:
                    cfg, actors, _, _, _, _ = build_oracle(case)
                    o = actors[a]
                    t = lambda x: None if x is None else torch.from_numpy(x).to(dt)  # noqa: E731
                    fv = [o.fvp(t(obs), None, torch.from_numpy(v).to(dt)).numpy().astype(np.float64) for v in vs]
                    torch.manual_seed(1)
                    info = o.update(tuple(None if x is None else x.astype(np.float64 if dt == torch.float64 else np.float32)
                                          for x in sample))
                    ora[tag] = dict(grad=info["grad"].astype(np.float64), fvp=fv,
                                    cg={k: v.astype(np.float64) for k, v in info["cg_x"].items()})
                finally:
                    O.set_work_dtype(torch.float32)
            actor = r.actor[a]
            actor.actor.fold()
            d_obs = dev(obs)
            gfv = []
            sc, g = actor._surrogate(d_obs, M, dev(act), None, dev(old_logp), dev(adv.reshape(M)), None, dev(factor.reshape(M)),
                                     dev(active.reshape(M)), want_grad=True)
            g = g.cpu().numpy().astype(np.float64)
            for v in vs:
                gfv.append(actor._fvp(d_obs, M, M, None, dev(v)).cpu().numpy().astype(np.float64))
            taps = {}
            actor._cg_tap = lambda k, x: taps.__setitem__(k, x.cpu().numpy().astype(np.float64))
            torch.manual_seed(1)
            actor.update((obs, np.zeros((M, 1, 1), dtype=np.float32), act, None, active, old_logp, adv, None, factor))
            torch.cuda.synchronize()
            actor._cg_tap = None
            nrm = lambda x: float(np.max(np.abs(x)))  # noqa: E731
            e = lambda x, y: nrm(x - y) / (nrm(y) + 1e-300)  # noqa: E731
            worst["grad"] = max(worst["grad"], e(g, ora["f64"]["grad"]))
            ref32["grad"] = max(ref32["grad"], e(ora["f32"]["grad"], ora["f64"]["grad"]))
            for k in range(3):
                worst["fvp"] = max(worst["fvp"], e(gfv[k], ora["f64"]["fvp"][k]))
                ref32["fvp"] = max(ref32["fvp"], e(ora["f32"]["fvp"][k], ora["f64"]["fvp"][k]))
            for k in (1, 5, 10):
                if k in taps and k in ora["f64"]["cg"] and k in ora["f32"]["cg"]:
                    worst[f"cg{k}"] = max(worst[f"cg{k}"], e(taps[k], ora["f64"]["cg"][k]))
                    ref32[f"cg{k}"] = max(ref32[f"cg{k}"], e(ora["f32"]["cg"][k], ora["f64"]["cg"][k]))
        for k in worst:
            out[f"_{pre.strip('_')}_{k}_hip_vs_f64".replace("__", "_")] = worst[k]
            out[f"_{pre.strip('_')}_{k}_ref32_vs_f64".replace("__", "_")] = ref32[k]
        if masked:
            out["grad_vs_f64"] = worst["grad"]      # asserted < 1e-5 of the inf-norm
            out["fvp_vs_f64"] = worst["fvp"]
            for k in (1, 5, 10):                    # asserted <= 1: within 4 x the fp32 oracle's own distance (floor 1e-5)
                out[f"cg{k}_excess"] = worst[f"cg{k}"] / max(4.0 * ref32[f"cg{k}"], 1e-5)
        del r
    return out


def check_buffer_slots(case: str) -> Dict[str, float]:
    """Slot contents of the rollout buffers after T + 2 runner.insert() calls with after_update() in between, against the
    arrays recorded from the REFERENCE's own insert()/after_update() on the same seeded inputs (oracle/gen_buffer_golden.py:
    on_policy_base_runner.py:342-460, on_policy_actor_buffer.py:49-96, on_policy_critic_buffer_{ep,fp}.py).  Bit-exact."""
    from oracle.gen_buffer_golden import ACTOR_KEYS, CASES, CRITIC_KEYS, snapshot, step_inputs
    from harl_amd.buffers import OnPolicyActorBuffer, OnPolicyCriticBufferEP, OnPolicyCriticBufferFP
    from harl_amd.runner import OnPolicyHARunner
    st, disc, rn, T, N, A, D, S, act, H = CASES[case]
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "buffers", f"{case}.npz"))
    args = dict(episode_length=T, n_rollout_threads=N, hidden_sizes=[H, H], recurrent_n=rn, gamma=0.99, gae_lambda=0.95,
                use_gae=True, use_proper_time_limits=True)
    space = Discrete(act) if disc else Box((act,))
    r = OnPolicyHARunner.__new__(OnPolicyHARunner)  # the buffers and insert() only: no networks of width 8 exist
    r.device, r.num_agents, r.state_type = torch.device(DEV), A, st
    r.actor_buffer = [OnPolicyActorBuffer(args, Box((D,)), space, device=DEV) for _ in range(A)]
    r.critic_buffer = (OnPolicyCriticBufferEP(args, Box((S,)), device=DEV) if st == "EP"
                       else OnPolicyCriticBufferFP(args, Box((S,)), A, device=DEV))
    out, bad = {}, []
    get = lambda t: t.detach().cpu().numpy()  # noqa: E731

    def compare(prefix):
        snap = snapshot(r.actor_buffer, r.critic_buffer, get)
        for k, v in snap.items():
            g = z[f"{prefix}_{k}"]
            if v.shape != g.shape or not np.array_equal(v, g.astype(v.dtype)):
                bad.append(f"{prefix}_{k}")
        missing = [k for k in z.files if k.startswith(prefix + "_") and k[len(prefix) + 1:] not in snap]
        bad.extend(missing)

    for step in range(T + 2):
        obs, share, rewards, dones, infos, avail, values, actions, logp, rnn, rnn_c = step_inputs(case, step)
        r.insert((obs, share, rewards, dones, infos, avail, dev(values), dev(actions), dev(logp), dev(rnn), dev(rnn_c)))
        if step == T - 1:
            compare("full")
            r.after_update()
    compare("end")
    out["buffer_slot_mismatch"] = float(len(bad))
    out["_mismatched"] = ",".join(bad)
    return out


def check_run_eval_save_restore(tmpdir: str) -> Dict[str, float]:
    """run() with the reference's end-of-episode block (on_policy_base_runner.py:252-258): every ``eval_interval`` episodes
    eval() on ``eval_envs`` (deterministic actions, logger callbacks) and save(); a second runner constructed with
    ``train.model_dir`` pointing at the checkpoints restores them in its constructor (:168-169) -- parameters and ValueNorm
    statistics bit-identical."""
    from harl_amd.runner import OnPolicyHARunner
    from tests.fake_env import FakeVecEnv
    torch.manual_seed(4)
    np.random.seed(4)
    N, T = 64, 25
    a = default_args([64, 64])
    model = {k: a[k] for k in ("hidden_sizes", "activation_func", "use_feature_normalization", "initialization_method",
                                "gain", "use_naive_recurrent_policy", "use_recurrent_policy", "recurrent_n",
                                "data_chunk_length", "lr", "critic_lr", "opti_eps", "weight_decay", "std_x_coef", "std_y_coef")}
    algo = {k: v for k, v in a.items() if k not in model}

    class Log:  # the logger surface run() / eval() call (common/base_logger.py)
        def __init__(self):
            self.n = {}

        def __getattr__(self, name):
            def f(*args, **kw):
                self.n[name] = self.n.get(name, 0) + 1
            return f

    def make(model_dir=None):
        train = dict(n_rollout_threads=N, episode_length=T, use_valuenorm=True, use_proper_time_limits=True,
                     use_linear_lr_decay=True, num_env_steps=N * T * 4, log_interval=1, eval_interval=2, model_dir=model_dir)
        ev = dict(use_eval=True, n_eval_rollout_threads=8, eval_episodes=16)
        lg = Log()
        r = OnPolicyHARunner(dict(algo="happo"), dict(train=train, model=model, algo=algo, eval=ev), dict(state_type="EP"),
                             envs=FakeVecEnv(N, n_agents=3, state_dim=6, act_dim=2, horizon=25, seed=5),
                             eval_envs=FakeVecEnv(8, n_agents=3, state_dim=6, act_dim=2, horizon=25, seed=6),
                             logger=lg, save_dir=os.path.join(tmpdir, "models"), device=DEV)
        return r, lg

    r, lg = make()
    hist = r.run()
    torch.cuda.synchronize()
    out = {"episodes_mismatch": float(len(hist) != 4), "eval_calls_mismatch": float(lg.n.get("eval_log", 0) != 2),
           "eval_steps_missing_count": float(lg.n.get("eval_per_step", 0) < 2 * 2 * 25 // 1 // 8)}
    files = set(os.listdir(os.path.join(tmpdir, "models")))
    want = {"actor_agent0.pt", "actor_agent1.pt", "actor_agent2.pt", "critic_agent.pt", "value_normalizer.pt"}
    out["checkpoint_files_missing_count"] = float(len(want - files))
    r2, _ = make(model_dir=os.path.join(tmpdir, "models"))
    d = 0.0
    for a1, a2 in zip(r.actor, r2.actor):
        d = max(d, float((a1.actor.flat_param - a2.actor.flat_param).abs().max().item()))
    d = max(d, float((r.critic.critic.flat_param - r2.critic.critic.flat_param).abs().max().item()))
    d = max(d, float((r.value_normalizer.stats - r2.value_normalizer.stats).abs().max().item()))
    out["restored_param_max_abs"] = d
    # render(): deterministic episodes of the restored policy on a vectorised environment (on_policy_base_runner.py:594-710)
    r2.algo_args["render"] = dict(render_episodes=2)
    r2.envs = FakeVecEnv(4, n_agents=3, state_dim=6, act_dim=2, horizon=25, seed=9)
    rets = r2.render()
    out["render_episodes_mismatch"] = float(len(rets) != 2 or not all(np.isfinite(rets)))
    r.close()
    r2.close()
    return out
