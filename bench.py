#!/usr/bin/env python3
"""bench.py -- transitions/sec through the on-policy sequential update (compute_returns + OnPolicyHARunner.train()).

Default workload = BASELINE.json configs[1]: MPE simple_spread_v2, 3 agents, HAPPO, n_rollout_threads=4096 per GPU,
episode_length=200, obs 18 / share_obs 54 / Box(5), MLP [128,128], happo.yaml defaults (ppo_epoch=5, critic_epoch=5,
1 mini-batch, ValueNorm + GAE + proper time limits, Huber, clip 0.2, max_grad_norm 10).  `--config` selects the other
BASELINE.json configurations at their real shapes (one JSON line each):

    mpe         configs[1]  MPE simple_spread, 3 agents, HAPPO                         T=200 N=4096/GPU
    cheetah6    configs[2]  MAMuJoCo HalfCheetah-6x1, 6 agents, HAPPO, MLP [128]x3      T=200 N=4096/GPU (8192 global with N > 1 ranks)
    smac3s5z    configs[3]  SMAC 3s5z, 8 agents, HAPPO, GRU policy, Discrete(14)        T=160 N=512/GPU, chunks of 10  (smac3s5z_n4096: N=4096)
    humanoid17  configs[4]  MAMuJoCo Humanoid-17x1, 17 agents, HATRPO, obs 393          T=200 N=1024/GPU  (humanoid17_n4096: the north_star's N=4096 on one GPU)
    hatrpo_gru128  (not in BASELINE.json) SMAC shape, HATRPO, MLP [128,128] + 128-wide GRU: the composed coverage path

A "step" = one compute() + train() over synthetic rollout buffers (SURVEY.md 8d recipe) resident in HBM before the timed
region.  One transition = one (t, n) environment step for all agents.
Multi-GPU (`--gpus N`, N > 1; `--scaling auto`, the default): the ONE line answers the BASELINE's question -- `value` is the
STRONG-scaling figure at the n_rollout_threads the BASELINE quotes the workload on (4096 global for mpe, 8192 for cheetah6 =
configs[2], split over the ranks; `scaling: "strong"`, `config.n_rollout_threads_global`), and the `weak` object carries the
weak-scaling figure of the same job (every rank at the config's single-GPU size), timed in a region of its own right after.
`config.ranks_seen` = a SUM all-reduce of ones over the update's communicator (not WORLD_SIZE from the environment).
`--scaling weak|strong` time only that one figure (`--global-threads`, `--threads-per-gpu` override the sizes).
Gradients / loss scalars are all-reduced each optimiser step (one message per step, harl_amd/dist.py: RCCL, or the one-hop
hipIpc exchange with HARL_ALLREDUCE=auto|oneshot);
`--dist-single` initialises the nccl(=RCCL) process group even with one rank so that branch runs on a 1-GPU box.

    python bench.py --gpus 1 --steps 5 --warmup 2 [--config cheetah6]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus 8 --steps 5 --warmup 2

The default run (mpe) appends the three other BASELINE workloads and the coverage workload `hatrpo_gru128` (5 steps each, no events inside their timed regions, fresh
runners, AFTER the headline region and the CPU baseline; each with its own bounded CPU baseline) to the same line as
`other_configs` (`--no-other-configs` to skip).

Prints ONE JSON line (rank 0).  `roofline`: the streaming kernel family with the largest total time per step (decided in the
last warm-up step, where every family is bracketed by HIP events; inside the timed region only that family is) -- achieved =
algorithmic HBM bytes of its launches in the timed region (harl_amd/traffic.py, from each launch's own arguments) / their
HIP-event time; `traffic` = measured HBM bytes per launch from the committed PMC pass of that kernel
(the newest profiles/r0N_hbm_traffic.json, stamped with the commit it was taken at).  `kernels`: full per-kernel breakdown from extra
instrumented steps after the timed region.  `cpu_baseline`: the oracle (torch-CPU restatement of the reference, same ATen
kernels) on this box's host cores on a bounded sample of the same workload, warm-up + best of 3.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_F32_PEAK = 157.3e12  # MI355X_MICROARCH.md: fp32-input MFMA, dense
MFMA_BF16_PEAK = 2516.6e12  # dense bf16 MFMA (1024 FLOP/clk/SIMD at 2.4 GHz)
HBM_PEAK = 8.0e12

WORKLOADS = {
    "mpe": dict(algo="happo", T=200, N=4096, A=3, obs=18, sobs=54, act=5, disc=False, hidden=[128, 128],
                metric="transitions/sec through HAPPO update (MPE spread, 3 agents)",
                name="MPE simple_spread_v2 3-agent HAPPO update"),
    "cheetah6": dict(algo="happo", T=200, N=4096, A=6, obs=23, sobs=17, act=1, disc=False, hidden=[128, 128, 128],
                     metric="transitions/sec through HAPPO update (MAMuJoCo HalfCheetah-6x1, 6 agents)",
                     name="MAMuJoCo HalfCheetah-6x1 6-agent HAPPO update"),
    "smac3s5z": dict(algo="happo", T=160, N=512, A=8, obs=128, sobs=216, act=14, disc=True, hidden=[64, 64, 64], rnn=True, L=10,
                     unavailable_p=0.3, metric="transitions/sec through HAPPO update (SMAC 3s5z, 8 agents, GRU policy)",
                     name="SMAC 3s5z 8-agent recurrent HAPPO update"),
    "humanoid17": dict(algo="hatrpo", T=200, N=1024, A=17, obs=393, sobs=376, act=1, disc=False, hidden=[128, 128, 128],
                       metric="transitions/sec through HATRPO update (MAMuJoCo Humanoid-17x1, 17 agents)",
                       name="MAMuJoCo Humanoid-17x1 17-agent HATRPO update"),
    # north_star: "(episode_length=200, n_rollout_threads=4096, n_agents in {3,6,17})" -- the 17-agent update at the full 4096
    # rollout threads on ONE GPU (22 GB of observations, ~85 GB of per-agent activation workspaces; VERDICT r05 missing 2), next
    # to the 8-GPU share above; and the recurrent workload at 4096 threads next to its tuned-config 512
    "humanoid17_n4096": dict(algo="hatrpo", T=200, N=4096, A=17, obs=393, sobs=376, act=1, disc=False, hidden=[128, 128, 128],
                             base="humanoid17",
                             metric="transitions/sec through HATRPO update (MAMuJoCo Humanoid-17x1, 17 agents)",
                             name="MAMuJoCo Humanoid-17x1 17-agent HATRPO update"),
    "smac3s5z_n4096": dict(algo="happo", T=160, N=4096, A=8, obs=128, sobs=216, act=14, disc=True, hidden=[64, 64, 64], rnn=True,
                           L=10, unavailable_p=0.3, base="smac3s5z",
                           metric="transitions/sec through HAPPO update (SMAC 3s5z, 8 agents, GRU policy)",
                           name="SMAC 3s5z 8-agent recurrent HAPPO update"),
    # not a BASELINE configuration: hatrpo.yaml's DEFAULT model ([128, 128]) with use_recurrent_policy at the SMAC shape -- the
    # composed 128-wide GRU (per-step launches, harl_amd/gru_wide.py) under HATRPO's tangent passes, the slowest coverage path
    # (it had no number)
    "hatrpo_gru128": dict(algo="hatrpo", T=160, N=512, A=8, obs=128, sobs=216, act=14, disc=True, hidden=[128, 128], rnn=True,
                          L=10, unavailable_p=0.3,
                          metric="transitions/sec through HATRPO update (SMAC-shaped, 8 agents, 128-wide GRU policy)",
                          name="SMAC-shaped 8-agent recurrent HATRPO update, hatrpo.yaml default widths (coverage path)"),
}
# n_rollout_threads of the WHOLE job the BASELINE quotes each workload on (`--scaling strong` / the multi-GPU headline): configs[1]
# = 4096 for MPE, configs[2] = 8192 for HalfCheetah-6x1; the others: their single-GPU size
BASELINE_GLOBAL_THREADS = {"mpe": 4096, "cheetah6": 8192}
# module-level aliases of the default workload (tools/ and older scripts import these)
T, N_PER_GPU, A = 200, 4096, 3
OBS, SOBS, ACT = 18, 54, 5
HIDDEN = [128, 128]


def algo_args(n_threads: int, T_: int = T, w: dict | None = None) -> dict:
    w = w or WORKLOADS["mpe"]
    args = dict(
        train=dict(n_rollout_threads=n_threads, episode_length=T_, use_valuenorm=True, use_proper_time_limits=True),
        model=dict(hidden_sizes=list(w["hidden"]), activation_func="relu", use_feature_normalization=True,
                   initialization_method="orthogonal_", gain=0.01, use_naive_recurrent_policy=False,
                   use_recurrent_policy=bool(w.get("rnn")), recurrent_n=1, data_chunk_length=w.get("L", 10), lr=5e-4,
                   critic_lr=5e-4, opti_eps=1e-5, weight_decay=0, std_x_coef=1, std_y_coef=0.5),
        algo=dict(ppo_epoch=5, critic_epoch=5, use_clipped_value_loss=True, clip_param=0.2, actor_num_mini_batch=1,
                  critic_num_mini_batch=1, entropy_coef=0.01, value_loss_coef=1, use_max_grad_norm=True,
                  max_grad_norm=10.0, use_gae=True, gamma=0.99, gae_lambda=0.95, use_huber_loss=True,
                  use_policy_active_masks=True, huber_delta=10.0, action_aggregation="prod", share_param=False,
                  fixed_order=True),
    )
    if w["algo"] == "hatrpo":  # hatrpo.yaml defaults
        args["algo"].update(kl_threshold=0.01, ls_step=10, accept_ratio=0.5, backtrack_coeff=0.8)
    return args


class Box:
    def __init__(self, shape):
        self.shape = shape


class Discrete:
    def __init__(self, n):
        self.n = n


def flops_per_transition(w: dict) -> float:
    """Linear FLOPs only.  HAPPO: (2 + 3 ppo_epoch) F_actor per agent + 3 critic_epoch F_critic (forward = 1,
    backward = 2; two log-prob passes per agent).  Not reported for HATRPO (CG iteration count is data dependent)."""
    def mlp(d):
        f, prev = 0, d
        for h in w["hidden"]:
            f += 2 * prev * h
            prev = h
        if w.get("rnn"):
            f += 2 * 6 * prev * prev
        return f, prev
    fa, pa = mlp(w["obs"])
    fc, pc = mlp(w["sobs"])
    fa += 2 * pa * w["act"]
    fc += 2 * pc
    return w["A"] * (2 + 3 * 5) * fa + 3 * 5 * fc


def fill_buffers(r, w: dict, n_local: int, rank: int, device, logp: str) -> None:
    """SURVEY.md 8d recipe, generated on the device (the 17-agent configuration holds 7 GB of observations): obs /
    share_obs / rewards / value_preds ~ N(0,1); Box actions ~ N(0,1) with stored log-probs -1 + 0.1 N(0,1); Discrete actions
    uniform with log-probs log(1/n) + 0.05 N(0,1); masks 0 w.p. 0.04, bad_masks 0 at the same places; GRU states 0.3 N(0,1).
    `logp = onpolicy` replaces the stored log-probs by log pi(a|o) + 0.05 N(0,1) under the initial weights (ratios ~ 1)."""
    g = torch.Generator(device=device)
    g.manual_seed(100 + rank)
    Tn = w["T"]
    rn = lambda *s: torch.randn(*s, generator=g, device=device)  # noqa: E731
    base_mask = (torch.rand(Tn + 1, n_local, 1, generator=g, device=device) >= 0.04).float()
    for a in range(w["A"]):
        b = r.actor_buffer[a]
        b.obs.copy_(rn(*b.obs.shape))
        if w["disc"]:
            act = torch.randint(0, w["act"], (Tn, n_local, 1), generator=g, device=device)
            b.actions.copy_(act.float())
            b.action_log_probs.copy_(float(np.log(1.0 / w["act"])) + 0.05 * rn(Tn, n_local, 1))
            av = (torch.rand(Tn + 1, n_local, w["act"], generator=g, device=device) >= w.get("unavailable_p", 0.0)).float()
            av[:-1].scatter_(-1, act, 1.0)  # the taken action always stays available
            b.available_actions.copy_(av)
        else:
            b.actions.copy_(rn(*b.actions.shape))
            b.action_log_probs.copy_(-1.0 + 0.1 * rn(*b.action_log_probs.shape))
        b.masks.copy_(base_mask)
        b.active_masks.fill_(1.0)
        if w.get("rnn"):
            b.rnn_states.copy_(0.3 * rn(*b.rnn_states.shape))
        if logp == "onpolicy":
            act_ = r.actor[a]
            B = Tn * n_local
            lp = torch.empty(B, act_.actor.act_w, device=device)
            act_.actor.fold()
            kw = dict(rnn_states=b.rnn_states[0], masks=b.flat("masks")) if w.get("rnn") else {}
            act_._logp_pass(b.flat("obs"), b.flat("actions"), None if b.available_actions is None else b.flat("available_actions"),
                            B, lp, **kw)
            b.action_log_probs.copy_((lp + 0.05 * rn(*lp.shape)).reshape(b.action_log_probs.shape))
    cb = r.critic_buffer
    cb.share_obs.copy_(rn(*cb.share_obs.shape))
    cb.rewards.copy_(rn(*cb.rewards.shape))
    cb.value_preds.copy_(rn(*cb.value_preds.shape))
    cb.masks.copy_(base_mask)
    cb.bad_masks.copy_(base_mask)
    if w.get("rnn"):
        cb.rnn_states_critic.copy_(0.3 * rn(*cb.rnn_states_critic.shape))


def build_gpu_runner(w: dict, n_local: int, rank: int, world: int, device, logp: str = "recipe"):
    from harl_amd.runner import RUNNER_REGISTRY

    args = algo_args(n_local * world, w["T"], w)
    torch.manual_seed(1)
    np.random.seed(1)
    space = Discrete(w["act"]) if w["disc"] else Box((w["act"],))
    r = RUNNER_REGISTRY[w["algo"]](dict(algo=w["algo"]), args, dict(state_type="EP"), obs_spaces=[Box((w["obs"],))] * w["A"],
                                   share_obs_space=Box((w["sobs"],)), act_spaces=[space] * w["A"], device=device)
    fill_buffers(r, w, n_local, rank, device, logp)
    r.prep_training()
    return r


def one_step(r) -> None:
    # In training every step sees a freshly filled rollout buffer; the synthetic buffers here never change.  train()
    # itself drops the per-buffer caches (the normalised-input images, nets.invalidate_caches) on entry, so that work
    # stays inside every timed step; the explicit calls only cover compute()'s critic forward.
    for a in r.actor:
        a.actor.invalidate_caches()
    r.critic.critic.invalidate_caches()
    r.compute()
    r.train()


def cpu_baseline(w: dict, n_cols: int, threads: int, reps: int = 3) -> dict:
    """The oracle (torch-CPU restatement of the reference path, same ATen kernels / autograd / Adam) on a bounded
    sample of the same workload: same shapes with fewer rollout threads.  One untimed warm-up update, then the best of
    `reps` timed ones (each from freshly built networks / buffers).  Reported, not a target."""
    from harl_amd.synthetic import Shapes, actor_param_shapes, critic_param_shapes, make_buffers, synthetic_state_dict
    from oracle import harl_oracle as O

    torch.set_num_threads(threads)
    Tn = w["T"]
    args = algo_args(n_cols, Tn, w)
    cfg = O.PathConfig.from_reference_dicts(args["train"], args["model"], args["algo"])
    sh = Shapes(T=Tn, N=n_cols, A=w["A"], obs_dim=w["obs"], share_obs_dim=w["sobs"], act_dim=w["act"], discrete=w["disc"],
                hidden_sizes=w["hidden"])
    rnn = bool(w.get("rnn"))
    d = make_buffers(sh, seed=100, unavailable_p=w.get("unavailable_p", 0.0), rnn=rnn)
    tc = (O.TrpoConfig(**{k: args["algo"][k] for k in ("kl_threshold", "ls_step", "accept_ratio", "backtrack_coeff")})
          if w["algo"] == "hatrpo" else None)

    def once() -> float:
        sds = [{k: torch.from_numpy(v) for k, v in synthetic_state_dict(actor_param_shapes(sh, True, rnn), 10 + a).items()}
               for a in range(w["A"])]
        actors = [O.OracleHATRPO(sd, cfg, tc) if tc is not None else O.OracleHAPPO(sd, cfg) for sd in sds]
        critic = O.OracleVCritic({k: torch.from_numpy(v) for k, v in
                                  synthetic_state_dict(critic_param_shapes(sh, True, rnn), 99).items()}, cfg)
        abufs = [O.OracleActorBuffer(d.obs[a], d.actions[a], d.action_log_probs[a], d.masks[a], d.active_masks[a],
                                     d.available_actions[a], rnn_states=None if not rnn else d.rnn["actor"][a])
                 for a in range(w["A"])]
        cbuf = O.OracleCriticBufferEP(d.share_obs, d.rewards, d.value_preds.copy(), d.critic_masks, d.bad_masks)
        if rnn:
            cbuf.rnn_states_critic = d.rnn["critic"]
        vn = O.OracleValueNorm()
        t0 = time.perf_counter()
        cbuf.compute_returns(cbuf.value_preds[-1].copy(), vn, cfg)
        O.ha_train(actors, critic, abufs, cbuf, vn, cfg)
        return time.perf_counter() - t0

    once()  # warm-up (thread pool, allocator, first-touch of the buffers)
    ts = [once() for _ in range(reps)]
    dt = min(ts)
    return dict(value=Tn * n_cols / dt, unit="transitions/s", cores=threads, host_cpu_count=os.cpu_count(), kind="port",
                runs_s=[round(t, 2) for t in ts],
                sample=f"1 update at T={Tn}, n_rollout_threads={n_cols} (same nets/epochs; oracle = torch-CPU restatement of the "
                       f"reference); 1 warm-up + best of {reps} ({dt:.2f} s); {threads} torch threads of {os.cpu_count()} host CPUs "
                       "(these nets are small: more threads is slower)")


def ranks_seen(comm, device) -> int:
    """Number of ranks that answer on the update's communicator: a SUM all-reduce of ones through the same path the gradients
    take (RCCL, or the one-shot exchange) -- not WORLD_SIZE read back from the environment."""
    t = torch.ones(1, dtype=torch.float32, device=device)
    comm.all_reduce_sum(t)
    torch.cuda.synchronize()
    return int(round(float(t.item())))


def measure(w: dict, cfg_name: str, args, comm, rank: int, world: int, device, n_local: int, steps: int, warmup: int,
            instr_steps: int, scaling: str = "weak", region_events: bool = True):
    """Build the runner of one workload, warm up, time `steps` steps between barriers and collect the per-kernel figures.
    Returns the JSON record of this workload on rank 0 (None elsewhere).
    ``region_events`` = False (the workloads ATTACHED to the default line as `other_configs`): no HIP events inside the timed
    region -- their `roofline` comes from the instrumented step behind it.  An event pair between two launches keeps the second
    from starting under the first one's tail; for the headline's 15 long launches per step that is noise, for the recurrent
    workload's 62 short ones it was 1.2 ms of a 21.5 ms step (22.7 in the line against 21.5 from `--config smac3s5z
    --no-kernel-timing` on the same box, round 6)."""
    from harl_amd import _lib

    Tn = w["T"]
    r = build_gpu_runner(w, n_local, rank, world, device, args.logp)

    def barrier():
        torch.cuda.synchronize()
        if comm.enabled:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Roofline timing lives INSIDE the timed region: every launch of the DOMINANT streaming kernel family is bracketed by HIP
    # events on its launch stream.  Which family that is, is decided in the last warm-up step, where all of them are
    # bracketed (single stream): an event pair between two kernels keeps the second from starting under the first one's tail, and bracketing
    # all ~20 families inside the region cost 2.5-3 % of the step (MPE; 10-15 % for the launch-heavy 17-agent HATRPO step).
    ROOF_TAGS = ("fwd_fused2", "fwd_fused2_k64", "fwd_hidden", "bwd_dx", "bwd_dx_dw1", "bwd_full", "bwd_full_dw1", "dw_hidden", "fwd_wide", "dw_input",
                 "tangent_wide", "tangent_hidden", "gru_fwd", "gru_bwd", "update_fwd", "update_bwd", "update_logp",
                 "update_fwd_critic", "update_last", "update_last_critic", "fwd_panel", "bwd_panel", "fwd_trunk", "bwd_trunk",
                 "dw_trunk", "dw_gru")
    warm_kern = {}
    for k in range(warmup):
        last = k == warmup - 1 and not args.no_kernel_timing and not args.time_all_tags
        if last:  # (critic chain on the main stream for this step: per-launch durations without a second kernel sharing the chip)
            prev_cs = os.environ.get("HARL_CRITIC_STREAM")
            os.environ["HARL_CRITIC_STREAM"] = "0"
            _lib.enable_kernel_timing(True, ROOF_TAGS)
        one_step(r)
        if last:
            warm_kern = _lib.collect_kernel_timing()
            _lib.enable_kernel_timing(False)
            if prev_cs is None:
                os.environ.pop("HARL_CRITIC_STREAM", None)
            else:
                os.environ["HARL_CRITIC_STREAM"] = prev_cs
    barrier()
    # Shader clock under THIS load (one untimed step more): a one-lane probe kernel on a side stream counts shader cycles
    # against the constant 100 MHz counter while a whole step runs next to it (harl_clock_probe).  The matrix-pipe fractions
    # below are computed against this clock; the nominal 2.4 GHz figure stays next to them as `*_nominal`.
    clock_ghz = None
    if not args.no_kernel_timing:
        t_est = time.perf_counter()
        one_step(r)
        torch.cuda.synchronize()
        est_ms = (time.perf_counter() - t_est) * 1e3
        probe_out = torch.zeros(2, dtype=torch.int64, device=device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        _lib.call("harl_clock_probe", _lib.ptr(probe_out), int(max(0.5, min(0.85 * est_ms, 900.0)) * 1e5), side.cuda_stream)
        one_step(r)
        torch.cuda.synchronize()
        cyc, ticks = (int(v) for v in probe_out.tolist())
        if ticks > 0 and cyc > 0:
            clock_ghz = cyc / (ticks * 10e-9) / 1e9
        barrier()
    region_tags = ROOF_TAGS
    cw = {k: v for k, v in warm_kern.items() if v["n"] > 0 and v.get("bytes")}
    if cw:
        region_tags = (max(cw, key=lambda k: cw[k]["total_ms"]),)
    if not args.no_kernel_timing and region_events:
        if cw:  # two events per bracketed launch, created before the clock starts
            _lib.reserve_timing_events(2 * (cw[region_tags[0]]["n"] + 8) * steps)
        _lib.enable_kernel_timing(True, None if args.time_all_tags else region_tags)
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step(r)
    barrier()
    dt = time.perf_counter() - t0
    roof_kern = {}
    if not args.no_kernel_timing and region_events:
        roof_kern = _lib.collect_kernel_timing()
        _lib.enable_kernel_timing(False)
    if comm.enabled:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    # Full per-kernel breakdown (every tagged launch, ~10 % of wall time): `instr_steps` further steps of the SAME
    # workload after the timed region, so that it does not understate `value`.
    kern = dict(roof_kern) if args.time_all_tags else {}
    if not args.no_kernel_timing and instr_steps > 0:
        # (ONE stream here: in the timed region the critic's update runs on a stream of its own next to the actors' chain,
        # so two kernels share the chip and every per-launch duration contains its neighbour's slices; the per-kernel table
        # and `roofline.single_stream` are taken with the critic chain back on the main stream)
        prev_cs = os.environ.get("HARL_CRITIC_STREAM")
        os.environ["HARL_CRITIC_STREAM"] = "0"
        _lib.enable_kernel_timing(True)
        for _ in range(instr_steps):
            one_step(r)
        kern = _lib.collect_kernel_timing()
        _lib.enable_kernel_timing(False)
        if prev_cs is None:
            os.environ.pop("HARL_CRITIC_STREAM", None)
        else:
            os.environ["HARL_CRITIC_STREAM"] = prev_cs

    if not region_events:  # attached workloads: the streaming families of the instrumented step behind the region
        roof_kern = {t: x for t, x in kern.items() if t in ROOF_TAGS}
    n_seen = ranks_seen(comm, device) if comm.enabled else 1  # (collective: every rank)
    if rank != 0:
        return None
    if True:
        trans_per_step = Tn * n_local * world
        value = trans_per_step * steps / dt
        # dominant kernel = the streaming family with the largest total time inside the timed region; `achieved` =
        # algorithmic bytes of the launches that ran (harl_amd/traffic.py) / their HIP-event time.  The GEMMs run on the
        # bf16 matrix pipe (exact three-way fp32 split, 6 products: csrc/split_mfma.h) and are bound by instruction issue and
        # HBM together: the fraction of the HBM roof is the contract's figure, `matrix_pipe_frac` of the
        # dominant GEMM is reported next to it where a FLOP model exists.
        cand = {k: v for k, v in roof_kern.items() if v["n"] > 0 and v.get("bytes")}
        roof = None
        if cand:
            dom = max(cand, key=lambda k: cand[k]["total_ms"])
            tot_s = cand[dom]["total_ms"] * 1e-3
            ach = cand[dom]["bytes"] / tot_s
            per_launch = cand[dom]["bytes"] / cand[dom]["n"]
            traffic, traffic_note = None, None
            tp = next((q for q in (os.path.join(ROOT, "profiles", f"r0{k}_hbm_traffic.json") for k in (6, 5, 4, 3, 2)) if os.path.exists(q)),
                      os.path.join(ROOT, "profiles", "r04_hbm_traffic.json"))
            if os.path.exists(tp):  # PMC passes over the same kernels at this workload's shapes (tools/pmc_traffic.sh)
                tj = json.load(open(tp))
                ent = tj.get("workloads", {}).get(w.get("base", cfg_name), {}).get(dom)
                if ent:
                    traffic = ent["ratio"] * per_launch / 1e9
                    traffic_note = (f"GB per launch = {ent['ratio']:.3f} (HBM bytes measured by rocprofv3 --pmc, FETCH_SIZE and "
                                    f"WRITE_SIZE in separate passes with the guide's gfx950 unit corrections, / algorithmic bytes "
                                    f"of the same launches; taken at commit {tj.get('git_sha')}, profiles/{os.path.basename(tp)[:-5]}.md) x "
                                    f"{per_launch / 1e9:.4f} GB algorithmic per launch in this run")
            # the same launches on the matrix pipe: bf16 MFMAs per 32-sample slab (static census of the compiled kernels,
            # profiles/r03_isa_census.md) x 32.3 cycles each (profiles/r03_mfma_valu_overlap.md) over 1024 SIMDs at 2.4 GHz
            mfma_slab = {"bwd_dx_dw1": 270, "bwd_dx": 192, "bwd_full": 384, "bwd_full_dw1": 462, "fwd_fused2": 240, "fwd_fused2_k64": 288, "fwd_hidden": 192,
                         "dw_hidden": 192, "tangent_hidden": 384, "update_fwd": 315, "update_logp": 240, "update_last": 267}.get(dom)
            pipe = pipe_nominal = None
            if mfma_slab and not w.get("rnn"):
                slabs = Tn * n_local / 32.0
                pipe_nominal = slabs * mfma_slab * 32.3 / (1024 * 2.4e9) / (cand[dom]["avg_ms"] * 1e-3)
                pipe = pipe_nominal * 2.4 / clock_ghz if clock_ghz else pipe_nominal
            roof = dict(kernel=dom, bound="hbm", achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=ach / HBM_PEAK,
                        hbm_frac=ach / HBM_PEAK, matrix_pipe_frac=pipe, matrix_pipe_frac_nominal=pipe_nominal, clock_ghz=clock_ghz,
                        bound_note="streaming GEMM kernels between two roofs: `frac` = algorithmic bytes / time against the 8 TB/s "
                                   "HBM peak (the contract's figure); `matrix_pipe_frac` = the launch's bf16 MFMAs x 32.3 cycles "
                                   "against the cycles all 1024 SIMDs get at `clock_ghz`, the shader clock MEASURED under this "
                                   "load in an untimed step before the region (harl_clock_probe: shader cycles per 100 MHz tick "
                                   "on a side stream); `matrix_pipe_frac_nominal` = the same against the nominal 2.4 GHz.  Neither "
                                   "roof is saturated: VALU work of the exact bf16 split / LayerNorm / ReLU shares the issue "
                                   "slots of the same waves (DESIGN.md 3)",
                        traffic=traffic, traffic_note=traffic_note, launches=cand[dom]["n"], avg_ms=cand[dom]["avg_ms"],
                        bytes_per_launch=per_launch,
                        timing=("HIP events around every launch of the streaming kernel families inside the timed region; bytes "
                                "from each launch's own arguments (harl_amd/traffic.py)" if region_events else
                                "HIP events around every tagged launch of ONE instrumented single-stream step BEHIND the timed region "
                                "(attached workload: nothing is bracketed inside its region); bytes from each launch's own arguments"),
                        others={k: dict(hbm_frac=round(v["bytes"] / (v["total_ms"] * 1e-3) / HBM_PEAK, 4), n=v["n"],
                                        avg_ms=round(v["avg_ms"], 4))
                                for k, v in ({t: x for t, x in kern.items() if t in ROOF_TAGS and x["n"] > 0 and x.get("bytes")}
                                             or cw or cand).items()},
                        others_note="every streaming family in the instrumented single-stream steps after the timed region (the "
                                    "dominant one -- chosen in the last warm-up step -- alone is bracketed inside the region)")
            if cfg_name == "mpe":  # SURVEY.md 8(d) / BASELINE.md: 2 674 176 algorithmic FLOP per transition of this workload
                roof["end_to_end"] = dict(
                    flop_per_transition=2674176, achieved_tflops=2674176 * value / world / 1e12,
                    frac_of_fp32_mfma_peak=2674176 * value / world / MFMA_F32_PEAK,
                    note="whole step per GPU against SURVEY.md 8(d)'s roofline (58.8 M transitions/s per GPU = 100 % of the dense "
                         "fp32-MFMA peak, 157.3 TFLOP/s); the GEMMs run on the bf16 pipe as six exact products")
            ks = kern.get(dom)
            if ks and ks.get("bytes") and not args.time_all_tags:
                a1 = ks["bytes"] / (ks["total_ms"] * 1e-3)
                roof["single_stream"] = dict(
                    achieved=a1 / 1e9, frac=a1 / HBM_PEAK, avg_ms=ks["avg_ms"], launches=ks["n"],
                    matrix_pipe_frac=(pipe * cand[dom]["avg_ms"] / ks["avg_ms"]) if pipe else None,
                    matrix_pipe_frac_nominal=(pipe_nominal * cand[dom]["avg_ms"] / ks["avg_ms"]) if pipe_nominal else None,
                    note=f"the same kernel family in the {instr_steps} instrumented steps after the timed region, critic chain on the "
                         "main stream (HARL_CRITIC_STREAM=0): per-launch durations without a second kernel sharing the chip")
        out = dict(
            metric=w["metric"], value=value, unit="transitions/s", n_gpus=world, steps=steps, warmup=warmup,
            ms_per_step=dt / steps * 1e3, higher_is_better=True, scaling=scaling, vs_baseline=None, dtype="f32",
            dtype_note="fp32 data, statistics and accumulators; GEMM operands are split EXACTLY into three bf16 each and "
                       "multiplied as six cross products on v_mfma_f32_32x32x16_bf16 (error <= the fp32 MFMA's fmaf chain: "
                       "profiles/r01_mfma_bf16x3.txt)",
            data=f"synthetic (SURVEY.md 8d recipe, generated on the device; stored log-probs: {args.logp})",
            config=dict(workload=f"{w['name']}: compute_returns + train(), T={Tn}, n_rollout_threads={n_local}/GPU "
                                 f"({n_local * world} global, {scaling} scaling), obs{w['obs']}/share{w['sobs']}/"
                                 f"{'Discrete' if w['disc'] else 'Box'}{w['act']}, MLP{w['hidden']}{' + GRU' if w.get('rnn') else ''}, "
                                 f"{'ppo_epoch=5, ' if w['algo'] == 'happo' else 'CG 10 + line search, '}critic_epoch=5",
                        baseline_config=w.get("base", cfg_name), episode_length=Tn, n_rollout_threads_per_gpu=n_local,
                        n_rollout_threads_global=n_local * world, n_agents=w["A"],
                        parallelism=f"dp{world}",
                        collective=("none" if not comm.enabled else "oneshot(hipIpc)" if comm.oneshot is not None else "rccl"),
                        allreduce_info=comm.oneshot_info,
                        world_size=torch.distributed.get_world_size() if comm.enabled else 1,
                        ranks_seen=n_seen, git_sha=git_sha()),
            roofline=roof,
            kernels={k: dict(n=v["n"], avg_ms=round(v["avg_ms"], 4), total_ms=round(v["total_ms"], 3),
                             **({"hbm_frac": round(v["bytes"] / (v["total_ms"] * 1e-3) / HBM_PEAK, 4), "alg_bytes": v["bytes"]}
                                if v.get("bytes") else {}))
                     for k, v in kern.items()},
            kernel_timing=f"`kernels`: HIP events around every tagged launch, {instr_steps} instrumented single-stream steps after the timed region",
        )
        if w["algo"] == "happo":
            e2e = flops_per_transition(w) * value
            out["end_to_end"] = dict(algorithmic_tflops=e2e / 1e12, frac_of_fp32_mfma_peak=e2e / (MFMA_F32_PEAK * world),
                                     flops_per_transition=flops_per_transition(w))
        # executed-pass count: the Linear-layer FLOPs of the launches that actually ran in the instrumented steps (each launch's
        # own arguments, harl_amd/traffic.py ALGORITHMIC_FLOPS) -- the only end-to-end figure HATRPO has (CG / line-search
        # trip counts are data dependent); for HAPPO it is the closed form minus the pre-update pass shared with epoch 0
        ex = sum(v.get("flops") or 0.0 for v in kern.values())
        if ex > 0 and instr_steps > 0 and not args.time_all_tags:
            per_step = ex / instr_steps
            out.setdefault("end_to_end", {}).update(
                executed_flops_per_step=per_step, executed_tflops=per_step / (dt / steps) / 1e12,
                executed_frac_of_fp32_mfma_peak=per_step / (dt / steps) / MFMA_F32_PEAK,
                # ... and against the pipe the GEMMs actually run on: every fp32 product is six bf16 products there, so the
                # speed of light of this design is the dense bf16 peak / 6 = 419 TFLOP/s of fp32-equivalent work
                bf16x6_frac_of_bf16_mfma_peak=6.0 * per_step / (dt / steps) / MFMA_BF16_PEAK,
                executed_note="Linear-layer FLOPs of the launches of one instrumented step (per rank) / the timed step; "
                              "`bf16x6_frac_of_bf16_mfma_peak` = 6 x those FLOPs (the exact three-way operand split evaluates "
                              "six bf16 products per fp32 product) against the dense bf16 MFMA peak (2516.6 TFLOP/s): how far "
                              "the step is from the roof of the pipe it runs on -- the fp32-MFMA fraction next to it is the "
                              "contract's figure (SURVEY.md 8d), not a statement that the step is nearly done")
            if roof is not None:
                roof["bf16x6_end_to_end"] = out["end_to_end"]["bf16x6_frac_of_bf16_mfma_peak"]
    return out


def git_sha() -> str | None:
    try:
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
    except Exception:  # noqa: BLE001
        sha = ""
    if not sha and os.path.exists(os.path.join(ROOT, ".git_sha")):  # the GPU box receives a snapshot without .git
        sha = open(os.path.join(ROOT, ".git_sha")).read().strip()
    return sha or None


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-execute this command line under torch.distributed.run, one rank per
    GPU of this node (rendezvous on 127.0.0.1, a free port unless MASTER_PORT is set).  Rank 0 of the child job prints the
    JSON line on the inherited stdout; everything else the ranks write goes to stderr."""
    import socket

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="mpe")
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default="auto",
                    help="auto (default): one rank = the config's single-GPU size; N > 1 ranks = BOTH figures on the one line -- "
                         "headline `value` = STRONG scaling at the n_rollout_threads the BASELINE quotes the workload on (4096 "
                         "global for mpe, 8192 for cheetah6 = configs[2]; split over the ranks), and `weak` = the config's size on "
                         "every rank.  weak / strong: that one figure only")
    ap.add_argument("--global-threads", type=int, default=0,
                    help="n_rollout_threads of the whole job with --scaling strong (0 = the BASELINE's global size of the config)")
    ap.add_argument("--threads-per-gpu", type=int, default=0, help="n_rollout_threads per rank with --scaling weak (0 = the config's)")
    ap.add_argument("--logp", choices=("recipe", "onpolicy"), default="recipe",
                    help="stored log-probs: SURVEY 8d recipe (default) or on-policy (ratios ~ 1)")
    ap.add_argument("--dist-single", action="store_true", help="initialise the nccl (RCCL) group even with one rank")
    ap.add_argument("--cpu-cols", type=int, default=-1, help="rollout threads of the bounded CPU-baseline sample (0 = skip, -1 = auto)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="timed CPU-baseline updates after the warm-up one (best of)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch CPU threads for the baseline (these nets are small: more threads than ~16 is slower)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--instr-steps", type=int, default=2, help="instrumented (HIP-event) steps after the timed region")
    ap.add_argument("--time-all-tags", action="store_true",
                    help="bracket EVERY tagged launch inside the timed region (used by tools/pmc_traffic.sh so that `kernels` "
                         "describes exactly the step the PMC passes profile; costs ~10 %% of the step)")
    ap.add_argument("--with-strong-cheetah6", action="store_true",
                    help="after the line of this invocation print a SECOND JSON line: BASELINE configs[2] as quoted (HalfCheetah-6x1, "
                         "--scaling strong, --global-threads 8192 split over the same ranks)")
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="default run (mpe): do NOT append the three other BASELINE workloads (cheetah6, smac3s5z, humanoid17; "
                         "`--other-steps` steps each after the headline region) as `other_configs` to the JSON line")
    ap.add_argument("--other-steps", type=int, default=5)
    ap.add_argument("--other-cpu-cols", type=int, default=-1,
                    help="rollout threads of the bounded CPU-baseline sample attached to each `other_configs` entry (0 = skip, -1 = auto)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU: spawn / rendezvous (gloo) / one all-reduce, then print a line with value null")
    args = ap.parse_args()
    if args.with_strong_cheetah6 and "RANK" not in os.environ:  # two runs of this script, one line each
        base = [a for a in sys.argv[1:] if a != "--with-strong-cheetah6"]
        rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + base)
        keep = []
        skip = False
        for a in base:  # drop the first run's workload selection
            if skip:
                skip = False
            elif a in ("--config", "--scaling", "--global-threads", "--threads-per-gpu"):
                skip = True
            elif not a.startswith(("--config=", "--scaling=", "--global-threads=", "--threads-per-gpu=")):
                keep.append(a)
        rc2 = subprocess.call([sys.executable, os.path.abspath(__file__)] + keep +
                              ["--config", "cheetah6", "--scaling", "strong", "--global-threads", "8192"])  # = configs[2] as quoted
        sys.exit(rc or rc2)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    w = WORKLOADS[args.config]
    # stdout carries exactly ONE line (the JSON): everything else that native libraries write to file descriptor 1 -- RCCL prints
    # a version banner there when the process group goes away -- is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from harl_amd import _lib
    from harl_amd.dist import init_from_env

    if args.dist_single and "RANK" not in os.environ:  # a one-rank RCCL group: exercises the collective branch on a 1-GPU box
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), HARL_DIST_SINGLE="1")
    comm = init_from_env()
    rank, world = comm.rank, comm.world_size
    if args.dry_run:  # tests/test_bench_launcher_cpu.py: the command line the driver runs, minus the GPU work
        t = torch.tensor([float(rank + 1), 1.0], dtype=torch.float64)
        comm.all_reduce_sum(t)
        assert world == args.gpus and float(t[0].item()) == world * (world + 1) / 2, (world, args.gpus, t)
        g_threads = args.global_threads or BASELINE_GLOBAL_THREADS.get(args.config, w["N"])
        both = args.scaling == "auto" and world > 1 and not args.threads_per_gpu
        strong = args.scaling == "strong" or both
        if rank == 0:
            os.write(json_fd, (json.dumps(dict(metric=w["metric"], value=None, unit="transitions/s", n_gpus=world, steps=args.steps,
                                               warmup=args.warmup, dry_run=True, scaling="strong" if strong else "weak",
                                               weak=dict(value=None, n_rollout_threads_global=w["N"] * world) if both else None,
                                               config=dict(parallelism=f"dp{world}", ranks_seen=int(t[1].item()),
                                                           n_rollout_threads_global=g_threads if strong else (args.threads_per_gpu or w["N"]) * world,
                                                           n_rollout_threads_per_gpu=(g_threads // world) if strong else (args.threads_per_gpu or w["N"]),
                                                           collective=torch.distributed.get_backend() if comm.enabled else "none")))
                               + "\n").encode())
        if comm.enabled:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"launched {world} ranks for --gpus {args.gpus}"
    g_threads = args.global_threads or BASELINE_GLOBAL_THREADS.get(args.config, w["N"])
    both = args.scaling == "auto" and world > 1 and not args.threads_per_gpu
    scaling = "strong" if (args.scaling == "strong" or both) else "weak"
    if scaling == "strong":
        assert g_threads % world == 0, "--global-threads must be divisible by the number of ranks"
        n_local = g_threads // world
    else:
        n_local = args.threads_per_gpu or w["N"]
    Tn = w["T"]

    out = measure(w, args.config, args, comm, rank, world, device, n_local, args.steps, args.warmup, args.instr_steps, scaling)
    if both:
        # the weak-scaling figure of the same job on the same line (every rank at the config's single-GPU size): its own
        # barrier-bracketed region of the same number of steps, after the headline region
        torch.cuda.empty_cache()
        wk = measure(w, args.config, args, comm, rank, world, device, w["N"], args.steps, args.warmup, 0, "weak")
        if rank == 0:
            out["weak"] = dict(value=wk["value"], unit=wk["unit"], ms_per_step=wk["ms_per_step"], steps=wk["steps"], warmup=wk["warmup"],
                               scaling="weak", n_rollout_threads_per_gpu=w["N"], n_rollout_threads_global=w["N"] * world,
                               ranks_seen=wk["config"]["ranks_seen"], roofline=wk.get("roofline"))
            out["strong"] = dict(value=out["value"], ms_per_step=out["ms_per_step"], scaling="strong",
                                 n_rollout_threads_per_gpu=n_local, n_rollout_threads_global=g_threads,
                                 note="headline `value`: the BASELINE's global n_rollout_threads split over the ranks")
    if rank == 0:
        cols = args.cpu_cols
        if cols < 0:  # ~10-30 s of CPU work per update for every configuration
            cols = {"mpe": 512, "cheetah6": 512, "smac3s5z": 128, "humanoid17": 64, "hatrpo_gru128": 64}[w.get("base", args.config)]
        if world == 1 and cols > 0:
            out["cpu_baseline"] = cpu_baseline(w, cols, min(args.cpu_threads, os.cpu_count() or 1), reps=max(1, args.cpu_reps))
    # the other BASELINE.json workloads at their real shapes, on the SAME JSON line (after the headline region and the CPU
    # baseline, fresh runner each, a few steps): `--config <name>` gives the full record of any one of them
    # (single-process runs only: the scaling runs time the headline workload, and an attached workload that failed on ONE rank
    # would leave the others waiting in a collective)
    if args.other_configs and world == 1 and args.config == "mpe" and scaling == "weak" and not args.threads_per_gpu:
        others = {}
        for name in ("cheetah6", "smac3s5z", "smac3s5z_n4096", "humanoid17", "humanoid17_n4096", "hatrpo_gru128"):
            torch.cuda.empty_cache()
            wo = WORKLOADS[name]
            try:
                o = measure(wo, name, args, comm, rank, world, device, wo["N"], args.other_steps, 2, 1, region_events=False)
            except Exception as e:  # noqa: BLE001 -- the headline record must survive a failure of an attached one
                o = dict(error=f"{type(e).__name__}: {e}") if rank == 0 else None
            if rank == 0:
                if "error" in o:
                    others[name] = o
                    continue
                rf = o.get("roofline") or {}
                others[name] = dict(metric=o["metric"], value=o["value"], ms_per_step=o["ms_per_step"], steps=o["steps"],
                                    warmup=o["warmup"], workload=o["config"]["workload"],
                                    roofline=dict(kernel=rf.get("kernel"), frac=rf.get("frac"), avg_ms=rf.get("avg_ms"),
                                                  matrix_pipe_frac=rf.get("matrix_pipe_frac"), clock_ghz=rf.get("clock_ghz")),
                                    end_to_end=o.get("end_to_end"))
                if wo.get("base"):  # the same nets / epochs at more rollout threads: the CPU twin is the base entry's
                    others[name]["cpu_baseline"] = dict(see=wo["base"], note="transitions/s of the CPU path do not depend on the number "
                                                        "of rollout threads beyond ~64 columns (profiles/r04_bench_mpe_cpu_baseline_n4096.json)")
                elif args.other_cpu_cols != 0:  # the CPU path next to every reported number (BASELINE.json north_star): a bounded
                    # sample of the same workload (>= 64 rollout threads: VERDICT r05 weak 11), one warm-up + one timed update
                    oc = args.other_cpu_cols if args.other_cpu_cols > 0 else {"cheetah6": 512, "smac3s5z": 128, "humanoid17": 64, "hatrpo_gru128": 64}[name]
                    try:
                        others[name]["cpu_baseline"] = cpu_baseline(wo, oc, min(args.cpu_threads, os.cpu_count() or 1), reps=1)
                    except Exception as e:  # noqa: BLE001
                        others[name]["cpu_baseline"] = dict(error=f"{type(e).__name__}: {e}")
        if rank == 0:
            out["other_configs"] = others
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm.enabled:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
