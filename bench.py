#!/usr/bin/env python3
"""bench.py -- transitions/sec through the HAPPO update (compute_returns + OnPolicyHARunner.train()).

Workload = BASELINE.json configs[1]: MPE simple_spread_v2, 3 agents, HAPPO, n_rollout_threads=4096 per GPU,
episode_length=200, obs 18 / share_obs 54 / Box(5), MLP [128,128], happo.yaml defaults (ppo_epoch=5,
critic_epoch=5, 1 mini-batch, ValueNorm + GAE + proper time limits, Huber, clip 0.2, max_grad_norm 10).
A "step" = one compute() + train() over the synthetic rollout buffers, which are resident in HBM before
the timed region.  One transition = one (t, n) environment step for all agents.  Weak scaling: every rank owns
4096 rollout threads; gradients / loss scalars are all-reduced over RCCL each optimiser step.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 5 --warmup 2

Prints ONE JSON line (rank 0).  `roofline` is for the dominant GEMM kernel family (HIP-event timing of every launch
of the GEMM families inside the timed region; HBM-bound since the GEMMs moved to the bf16 matrix pipe with an exact
fp32 operand split; `traffic` from the committed PMC pass in profiles/); `kernels` is the
full per-kernel breakdown from extra instrumented steps after the timed region; `cpu_baseline` is the oracle (a torch-CPU restatement of the reference,
same ATen kernels) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_F32_PEAK = 157.3e12  # MI355X_MICROARCH.md: fp32-input MFMA, dense
MFMA_BF16_PEAK = 2516.6e12  # 16 x the fp32 rate (dense bf16 MFMA, 1024 FLOP/clk/SIMD at 2.4 GHz)
HBM_PEAK = 8.0e12

T, N_PER_GPU, A = 200, 4096, 3
OBS, SOBS, ACT = 18, 54, 5
HIDDEN = [128, 128]


def algo_args(n_threads: int, T_: int = T) -> dict:
    return dict(
        train=dict(n_rollout_threads=n_threads, episode_length=T_, use_valuenorm=True, use_proper_time_limits=True),
        model=dict(hidden_sizes=HIDDEN, activation_func="relu", use_feature_normalization=True,
                   initialization_method="orthogonal_", gain=0.01, use_naive_recurrent_policy=False,
                   use_recurrent_policy=False, recurrent_n=1, data_chunk_length=10, lr=5e-4, critic_lr=5e-4,
                   opti_eps=1e-5, weight_decay=0, std_x_coef=1, std_y_coef=0.5),
        algo=dict(ppo_epoch=5, critic_epoch=5, use_clipped_value_loss=True, clip_param=0.2, actor_num_mini_batch=1,
                  critic_num_mini_batch=1, entropy_coef=0.01, value_loss_coef=1, use_max_grad_norm=True,
                  max_grad_norm=10.0, use_gae=True, gamma=0.99, gae_lambda=0.95, use_huber_loss=True,
                  use_policy_active_masks=True, huber_delta=10.0, action_aggregation="prod", share_param=False,
                  fixed_order=True),
    )


class Box:
    def __init__(self, shape):
        self.shape = shape


def flops_per_transition() -> float:
    """Linear FLOPs only. (2 + 3*ppo_epoch) F_actor per agent + 3*critic_epoch F_critic."""
    f_actor = 2 * (OBS * 128 + 128 * 128 + 128 * ACT)
    f_critic = 2 * (SOBS * 128 + 128 * 128 + 128 * 1)
    return A * (2 + 3 * 5) * f_actor + 3 * 5 * f_critic


def build_gpu_runner(n_local: int, rank: int, world: int, device):
    from harl_amd.runner import OnPolicyHARunner
    from harl_amd.synthetic import Shapes, make_buffers

    args = algo_args(n_local * world)
    torch.manual_seed(1)
    np.random.seed(1)
    r = OnPolicyHARunner(dict(algo="happo"), args, dict(state_type="EP"), obs_spaces=[Box((OBS,))] * A,
                         share_obs_space=Box((SOBS,)), act_spaces=[Box((ACT,))] * A, device=device)
    sh = Shapes(T=T, N=n_local, A=A, obs_dim=OBS, share_obs_dim=SOBS, act_dim=ACT, hidden_sizes=HIDDEN)
    d = make_buffers(sh, seed=100 + rank)
    up = lambda x: torch.from_numpy(x).to(device)  # noqa: E731
    for a in range(A):
        b = r.actor_buffer[a]
        b.obs.copy_(up(d.obs[a]))
        b.actions.copy_(up(d.actions[a]))
        b.masks.copy_(up(d.masks[a]))
        b.active_masks.copy_(up(d.active_masks[a]))
        # stored log-probs on-policy (ratio ~ 1, the regime PPO operates in): log pi(a|o) + 0.05 N(0,1)
        lp, _, _ = r.actor[a].evaluate_actions(b.flat("obs"), None, b.flat("actions"), None)
        noise = torch.from_numpy((0.05 * np.random.default_rng(7 + a).standard_normal(lp.shape)).astype(np.float32)).to(device)
        b.action_log_probs.copy_((lp + noise).reshape(b.action_log_probs.shape))
    cb = r.critic_buffer
    cb.share_obs.copy_(up(d.share_obs))
    cb.rewards.copy_(up(d.rewards))
    cb.value_preds.copy_(up(d.value_preds))
    cb.masks.copy_(up(d.critic_masks))
    cb.bad_masks.copy_(up(d.bad_masks))
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


def cpu_baseline(n_cols: int, threads: int) -> dict:
    """The oracle (torch-CPU restatement of the reference path, same ATen kernels / autograd / Adam) on a bounded
    sample of the same workload: same shapes with fewer rollout threads.  Reported, not a target."""
    from harl_amd.synthetic import Shapes, actor_param_shapes, critic_param_shapes, make_buffers, synthetic_state_dict
    from oracle import harl_oracle as O

    torch.set_num_threads(threads)
    args = algo_args(n_cols)
    cfg = O.PathConfig.from_reference_dicts(args["train"], args["model"], args["algo"])
    sh = Shapes(T=T, N=n_cols, A=A, obs_dim=OBS, share_obs_dim=SOBS, act_dim=ACT, hidden_sizes=HIDDEN)
    d = make_buffers(sh, seed=100)
    actors = [O.OracleHAPPO({k: torch.from_numpy(v) for k, v in synthetic_state_dict(actor_param_shapes(sh), 10 + a).items()}, cfg)
              for a in range(A)]
    critic = O.OracleVCritic({k: torch.from_numpy(v) for k, v in synthetic_state_dict(critic_param_shapes(sh), 99).items()}, cfg)
    abufs = []
    for a in range(A):
        with torch.no_grad():
            lp, _, _ = actors[a].evaluate_actions(d.obs[a][:-1].reshape(T * n_cols, -1), d.actions[a].reshape(T * n_cols, -1))
        logp = (lp.numpy() + 0.05 * np.random.default_rng(7 + a).standard_normal(lp.shape)).astype(np.float32)
        abufs.append(O.OracleActorBuffer(d.obs[a], d.actions[a], logp.reshape(d.actions[a].shape), d.masks[a], d.active_masks[a]))
    cbuf = O.OracleCriticBufferEP(d.share_obs, d.rewards, d.value_preds, d.critic_masks, d.bad_masks)
    vn = O.OracleValueNorm()
    t0 = time.perf_counter()
    with torch.no_grad():
        nv = critic.get_values(cbuf.share_obs[-1]).numpy()
    cbuf.compute_returns(nv, vn, cfg)
    O.ha_train(actors, critic, abufs, cbuf, vn, cfg)
    dt = time.perf_counter() - t0
    return dict(value=T * n_cols / dt, unit="transitions/s", cores=threads, kind="port",
                sample=f"1 update at T={T}, n_rollout_threads={n_cols} (same nets/epochs; oracle = torch-CPU restatement "
                       f"of the reference, {dt:.1f} s)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--threads-per-gpu", type=int, default=N_PER_GPU)
    ap.add_argument("--cpu-cols", type=int, default=512, help="rollout threads of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch CPU threads for the baseline (these nets are small: more threads than ~16 is slower)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--instr-steps", type=int, default=2, help="instrumented (HIP-event) steps after the timed region")
    args = ap.parse_args()

    from harl_amd import _lib
    from harl_amd.dist import init_from_env

    comm = init_from_env()
    rank, world = comm.rank, comm.world_size
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"launched {world} ranks for --gpus {args.gpus}"

    r = build_gpu_runner(args.threads_per_gpu, rank, world, device)

    def barrier():
        torch.cuda.synchronize()
        if comm.enabled:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(r)
    barrier()
    # Roofline timing lives INSIDE the timed region: every launch of the four MFMA kernel families is bracketed by HIP
    # events on the launch stream (64 event pairs per step, <1 % of wall time).
    MFMA_TAGS = ("fwd_fused2", "fwd_hidden", "bwd_dx", "bwd_dx_dw1", "dw_hidden")
    if not args.no_kernel_timing:
        _lib.enable_kernel_timing(True, MFMA_TAGS)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(r)
    barrier()
    dt = time.perf_counter() - t0
    mfma_kern = {}
    if not args.no_kernel_timing:
        mfma_kern = _lib.collect_kernel_timing()
        _lib.enable_kernel_timing(False)
    if comm.enabled:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    # Full per-kernel breakdown (every tagged launch, ~270 event pairs per step ~ 10 % of wall time): `instr_steps`
    # further steps of the SAME workload after the timed region, so that it does not understate `value`.
    kern = {}
    if not args.no_kernel_timing and args.instr_steps > 0:
        _lib.enable_kernel_timing(True)
        for _ in range(args.instr_steps):
            one_step(r)
        kern = _lib.collect_kernel_timing()
        _lib.enable_kernel_timing(False)
    breakdown = kern

    if rank == 0:
        n_local = args.threads_per_gpu
        trans_per_step = T * n_local * world
        value = trans_per_step * args.steps / dt
        B = T * n_local
        # dominant kernel = the GEMM kernel family with the largest total time inside the timed region.  Since the GEMMs
        # run on the bf16 matrix pipe (exact three-way fp32 split, 6 products: csrc/split_mfma.h) these kernels are
        # HBM-bound: `achieved` = ALGORITHMIC bytes per launch / average HIP-event duration.  Reported
        # next to it: the fp32-equivalent FLOP rate (Linear layers only) against the fp32 MFMA peak it no longer runs on,
        # and the matrix-pipe time actually issued (6 x bf16 GEMM FLOPs + 16 x the FLOPs left on the fp32 MFMA).
        flops = dict(fwd_hidden=2.0 * B * 128 * 128, bwd_dx=2.0 * B * 128 * 128, dw_hidden=2.0 * B * 128 * 128,
                     fwd_fused2=2.0 * B * (128 * 128 + OBS * 128),
                     # dX of layer 2 + the fused first-layer weight gradient (15 actor launches with D=18, 5 critic with 54)
                     bwd_dx_dw1=2.0 * B * (128 * 128 + 128 * (15 * OBS + 5 * SOBS) / 20.0))
        gemm_bf16 = 2.0 * B * 128 * 128  # the 128 x 128 GEMM of every family runs as 6 bf16 products
        pipe = {k: 6.0 * gemm_bf16 + 16.0 * (v - gemm_bf16) for k, v in flops.items()}  # bf16-pipe-equivalent FLOPs issued
        # ALGORITHMIC HBM bytes per launch of each family for THIS run's launch mix: the fused forward is
        # launched 15x per step in training mode (reads the cached x0n image, writes x_hat_1, x_hat_2, masks, statistics)
        # and 3x in log-prob mode (x_hat_2 only).
        alg = dict(fwd_hidden=B * (512 + 512 + 16 + 4), bwd_dx=B * (512 + 512 + 16 + 4 + 512), dw_hidden=B * (512 + 512),
                   bwd_dx_dw1=B * (512 + 512 + 16 + 4 + (15 * 128 + 5 * 256) / 20.0),
                   # fused forward from the cached x0n image: 128 B in; training mode writes x_hat_1, x_hat_2, masks, rstd
                   fwd_fused2=B * (15 * (128 + 512 + 512 + 32 + 8) + 3 * (128 + 512 + 16 + 4)) / 18.0)
        cand = {k: v for k, v in mfma_kern.items() if k in flops and v["n"] > 0}
        roof = None
        if cand:
            dom = max(cand, key=lambda k: cand[k]["total_ms"])
            avg_s = cand[dom]["avg_ms"] * 1e-3
            traffic, traffic_note = None, None
            tp = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
            if os.path.exists(tp):  # PMC pass over the same kernels (tools/kbench.py); measured/algorithmic ratio per family
                tj = json.load(open(tp))
                if dom in tj["kernels"]:
                    ratio = tj["kernels"][dom]["ratio"]
                    traffic = ratio * alg[dom] / 1e9
                    traffic_note = (f"GB per launch = {ratio:.3f} (HBM bytes measured by rocprofv3 --pmc FETCH_SIZE x2 [gfx950] + "
                                    f"WRITE_SIZE, separate passes, / algorithmic bytes of the same kernel; "
                                    f"profiles/r01_hbm_traffic.md) x {alg[dom] / 1e9:.3f} GB algorithmic for this launch mix")
            ach = alg[dom] / avg_s
            roof = dict(kernel=dom, bound="hbm", achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                        frac=ach / HBM_PEAK, traffic=traffic, traffic_note=traffic_note, launches=cand[dom]["n"],
                        avg_ms=cand[dom]["avg_ms"], bytes_per_launch=alg[dom], flops_per_launch=flops[dom],
                        fp32_equiv_tflops=flops[dom] / avg_s / 1e12,
                        frac_of_fp32_mfma_peak=flops[dom] / avg_s / MFMA_F32_PEAK,
                        matrix_pipe_frac=pipe[dom] / avg_s / MFMA_BF16_PEAK,
                        timing="HIP events around every launch of the GEMM kernel families inside the timed region",
                        others={k: dict(hbm_frac=round(alg[k] / (v["avg_ms"] * 1e-3) / HBM_PEAK, 4),
                                        fp32_equiv_frac=round(flops[k] / (v["avg_ms"] * 1e-3) / MFMA_F32_PEAK, 4),
                                        matrix_pipe_frac=round(pipe[k] / (v["avg_ms"] * 1e-3) / MFMA_BF16_PEAK, 4))
                                for k, v in cand.items()})
        e2e = flops_per_transition() * value
        out = dict(
            metric="transitions/sec through HAPPO update (MPE spread, 3 agents)", value=value, unit="transitions/s",
            n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
            higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
            dtype_note="fp32 data, statistics and accumulators; the 128x128 GEMM operands are split EXACTLY into three bf16 "
                       "each and multiplied as six cross products on v_mfma_f32_32x32x16_bf16 (error <= the fp32 MFMA's fmaf "
                       "chain: profiles/r01_mfma_bf16x3.txt)",
            data="synthetic (SURVEY.md 8d recipe; stored log-probs set on-policy so ratios ~ 1)",
            config=dict(workload="MPE simple_spread_v2 3-agent HAPPO update: compute_returns + train(), T=200, "
                                 f"n_rollout_threads={n_local}/GPU, obs18/share54/Box5, MLP[128,128], ppo_epoch=5, critic_epoch=5",
                        episode_length=T, n_rollout_threads_per_gpu=n_local, n_agents=A, parallelism=f"dp{world}"),
            roofline=roof,
            end_to_end=dict(algorithmic_tflops=e2e / 1e12, frac_of_mfma_peak=e2e / (MFMA_F32_PEAK * world),
                            flops_per_transition=flops_per_transition()),
            kernels={k: dict(n=v["n"], avg_ms=round(v["avg_ms"], 4), total_ms=round(v["total_ms"], 3))
                     for k, v in breakdown.items()},
            kernel_timing=f"`kernels`: HIP events around every tagged launch, {args.instr_steps} instrumented steps after the timed region",
        )
        if world == 1 and args.cpu_cols > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_cols, min(args.cpu_threads, os.cpu_count() or 1))
        print(json.dumps(out), flush=True)
    if comm.enabled:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
