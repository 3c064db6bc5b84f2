#!/usr/bin/env python3
"""Per-kernel micro-benchmark through the C ABI at the bench shape (B = 200*4096 samples, H = 128).

    python tools/kbench.py [--reps 20] [--lib path/to/libharl_hip.so] [filter ...]

Prints ms per launch and the algorithmic TFLOP/s or TB/s of each kernel (HIP events on the launch stream)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--B", type=int, default=200 * 4096)
    ap.add_argument("filters", nargs="*")
    args = ap.parse_args()
    from harl_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from harl_amd._lib import call, ptr, stream
    dev = torch.device("cuda:0")
    B, H = args.B, 128
    ns = (B + 31) // 32
    mp = ns * 32
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    obs, sobs = rn(B, 18), rn(B, 54)
    xh1, xh2, dz, dz2 = rn(mp * H), rn(mp * H), rn(mp * H), torch.empty(mp * H, device=dev)
    mask = torch.randint(-2**31, 2**31 - 1, (ns * 2 * 64,), device=dev, dtype=torch.int32, generator=g)
    rstd = torch.rand(mp, device=dev, generator=g) + 0.5
    mu0, rstd0 = torch.zeros(mp, device=dev), torch.ones(mp, device=dev)
    W = rn(H * H) * 0.1
    b = rn(H) * 0.1
    W1, W1c = rn(H * 18) * 0.2, rn(H * 54) * 0.2
    Wh, bh, ls = rn(5 * H) * 0.05, rn(5) * 0.1, torch.ones(5, device=dev)
    Wv, bv = rn(H) * 0.05, rn(1)
    actions, old_logp = rn(B, 5), rn(B, 5) * 0.1 - 1
    adv, factor, active = rn(B), torch.ones(B, device=dev), torch.ones(B, device=dev)
    dhead = rn(mp * 32) * 0.01
    n_wg = 512
    part = torch.empty(n_wg * (H * H + H), device=dev)
    dwp = torch.empty(H * H + H, device=dev)
    nb = _lib.load().harl_head_blocks(B)
    ps = torch.zeros(nb * 48, device=dev)
    logp_out = torch.empty(B, 5, device=dev)
    vals = torch.empty(B, device=dev)
    vp, ret = rn(B), rn(B)
    vn = torch.tensor([0.1, 1.2, 0.9], device=dev)
    T, N = 200, B // 200
    rew, vpT, mk = rn(T, N), rn(T + 1, N), (torch.rand(T + 1, N, device=dev) > 0.04).float()
    rets, advs = torch.empty(T + 1, N, device=dev), torch.empty(T, N, device=dev)
    P = 20142
    pp, gg, mm, vv = rn(P), rn(P), torch.zeros(P, device=dev), torch.zeros(P, device=dev)
    x0n32 = rn(mp * 32)
    x0n64 = rn(mp * 64)
    wimg = torch.empty(3 * H * 64 // 2, device=dev)
    ps_f = torch.zeros(n_wg * 48, device=dev)
    part_h = torch.empty(n_wg * (32 * H + 32), device=dev)
    part_1 = torch.empty(n_wg * (H * 32 + H), device=dev)
    s = stream()
    GF = lambda f: ("TFLOP/s", f / 1e12)  # noqa: E731
    GB = lambda f: ("TB/s", f / 1e12)  # noqa: E731
    fl = 2.0 * B * H * H
    jobs = [
        ("fwd_input_D18", lambda: call("harl_mlp_fwd_input", ptr(obs), 18, None, B, 18, ptr(W1), ptr(b), 1, H, ptr(xh1), ptr(mask), ptr(rstd), ptr(mu0), ptr(rstd0), None, s), GB(B * (72 + 512 + 20))),
        ("fwd_input_D54", lambda: call("harl_mlp_fwd_input", ptr(sobs), 54, None, B, 54, ptr(W1c), ptr(b), 1, H, ptr(xh1), ptr(mask), ptr(rstd), ptr(mu0), ptr(rstd0), None, s), GB(B * (216 + 512 + 20))),
        ("fwd_fused2_D18_train", lambda: call("harl_mlp_fwd_fused2", ptr(obs), 18, None, B, 18, ptr(W1), ptr(b), 1, ptr(W), ptr(b), H, 1, ptr(xh1), ptr(mask), ptr(rstd), ptr(mu0), ptr(rstd0), ptr(xh2), ptr(mask), ptr(rstd), None, s), GF(fl + 2.0 * B * 18 * H)),
        ("fwd_fused2_D18_logp", lambda: call("harl_mlp_fwd_fused2", ptr(obs), 18, None, B, 18, ptr(W1), ptr(b), 1, ptr(W), ptr(b), H, 0, ptr(xh1), ptr(mask), ptr(rstd), ptr(mu0), ptr(rstd0), ptr(xh2), ptr(mask), ptr(rstd), None, s), GF(fl + 2.0 * B * 18 * H)),
        ("x0n_D18", lambda: call("harl_mlp_x0n_wide", ptr(obs), 18, None, B, 18, 1, ptr(x0n32), ptr(mu0), ptr(rstd0), s), GB(B * (72 + 128 + 8))),
        ("x0n_D54", lambda: call("harl_mlp_x0n_wide", ptr(sobs), 54, None, B, 54, 1, ptr(x0n64), ptr(mu0), ptr(rstd0), s), GB(B * (216 + 256 + 8))),
        ("fwd_fused2x_train", lambda: call("harl_mlp_fwd_fused2x", ptr(x0n32), B, ptr(W1), 18, ptr(b), ptr(W), ptr(b), H, 1, ptr(xh1), ptr(mask), ptr(rstd), ptr(xh2), ptr(mask), ptr(rstd), s), GF(fl + 2.0 * B * 18 * H)),
        ("fwd_fused2x_logp", lambda: call("harl_mlp_fwd_fused2x", ptr(x0n32), B, ptr(W1), 18, ptr(b), ptr(W), ptr(b), H, 0, ptr(xh1), ptr(mask), ptr(rstd), ptr(xh2), ptr(mask), ptr(rstd), s), GF(fl + 2.0 * B * 18 * H)),
        ("fwd_wide_K64", lambda: call("harl_mlp_fwd_wide", ptr(x0n64), B, 64, ptr(W1c), 54, ptr(b), H, ptr(wimg), ptr(xh1), ptr(mask), ptr(rstd), s), GB(B * (256 + 512 + 20))),
        ("fwd_hidden", lambda: call("harl_mlp_fwd_hidden", ptr(xh1), B, H, H, ptr(W), ptr(b), ptr(xh2), ptr(mask), ptr(rstd), s), GF(fl)),
        ("bwd_dx", lambda: call("harl_mlp_bwd_dx", ptr(dz), ptr(xh1), ptr(mask), ptr(rstd), B, H, H, ptr(W), ptr(dz2), None, 0, None, 0, s), GF(fl)),
        ("bwd_full_fill", lambda: call("harl_mlp_bwd_dx_dw", ptr(dz), ptr(xh1), ptr(mask), ptr(rstd), B, H, H, ptr(W), ptr(dz2), None, 0, None, ptr(part), n_wg, 1, s), GF(2 * fl)),
        ("bwd_full_nofill", lambda: call("harl_mlp_bwd_dx_dw", ptr(dz), ptr(xh1), ptr(mask), ptr(rstd), B, H, H, ptr(W), ptr(dz2), None, 0, None, ptr(part), n_wg, 0, s), GF(2 * fl)),
        ("bwd_full_dw1_fill", lambda: call("harl_mlp_bwd_dx_dw", ptr(dz), ptr(xh1), ptr(mask), ptr(rstd), B, H, H, ptr(W), None, ptr(x0n32), 32, ptr(part_1), ptr(part), n_wg, 1, s), GF(2 * fl + 2.0 * B * H * 18)),
        ("bwd_full_dw1_nofill", lambda: call("harl_mlp_bwd_dx_dw", ptr(dz), ptr(xh1), ptr(mask), ptr(rstd), B, H, H, ptr(W), None, ptr(x0n32), 32, ptr(part_1), ptr(part), n_wg, 0, s), GF(2 * fl + 2.0 * B * H * 18)),
        ("bwd_dx_dw1_fused", lambda: call("harl_mlp_bwd_dx", ptr(dz), ptr(xh1), ptr(mask), ptr(rstd), B, H, H, ptr(W), None, ptr(x0n32), 32, ptr(part_1), n_wg, s), GF(fl + 2.0 * B * H * 18)),
        ("dw_hidden", lambda: call("harl_mlp_dw_partials", ptr(dz), 0, 0, H, ptr(xh1), 0, 0, None, None, None, H, B, ptr(part), n_wg, s), GF(fl)),
        ("dw_head", lambda: call("harl_mlp_dw_partials", ptr(dhead), 1, 32, 5, ptr(xh2), 0, 0, None, None, None, H, B, ptr(part), n_wg, s), GB(B * (512 + 128))),
        ("dw_input_D18", lambda: call("harl_mlp_dw_partials", ptr(dz), 0, 0, H, ptr(obs), 1, 18, None, ptr(mu0), ptr(rstd0), 18, B, ptr(part), n_wg, s), GB(B * (512 + 72 + 8))),
        ("dw_input_atl32", lambda: call("harl_mlp_dw_partials", ptr(dz), 0, 0, H, ptr(xh1), 0, 0, None, None, None, 32, B, ptr(part), n_wg, s), GB(B * (512 + 128))),
        ("dw_input_atl64", lambda: call("harl_mlp_dw_partials", ptr(dz), 0, 0, H, ptr(xh1), 0, 0, None, None, None, 64, B, ptr(part), n_wg, s), GB(B * (512 + 256))),
        ("dw_input_D54", lambda: call("harl_mlp_dw_partials", ptr(dz), 0, 0, H, ptr(sobs), 1, 54, None, ptr(mu0), ptr(rstd0), 54, B, ptr(part), n_wg, s), GB(B * (512 + 216 + 8))),
        ("reduce_partials", lambda: call("harl_reduce_partials", ptr(part), n_wg, H * H + H, ptr(dwp), s), GB(n_wg * (H * H + H) * 4)),
        ("actor_head_logp", lambda: call("harl_actor_head_logp", ptr(xh2), B, H, ptr(Wh), ptr(bh), ptr(ls), 1.0, 0.5, 0, 5, ptr(actions), None, ptr(logp_out), None, None, 0, None, 0, 0, s), GB(B * (512 + 40))),
        ("actor_head_loss", lambda: call("harl_actor_head_loss", ptr(xh2), ptr(mask), ptr(rstd), B, H, ptr(Wh), ptr(bh), ptr(ls), 1.0, 0.5, 0, 5, None, ptr(actions), None, ptr(old_logp), ptr(adv), None, ptr(factor), ptr(active), 0.2, 0.01, 0, 0, 0, 0, None, ptr(dz2), ptr(dhead), ptr(ps), None, 0, s), GB(B * (512 + 512 + 128 + 60))),
        ("actor_head_loss_fused_dw", lambda: call("harl_actor_head_loss", ptr(xh2), ptr(mask), ptr(rstd), B, H, ptr(Wh), ptr(bh), ptr(ls), 1.0, 0.5, 0, 5, None, ptr(actions), None, ptr(old_logp), ptr(adv), None, ptr(factor), ptr(active), 0.2, 0.01, 0, 0, 0, 0, None, ptr(dz2), None, ptr(ps_f), ptr(part_h), n_wg, s), GB(B * (512 + 512 + 20 + 60))),
        ("critic_head_loss_fused_dw", lambda: call("harl_critic_head_loss", ptr(xh2), ptr(mask), ptr(rstd), B, H, ptr(Wv), ptr(bv), None, ptr(vp), ptr(ret), ptr(vn), 0.2, 1, 1, 10.0, 0, 0, ptr(dz2), None, ptr(ps_f), ptr(part_h), n_wg, s), GB(B * (512 + 512 + 20 + 8))),
        ("critic_head_loss", lambda: call("harl_critic_head_loss", ptr(xh2), ptr(mask), ptr(rstd), B, H, ptr(Wv), ptr(bv), None, ptr(vp), ptr(ret), ptr(vn), 0.2, 1, 1, 10.0, 0, 0, ptr(dz2), ptr(dhead), ptr(ps), None, 0, s), GB(B * (512 + 512 + 128 + 8))),
        ("critic_head_values", lambda: call("harl_critic_head_values", ptr(xh2), B, H, ptr(Wv), ptr(bv), ptr(vals), s), GB(B * 516)),
        ("gae_returns", lambda: call("harl_gae_returns", ptr(rew), ptr(vpT), ptr(mk), ptr(mk), ptr(vpT[-1].contiguous()), ptr(vn), ptr(rets), ptr(advs), T, N, 0.99, 0.9405, 1, 1, 0, s), GB(B * 24)),
        ("gradnorm_clip_adam", lambda: call("harl_gradnorm_clip_adam", ptr(pp), ptr(gg), ptr(mm), ptr(vv), P, None, 1, 10.0, 5e-4, 0.9, 0.999, 1e-5, 0.0, 0.1, 0.001, None, s), GB(P * 28)),
    ]
    # ---- the 17-agent HATRPO shape (204 800 rows per launch, obs 393 -> KP 416): tangent / wide-input kernels of the FVP
    B2 = 200 * 1024
    ns2 = (B2 + 31) // 32
    mp2 = ns2 * 32
    x0n416 = rn(mp2 * 416)
    W1h, W1hd = rn(H * 393) * 0.05, rn(H * 393) * 0.05
    wimg416 = torch.empty(3 * H * 416 // 2, device=dev)
    timg = torch.empty(3 * H * H, device=dev)
    xa, xb, xc, xo = rn(mp2 * H), rn(mp2 * H), rn(mp2 * H), torch.empty(mp2 * H, device=dev)
    Wd = rn(H * H) * 0.1
    mask2_, rstd2_ = mask[:ns2 * 2 * 64], rstd[:mp2]
    fl2 = 2.0 * B2 * H * H
    jobs += [
        ("h17_fwd_wide_K416", lambda: call("harl_mlp_fwd_wide", ptr(x0n416), B2, 416, ptr(W1h), 393, ptr(b), H, ptr(wimg416), ptr(xo), ptr(mask2_), ptr(rstd2_), s), GB(B2 * (1664 + 512 + 20))),
        ("h17_tangent_wide_K416", lambda: call("harl_mlp_tangent_wide", ptr(x0n416), B2, 416, ptr(W1hd), 393, ptr(b), H, ptr(wimg416), ptr(xa), ptr(mask2_), ptr(rstd2_), ptr(xo), s), GB(B2 * (1664 + 512 + 512 + 20))),
        ("h17_tangent_hidden2", lambda: call("harl_mlp_tangent_hidden2", ptr(xb), ptr(xa), B2, H, H, ptr(W), ptr(Wd), ptr(b), ptr(timg), ptr(xc), ptr(mask2_), ptr(rstd2_), ptr(xo), s), GB(B2 * (512 * 4 + 20))),
        ("h17_fwd_hidden", lambda: call("harl_mlp_fwd_hidden", ptr(xa), B2, H, H, ptr(W), ptr(b), ptr(xo), ptr(mask2_), ptr(rstd2_), s), GB(B2 * (1024 + 20))),
        ("h17_bwd_dx", lambda: call("harl_mlp_bwd_dx", ptr(xa), ptr(xb), ptr(mask2_), ptr(rstd2_), B2, H, H, ptr(W), ptr(xo), None, 0, None, 0, s), GB(B2 * (1536 + 20))),
        ("h17_dw_hidden", lambda: call("harl_mlp_dw_partials", ptr(xa), 0, 0, H, ptr(xb), 0, 0, None, None, None, H, B2, ptr(part), n_wg, s), GB(B2 * 1024)),
    ]
    print(f"B={B}  H={H}  reps={args.reps}  lib={_lib.LIB_PATH}")
    for name, fn, (unit, work) in jobs:
        if args.filters and not any(f in name for f in args.filters):
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        evs = []
        for _ in range(args.reps):
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b_.record()
            evs.append((a, b_))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b_) for a, b_ in evs)
        med, mn = ms[len(ms) // 2], ms[0]
        print(f"{name:22s} median {med:8.4f} ms  min {mn:8.4f} ms   {work / (med * 1e-3):8.2f} {unit}")


if __name__ == "__main__":
    main()
