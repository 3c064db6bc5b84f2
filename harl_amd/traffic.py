"""Algorithmic HBM bytes of one launch of each streaming kernel, computed from the C-ABI arguments of that launch.

Measurement support only (bench.py's `roofline`): `_lib.call` evaluates the entry of the function it is
about to launch when kernel timing is enabled, so that  achieved GB/s = sum(bytes) / sum(HIP-event time)  per kernel family
uses the bytes of the launches that actually ran (training-mode and log-prob-mode forwards, actor and critic input
widths, ...).  "Algorithmic" = every operand the kernel must read once and every result it must write once, per
minibatch row, in the layouts of DESIGN.md section 2; weights, LDS staging and re-reads are not counted (they are what
`roofline.traffic`, the PMC figure, is compared against).

Per row of an activation of width H: the ATL image is 4H bytes, the ReLU bit mask H/8 bytes, the LayerNorm statistic 4 bytes.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence


def _act(H: int) -> float:  # x_hat image + ReLU mask + rstd of one row
    return 4.0 * H + H / 8.0 + 4.0


def _kp(D: int) -> int:  # padded input width of the cached normalised-input image
    return 32 * ((int(D) + 31) // 32)


def _fwd_fused2x(a: Sequence) -> float:
    # (x0n, M, W1p, D, b1p, W2p, b2p, H, store1, x1out, mask1, rstd1, x2out, mask2, rstd2, stream)
    M, D, H, store1 = a[1], a[3], a[7], a[8]
    return M * (4.0 * _kp(D) + (_act(H) if store1 else 0.0) + _act(H))


def _fwd_fused2(a: Sequence) -> float:
    # (X, ldx, idx, M, D, W1p, b1p, use_ln0, W2p, b2p, H, store1, x1out, mask1, rstd1, mu0, rstd0, x2out, mask2, rstd2, x0n, stream)
    M, D, H, store1, x0n = a[3], a[4], a[10], a[11], a[20]
    return M * (4.0 * D + (8.0 if a[2] else 0.0) + (_act(H) if store1 else 0.0) + _act(H) + 8.0 + (4.0 * _kp(D) if x0n else 0.0))


def _fwd_hidden(a: Sequence) -> float:  # (xin, M, HI, HO, Wp, bp, xout, relu_mask, rstd, stream)
    return a[1] * (4.0 * a[2] + _act(a[3]))


def _fwd_wide(a: Sequence) -> float:  # (x0n, M, KP, Wp, D, bp, H, w_img, xout, relu_mask, rstd, stream)
    return a[1] * (4.0 * a[2] + _act(a[6]))


def _fwd_input(a: Sequence) -> float:
    # (X, ldx, idx, M, D, Wp, bp, use_ln0, H, xout, relu_mask, rstd, mu0, rstd0, x0n, stream)
    return a[3] * (4.0 * a[4] + (8.0 if a[2] else 0.0) + _act(a[8]) + 8.0 + (4.0 * _kp(a[4]) if a[14] else 0.0))


def _x0n_wide(a: Sequence) -> float:  # (X, ldx, idx, M, D, use_ln0, x0n, mu0, rstd0, stream)
    return a[3] * (4.0 * a[4] + (8.0 if a[2] else 0.0) + 4.0 * _kp(a[4]) + 8.0)


def _tangent_wide(a: Sequence) -> float:  # (x0n, M, KP, Wdp, D, bdp, H, w_img, x1, mask1, rstd1, x1dot, stream)
    return a[1] * (4.0 * a[2] + _act(a[6]) + 4.0 * a[6])


def _bwd_dx(a: Sequence) -> float:
    # (dz, xprev, relu_mask_prev, rstd_prev, M, HO, HI, Wp, dz_prev, x0n, kp0, dw_part, n_wg, stream)
    M, HO, HI = a[4], a[5], a[6]
    return M * (4.0 * HO + _act(HI) + (4.0 * HI if a[8] else 0.0) + (4.0 * a[10] if a[9] else 0.0))


def _bwd_dx_dw(a: Sequence) -> float:
    # (dz, xprev, relu_mask_prev, rstd_prev, M, HO, HI, Wp, dz_prev, x0n, kp0, dw1_part, dw2_part, n_wg, fill, stream): the same
    # operands as harl_mlp_bwd_dx -- the weight gradient of the layer itself re-uses dz and x_hat_prev, read ONCE
    M, HO, HI = a[4], a[5], a[6]
    return M * (4.0 * HO + _act(HI) + (4.0 * HI if a[8] else 0.0) + (4.0 * a[10] if a[9] else 0.0))


def _dw_partials(a: Sequence) -> float:
    # (a, a_kind, lda, HO, b, b_kind, ldx, idx, mu0, rstd0, K, M, part, n_wg, stream): dz rows (a_kind 0: ATL(HO); 1: row-major
    # head gradient of `lda` floats) and the layer's input rows (ATL(K), or raw rows gathered through idx)
    lda, HO, K, M = a[2], a[3], a[10], a[11]
    return M * (4.0 * (lda if a[1] else HO) + 4.0 * K)


def _dw_partials_multi(a: Sequence) -> float:
    # (n, a_ptrs, b_ptrs, part_ptrs, HO, K, M, n_wg, stream): n problems, each reads its dz rows (ATL(HO)) and input rows (ATL(K))
    n, HO, K, M = a[0], a[4], a[5], a[6]
    return n * M * (4.0 * HO + 4.0 * K)


def _gru_fwd(a: Sequence) -> float:
    # (xin, mask_rows, h0, Wih, bih, Whh, bhh, H, L, m_pad, y, rstd_y, hpm, r, z, n, hn, h_last, save, gi_ws, stream)
    # save bit 1: the input half of the gates was written by harl_mlp_fwd_trunk -- the three gate images are read instead of x_hat
    H, L, m_pad, save = a[7], a[8], a[9], a[18]
    xin = 3 * 4.0 * H if (save & 2) else 4.0 * H
    return L * m_pad * (xin + 4.0 + 4.0 * H + 4.0 + (5 * 4.0 * H if (save & 1) else 0.0))


def _gru_bwd(a: Sequence) -> float:
    # reads dhout + the five saved images + the MLP output feeding the GRU (+mask, rstd); writes the four gate gradients
    # and the gradient into the MLP
    H, L, m_pad = a[9], a[10], a[11]
    if not a[19]:  # dz_mlp = NULL: the input side is harl_mlp_bwd_trunk's first stage
        return L * m_pad * (6 * 4.0 * H + 4.0 + 4 * 4.0 * H)
    return L * m_pad * (6 * 4.0 * H + 4.0 + _act(H) + 4 * 4.0 * H + 4.0 * H)


def _nonnull(arr) -> int:
    return sum(1 for p in arr if p)


def _fwd_trunk(a: Sequence) -> float:
    # (x0n, M, KP, W1p, D, b1p, H, w_img, n_hidden, Wp, bp, xout, relu_mask, rstd, Wih, bih, bhh, gi_ws, stream): the input image,
    # the activation records that are written, the three gate images
    M, KP, H = a[1], a[2], a[6]
    return M * (4.0 * KP + _nonnull(a[11]) * _act(H) + (3 * 4.0 * H if a[17] else 0.0))


def _bwd_trunk(a: Sequence) -> float:
    # (M, H, n_hidden, Wih, dr, dzg, dn, dz_in, Wp, xh, relu_mask, rstd, dz_out, stream): gate gradients (or dz of the top layer) in,
    # one activation record read and one dz written per stage
    M, H, nh = a[0], a[1], a[2]
    top = (3 * 4.0 * H + _act(H) + 4.0 * H) if a[3] else 4.0 * H
    return M * (top + nh * (_act(H) + 4.0 * H))


def _dw_partials_multi_v(a: Sequence) -> float:
    # (n, a_ptrs, b_ptrs, part_ptrs, HO, K[], tile0[], nt[], M, n_wg, stream): every problem reads its dz rows and its column group
    n, HO, M = a[0], a[4], a[8]
    return M * sum(4.0 * HO + 4.0 * 32 * a[7][k] for k in range(n))


def _head_rows(discrete: int, act_dim: int, avail) -> float:  # actions + old log-probs (+ availability mask) of one row
    w = 1 if discrete else act_dim
    return 4.0 * w + 4.0 * w + (4.0 * act_dim if avail else 0.0)


def _actor_head_loss(a: Sequence) -> float:
    # (xL, relu_mask, rstd, M, H, Whp, bhp, log_std, sx, sy, discrete, act_dim, idx, actions, avail, old_logp, adv, adv_moments,
    #  factor, active, clip, ent, agg_mean, trpo, m_valid, m_pad, logp_out, dzL, dhead, part_scalars, dw_part, n_wg, stream)
    M, H, disc, ad = a[3], a[4], a[10], a[11]
    w = 1 if disc else ad
    return M * (_act(H) + _head_rows(disc, ad, a[14]) + 4.0 + (4.0 if a[18] else 0.0) + (4.0 if a[19] else 0.0)
                + (4.0 * w if a[26] else 0.0) + 4.0 * H + (4.0 * 32 if (a[28] and not a[30]) else 0.0))  # dhead only when the head dW is not fused


def _actor_head_logp(a: Sequence) -> float:
    # (xL, M, H, Whp, bhp, log_std, sx, sy, discrete, act_dim, actions, avail, logp_out, old_logp, factor, agg_mean, head_out,
    #  m_valid, m_pad, stream)
    M, H, disc, ad = a[1], a[2], a[8], a[9]
    w = 1 if disc else ad
    return M * (4.0 * H + 4.0 * w + (4.0 * ad if a[11] else 0.0) + (4.0 * w if a[12] else 0.0) + (4.0 * w if a[13] else 0.0)
                + (8.0 if a[14] else 0.0) + (4.0 * ad if a[16] else 0.0))


def _critic_head_loss(a: Sequence) -> float:
    # (xL, relu_mask, rstd, M, H, Whp, bhp, idx, value_preds, returns, vn_stats, ...)
    return a[3] * (_act(a[4]) + 8.0 + 4.0 * a[4])


def _update_fwd_actor(a: Sequence) -> float:
    # (x0n, M, D, H, 7 weight pointers, sx, sy, discrete, act_dim, actions, avail, old_logp, adv, adv_moments, factor, active,
    #  clip, ent, agg_mean, trpo, logp_out, dz2, part_scalars, dw_part_head, n_part_rows, xh1, rmask1, rstd1, stream)
    M, D, H, disc, ad = a[1], a[2], a[3], a[13], a[14]
    w = 1 if disc else ad
    return M * (4.0 * _kp(D) + _head_rows(disc, ad, a[16]) + 4.0 + (4.0 if a[20] else 0.0) + (4.0 if a[21] else 0.0)
                + (4.0 * w if a[26] else 0.0) + 4.0 * H + (_act(H) if a[31] else 0.0))  # hybrid: + layer 1's record out


def _update_logp(a: Sequence) -> float:
    M, D, disc, ad = a[1], a[2], a[13], a[14]
    w = 1 if disc else ad
    return M * (4.0 * _kp(D) + 4.0 * w + (4.0 * ad if a[16] else 0.0) + (4.0 * w if a[17] else 0.0) + (4.0 * w if a[18] else 0.0)
                + (8.0 if a[19] else 0.0) + (4.0 * ad if a[21] else 0.0))


def _update_fwd_critic(a: Sequence) -> float:
    # (x0n, M, D, H, 6 weight pointers, value_preds, returns, vn_stats, clip, use_clipped, use_huber, delta, dz2, part_scalars,
    #  dw_part_head, n_part_rows, xh1, rmask1, rstd1, stream)
    return a[1] * (4.0 * _kp(a[2]) + 8.0 + 4.0 * a[3] + (_act(a[3]) if a[21] else 0.0))


def _update_last_actor(a: Sequence) -> float:
    # (xin, M, H, Wp, bp, Whp, bhp, log_std, sx, sy, discrete, act_dim, idx, actions, avail, old_logp, adv, adv_moments, factor,
    #  active, clip, ent, agg_mean, trpo, logp_out, dz, ...): x_hat_{L-1} in, the loss rows, dz_L out
    M, H, disc, ad = a[1], a[2], a[10], a[11]
    w = 1 if disc else ad
    return M * (4.0 * H + _head_rows(disc, ad, a[14]) + 4.0 + (4.0 if a[18] else 0.0) + (4.0 if a[19] else 0.0)
                + (4.0 * w if a[24] else 0.0) + 4.0 * H)


def _update_last_critic(a: Sequence) -> float:  # (xin, M, H, ...): x_hat_{L-1} in, value_preds + returns, dz_L out
    return a[1] * (4.0 * a[2] + 8.0 + 4.0 * a[2])


def _update_values(a: Sequence) -> float:
    return a[1] * (4.0 * _kp(a[2]) + 4.0)


def _update_bwd(a: Sequence) -> float:  # (x0n, dz2, M, D, H, ...): the normalised inputs and dz2; everything else is recomputed
    return a[2] * (4.0 * _kp(a[3]) + 4.0 * a[4])


def _panel_fwd(a: Sequence) -> float:  # (xin, M, KP, Wp, D, bp, HO, xout, relu_mask, rstd, stream)
    return a[1] * (4.0 * a[2] + _act(a[6]))


def _panel_bwd(a: Sequence) -> float:  # (dz, xprev, relu_mask_prev, rstd_prev, M, HO, HI, Wp, dz_prev, stream)
    return a[4] * (4.0 * a[5] + _act(a[6]) + 4.0 * a[6])


def _tangent_hidden(a: Sequence) -> float:
    # (xin_dot, xin, M, HI, HO, Wp, Wdp, bdp, xprimal, mask_in, rstd_in, xout_dot, stream): both inputs, the primal x_hat / mask /
    # rstd of the layer, the tangent out (the raw intermediate of the two-GEMM formulation is traffic, not algorithm)
    return a[2] * (2 * 4.0 * a[3] + _act(a[4]) + 4.0 * a[4])


def _gae(a: Sequence) -> float:
    # rewards, value_preds (T+1), masks (T+1), bad_masks (T+1) in; returns (T+1), advantages out
    T, n = a[8], a[9]
    return 4.0 * n * (T + 3 * (T + 1) + (T + 1) + T)


ALGORITHMIC_BYTES: Dict[str, Callable[[Sequence], float]] = {
    "harl_mlp_fwd_fused2x": _fwd_fused2x,
    "harl_mlp_fwd_fused2": _fwd_fused2,
    "harl_mlp_fwd_hidden": _fwd_hidden,
    "harl_mlp_fwd_wide": _fwd_wide,
    "harl_mlp_fwd_input": _fwd_input,
    "harl_mlp_x0n_wide": _x0n_wide,
    "harl_mlp_tangent_wide": _tangent_wide,
    "harl_mlp_bwd_dx": _bwd_dx,
    "harl_mlp_bwd_dx_dw": _bwd_dx_dw,
    "harl_mlp_dw_partials": _dw_partials,
    "harl_mlp_dw_partials_multi": _dw_partials_multi,
    "harl_mlp_dw_partials_multi_v": _dw_partials_multi_v,
    "harl_gru_dw6": lambda a: a[7] * 6 * 4.0 * 64,  # (dr, dz, dn, dhn, xhat, hpm, part, M, n_wg, stream): six ATL(64) images, each once
    "harl_mlp_fwd_trunk": _fwd_trunk,
    "harl_mlp_bwd_trunk": _bwd_trunk,
    "harl_gru_fwd": _gru_fwd,
    "harl_gru_bwd": _gru_bwd,
    "harl_actor_head_loss": _actor_head_loss,
    "harl_actor_head_logp": _actor_head_logp,
    "harl_critic_head_loss": _critic_head_loss,
    "harl_update_fwd_actor": _update_fwd_actor,
    "harl_update_logp": _update_logp,
    "harl_update_fwd_critic": _update_fwd_critic,
    "harl_update_values": _update_values,
    "harl_update_last_actor": _update_last_actor,
    "harl_update_last_critic": _update_last_critic,
    "harl_update_bwd": _update_bwd,
    "harl_gae_returns": _gae,
    "harl_mlp_panel_fwd": _panel_fwd,
    "harl_mlp_panel_bwd": _panel_bwd,
    "harl_mlp_tangent_hidden": _tangent_hidden,
    "harl_mlp_tangent_hidden2": _tangent_hidden,  # (same leading arguments: xin_dot, xin, M, HI, HO)
}


def algorithmic_bytes(name: str, args: Sequence) -> Optional[float]:
    f = ALGORITHMIC_BYTES.get(name)
    return None if f is None else float(f(args))


# ---------------------------------------------------------------------------------------------------------------------------
# Executed Linear-layer FLOPs of one launch (2 x rows x out x in per GEMM; head GEMMs included, element-wise work not) -- the
# unit of SURVEY.md section 8(d)'s end-to-end roofline.  bench.py sums them over an instrumented step: for HAPPO the total
# equals the closed form (2 + 3 ppo_epoch) F_actor + 3 critic_epoch F_critic minus the pre-update pass train() shares with
# its first epoch; for HATRPO (CG iterations, line-search steps: data dependent) it is the only count there is.
# ---------------------------------------------------------------------------------------------------------------------------
def _f_fused2x(a):  # (x0n, M, W1p, D, b1p, W2p, b2p, H, ...)
    return 2.0 * a[1] * (a[3] * a[7] + a[7] * a[7])


def _f_fused2(a):  # (X, ldx, idx, M, D, W1p, b1p, use_ln0, W2p, b2p, H, ...)
    return 2.0 * a[3] * (a[4] * a[10] + a[10] * a[10])


def _f_update_fwd_actor(a):  # forward of both layers + head, head dW and head backward (2 x act_dim x H each)
    M, D, H, ad = a[1], a[2], a[3], a[14]
    return 2.0 * M * (D * H + H * H + 3 * ad * H)


def _f_update_logp(a):
    M, D, H, ad = a[1], a[2], a[3], a[14]
    return 2.0 * M * (D * H + H * H + ad * H)


def _f_update_fwd_critic(a):
    M, D, H = a[1], a[2], a[3]
    return 2.0 * M * (D * H + H * H + 3 * H)


def _f_update_values(a):
    M, D, H = a[1], a[2], a[3]
    return 2.0 * M * (D * H + H * H + H)


def _f_update_last_actor(a):  # (xin, M, H, ..., act_dim at 11)
    return 2.0 * a[1] * (a[2] * a[2] + 3 * a[11] * a[2])


def _f_update_last_critic(a):
    return 2.0 * a[1] * (a[2] * a[2] + 3 * a[2])


def _f_update_bwd(a):  # (x0n, dz2, M, D, H): layer-1 recompute twice, dW2, dx, dW1
    M, D, H = a[2], a[3], a[4]
    return 2.0 * M * (2 * D * H + 2 * H * H + D * H)


def _f_bwd_dx(a):  # (dz, xprev, mask, rstd, M, HO, HI, Wp, dz_prev, x0n, kp0, ...): dx (+ fused first-layer dW)
    M, HO, HI = a[4], a[5], a[6]
    return 2.0 * M * (HO * HI + (HI * a[10] if a[9] else 0))


def _f_bwd_dx_dw(a):  # dx + the layer's own weight gradient (+ the fused first-layer one)
    M, HO, HI = a[4], a[5], a[6]
    return 2.0 * M * (2 * HO * HI + (HI * a[10] if a[9] else 0))


def _f_dw(a):  # (a, a_kind, lda, HO, b, b_kind, ldx, idx, mu0, rstd0, K, M, ...)
    return 2.0 * a[11] * a[3] * a[10]


def _f_dw_multi(a):  # (n, a_ptrs, b_ptrs, part_ptrs, HO, K, M, ...)
    return 2.0 * a[0] * a[6] * a[4] * a[5]


def _f_gru_fwd(a):  # input + recurrent halves of the three gates (save bit 1: the input half ran inside harl_mlp_fwd_trunk)
    H, L, m_pad = a[7], a[8], a[9]
    return 2.0 * L * m_pad * (3 if (a[18] & 2) else 6) * H * H


def _f_gru_bwd(a):
    H, L, m_pad = a[9], a[10], a[11]
    return 2.0 * L * m_pad * (6 if a[19] else 3) * H * H


def _f_fwd_trunk(a):  # first layer + hidden layers (+ the input half of the gates)
    M, D, H, nh = a[1], a[4], a[6], a[8]
    return 2.0 * M * (D * H + nh * H * H + (3 * H * H if a[17] else 0))


def _f_bwd_trunk(a):
    M, H, nh = a[0], a[1], a[2]
    return 2.0 * M * ((3 * H * H if a[3] else 0) + nh * H * H)


def _f_head(a_M, a_H, ad, train):
    return 2.0 * a_M * ad * a_H * (3 if train else 1)


ALGORITHMIC_FLOPS: Dict[str, Callable[[Sequence], float]] = {
    "harl_mlp_fwd_fused2x": _f_fused2x,
    "harl_mlp_fwd_fused2": _f_fused2,
    "harl_mlp_fwd_hidden": lambda a: 2.0 * a[1] * a[2] * a[3],
    "harl_mlp_linear": lambda a: 2.0 * a[1] * a[2] * a[3],
    "harl_mlp_linear3": lambda a: 3 * 2.0 * a[3] * a[4] * a[5],
    "harl_mlp_fwd_wide": lambda a: 2.0 * a[1] * a[4] * a[6],
    "harl_mlp_linear_wide": lambda a: 2.0 * a[1] * a[4] * a[6],
    "harl_mlp_fwd_input": lambda a: 2.0 * a[3] * a[4] * a[8],
    "harl_mlp_tangent_wide": lambda a: 2.0 * a[1] * a[4] * a[6],  # the inputs carry no tangent: one GEMM
    "harl_mlp_tangent_hidden": lambda a: 2.0 * 2 * a[2] * a[3] * a[4],
    "harl_mlp_tangent_hidden2": lambda a: 2.0 * 2 * a[2] * a[3] * a[4],
    "harl_mlp_bwd_dx": _f_bwd_dx,
    "harl_mlp_bwd_dx_dw": _f_bwd_dx_dw,
    "harl_mlp_dw_partials": _f_dw,
    "harl_mlp_dw_partials_multi": _f_dw_multi,
    "harl_mlp_dw_partials_multi_v": lambda a: 2.0 * a[8] * a[4] * 32 * sum(a[7][k] for k in range(a[0])),
    "harl_gru_dw6": lambda a: 2.0 * a[7] * 6 * 64 * 64,
    "harl_mlp_fwd_trunk": _f_fwd_trunk,
    "harl_mlp_bwd_trunk": _f_bwd_trunk,
    "harl_mlp_panel_fwd": lambda a: 2.0 * a[1] * a[4] * a[6],
    "harl_mlp_panel_bwd": lambda a: 2.0 * a[4] * a[5] * a[6],
    "harl_gru_fwd": _f_gru_fwd,
    "harl_gru_bwd": _f_gru_bwd,
    "harl_actor_head_loss": lambda a: _f_head(a[3], a[4], a[11], True) - (2.0 * a[3] * a[11] * a[4] if not a[30] else 0.0),
    "harl_actor_head_logp": lambda a: _f_head(a[1], a[2], a[9], False),
    "harl_actor_head_fvp": lambda a: 0.0,
    "harl_critic_head_loss": lambda a: _f_head(a[3], a[4], 1, True),
    "harl_update_fwd_actor": _f_update_fwd_actor,
    "harl_update_logp": _f_update_logp,
    "harl_update_fwd_critic": _f_update_fwd_critic,
    "harl_update_values": _f_update_values,
    "harl_update_last_actor": _f_update_last_actor,
    "harl_update_last_critic": _f_update_last_critic,
    "harl_update_bwd": _f_update_bwd,
}


def algorithmic_flops(name: str, args: Sequence) -> Optional[float]:
    f = ALGORITHMIC_FLOPS.get(name)
    if f is None:
        return None
    try:
        return float(f(args))
    except (TypeError, IndexError):
        return None
