"""GRU policies with 128-wide hidden layers (harl/models/base/rnn.py:8-81 on the reference's default
``hidden_sizes: [128, 128]``), composed from verified layer kernels.

The fused recurrence kernels of csrc/gru.hip keep the three bf16 images of W_hh resident in LDS, which stops at H = 64.  Here
a time step is four launches instead: the three gate products ``gh_g = W_hg h~ + b_hg`` through ``harl_mlp_linear`` (the
hidden-layer GEMM without an epilogue) and one element-wise cell kernel (csrc/gru_cell.hip); the input halves of the gates
for ALL steps are three more GEMM launches up front, the output LayerNorm one ``harl_rownorm``.  The backward pass mirrors
it (cell backward + ``W_hg^T dgh_g`` GEMMs per step) and ends in the tensors the fused ``harl_gru_bwd`` produces -- the gate
gradients ``net.rnn_dgate`` and the gradient into the last MLP layer ``net.dz[1]`` -- so that ``_FlatNet.backward_trunk``
continues unchanged (weight gradients of the six gate blocks, MLP layers).

Stacked layers (``recurrent_n`` > 1, rnn.py:14) run on this composition for 64- and 128-wide GRUs: layer after layer over
the whole chunk (forward), top layer first (backward: the gradient a layer sends to the one below is three more GEMMs over all
steps); goldens ``rnn2_box_h64`` / ``rnn2_disc_h128_naive_mb2`` recorded from the reference.

Coverage path: launch-bound (4 L small launches per pass and layer), not tuned.  Parity-green on hardware since round 3 (the
reference goldens ``rnn_box_h128`` / ``rnn_disc_h128_mb2``, and the same composition on 64-wide GRUs against the goldens of
the fused kernels): the default for 128-wide GRUs; ``HARL_GRU128=0`` makes ``_FlatNet`` refuse them as before.
"""
from __future__ import annotations

import os

import torch

from ._lib import call, ptr, stream


def _lin3(xs, rows, H, Ws, bs, outs, s, tag):
    """Three raw gate products  z_g = W_g x_g + b_g  over ``rows`` rows: ONE launch (harl_mlp_linear3, round 6) unless
    HARL_GRU_LIN3=0 (three harl_mlp_linear launches: A/B and the bit-for-bit cross-check of the two)."""
    if os.environ.get("HARL_GRU_LIN3", "1") != "0":
        call("harl_mlp_linear3", ptr(xs[0]), ptr(xs[1]), ptr(xs[2]), rows, H, H, ptr(Ws[0]), ptr(Ws[1]), ptr(Ws[2]),
             ptr(bs[0]), ptr(bs[1]), ptr(bs[2]), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), s, tag=tag)
        return
    for g in range(3):
        call("harl_mlp_linear", ptr(xs[g]), rows, H, H, ptr(Ws[g]), ptr(bs[g]), ptr(outs[g]), s, tag=tag)


def ensure_ws(net, mp_rows: int) -> None:
    """Extra workspaces next to the ones of ``_FlatNet._ensure_ws`` (``mp_rows`` = padded row count of the largest batch)."""
    H = net.hidden_sizes[-1]
    dev = net.device_
    f32 = torch.float32
    # h_l of every step, per GRU layer: input of the next layer / of the output LayerNorm, B operand of the next layer's W_ih gradient
    net.rnn_hraw_l = [torch.empty(mp_rows * H, dtype=f32, device=dev) for _ in range(net.recurrent_n)]
    net.rnn_hraw = net.rnn_hraw_l[-1]
    net.rnn_gh = torch.empty(3 * mp_rows * H, dtype=f32, device=dev)     # per-step gate products / backward GEMM outputs
    net.rnn_gz = torch.empty(mp_rows * H, dtype=f32, device=dev)         # G_l * z_l carried to the step before
    net.rnn_zero_bias = torch.zeros(H, dtype=f32, device=dev)
    net.rnn_tmp = torch.empty(mp_rows * H, dtype=f32, device=dev)        # GEMM outputs that are summed into another tensor
    # d(loss)/d(h) of every step of a LOWER layer (stacked GRUs: what the layer above sends down), ping-pong with net.dz[0]
    net.rnn_dh = torch.empty(mp_rows * H, dtype=f32, device=dev) if net.recurrent_n > 1 else None


def _layer_state(seq: dict, key: str, layer: int, n_layers: int, H: int):
    """Column block ``layer`` of the [m_pad, n_layers * H] state tensor ``seq[key]`` as a contiguous [m_pad, H] tensor
    (a view when there is one layer)."""
    t = seq.get(key)
    if t is None or n_layers == 1:
        return t
    return t.reshape(t.shape[0], n_layers, H)[:, layer, :].contiguous()


def forward(net, seq: dict, save: bool) -> None:
    """x_hat of the last MLP layer (net.xh[-1], L*m_pad rows) -> net.rnn_y / net.rnn_rstd (+ saved gates when ``save``).
    Stacked layers (recurrent_n > 1) run one after the other over the whole chunk: layer k at step l needs layer k - 1 at step
    l and itself at step l - 1, so a layer-major order is as valid as the reference's step-major one (rnn.py:14: nn.GRU with
    num_layers) and lets the input halves of a layer's gates be three GEMMs over ALL steps.  Every layer's state is
    multiplied by the mask of the step (rnn.py:27,67)."""
    H = net.hidden_sizes[-1]
    L, mp = seq["L"], seq["m_pad"]
    M, n = L * mp, mp * H
    RN = net.recurrent_n
    s = stream()
    gi = [net.rnn_gi[g * M * H:(g + 1) * M * H] for g in range(3)]
    gh = [net.rnn_gh[g * n:(g + 1) * n] for g in range(3)]
    mask_rows = seq["mask_rows"]
    h_last_all = seq.get("h_last")
    for layer in range(RN):
        gp, sv = net.gru_packs[layer], net.rnn_saved_l[layer]
        Wih, bih, Whh, bhh = gp["Wih"], gp["bih"], gp["Whh"], gp["bhh"]
        xin = net.xh[-1] if layer == 0 else net.rnn_hraw_l[layer - 1]
        hraw = net.rnn_hraw_l[layer]
        Wi = [Wih[g * H * H:(g + 1) * H * H] for g in range(3)]
        Wh = [Whh[g * H * H:(g + 1) * H * H] for g in range(3)]
        bi, bh = [bih[g * H:(g + 1) * H] for g in range(3)], [bhh[g * H:(g + 1) * H] for g in range(3)]
        _lin3([xin] * 3, M, H, Wi, bi, gi, s, "gru_gi")  # input halves of the gates, all steps at once
        hpm = sv[0]
        h0 = _layer_state(seq, "h0", layer, RN, H)
        h_last = None if h_last_all is None else (h_last_all if RN == 1 else torch.empty(mp, H, dtype=h0.dtype, device=h0.device))
        call("harl_gru_cell_init", ptr(h0), ptr(mask_rows), H, mp, ptr(hpm[:n]), s)
        for l in range(L):
            lo, hi = l * n, (l + 1) * n
            last = l == L - 1
            _lin3([hpm[lo:hi]] * 3, mp, H, Wh, bh, gh, s, "gru_gh")
            sl = (lambda t: ptr(t[lo:hi])) if save else (lambda t: None)  # noqa: E731
            call("harl_gru_cell_fwd", ptr(gi[0][lo:hi]), ptr(gi[1][lo:hi]), ptr(gi[2][lo:hi]), ptr(gh[0]), ptr(gh[1]), ptr(gh[2]),
                 ptr(hpm[lo:hi]), None if last else ptr(mask_rows[(l + 1) * mp:(l + 2) * mp]), H, mp,
                 sl(sv[1]), sl(sv[2]), sl(sv[3]), sl(sv[4]), ptr(hraw[lo:hi]),
                 None if last else ptr(hpm[hi:hi + n]), ptr(h_last) if last else None, s, tag="gru_cell_fwd")
        if h_last_all is not None and RN > 1:
            h_last_all.reshape(mp, RN, H)[:, layer, :].copy_(h_last)
    call("harl_rownorm", ptr(net.rnn_hraw_l[-1]), M, H, ptr(net.rnn_y), ptr(net.rnn_rstd), s, tag="gru_norm")


def backward(net, seq: dict) -> None:
    """d(loss)/d(h_l) of every step of the TOP layer (net.dz[0]) -> per layer the gate gradients net.rnn_dgate_l[k] = [dr, dz,
    dn, dhn], and net.dz[1] = the gradient at the last MLP layer's pre-activation (what ``harl_gru_bwd`` leaves behind)."""
    H = net.hidden_sizes[-1]
    L, mp = seq["L"], seq["m_pad"]
    M, n = L * mp, mp * H
    RN = net.recurrent_n
    s = stream()
    mask_rows = seq["mask_rows"]
    t = [net.rnn_gh[g * n:(g + 1) * n] for g in range(3)]
    gz, zero = net.rnn_gz[:n], net.rnn_zero_bias
    dh = net.dz[0]  # d(loss)/d(h of this layer), all steps
    for layer in range(RN - 1, -1, -1):
        gp, sv, dg = net.gru_packs[layer], net.rnn_saved_l[layer], net.rnn_dgate_l[layer]
        WhhT = gp["Whh"].view(3, H, H).transpose(1, 2).contiguous()  # t_g = W_hg^T dgh_g as a forward GEMM with the transposed block
        for l in range(L - 1, -1, -1):
            lo, hi = l * n, (l + 1) * n
            nxt = l < L - 1
            call("harl_gru_cell_bwd", ptr(dh[lo:hi]), ptr(t[0]) if nxt else None, ptr(t[1]) if nxt else None,
                 ptr(t[2]) if nxt else None, ptr(mask_rows[(l + 1) * mp:(l + 2) * mp]) if nxt else None,
                 ptr(sv[1][lo:hi]), ptr(sv[2][lo:hi]), ptr(sv[3][lo:hi]), ptr(sv[4][lo:hi]), ptr(sv[0][lo:hi]), H, mp, ptr(gz),
                 ptr(dg[0][lo:hi]), ptr(dg[1][lo:hi]), ptr(dg[2][lo:hi]), ptr(dg[3][lo:hi]), s, tag="gru_cell_bwd")
            if l > 0:  # d gh = [dr, dz, dhn]
                _lin3([dg[0][lo:hi], dg[1][lo:hi], dg[3][lo:hi]], mp, H, [WhhT[0], WhhT[1], WhhT[2]], [zero] * 3, t, s, "gru_gh_bwd")
        if layer > 0:
            # into the layer below: d(loss)/d(h^{layer-1}_l) = sum_g W_ig^T d gi_g for all steps (no LayerNorm in between)
            WihT = gp["Wih"].view(3, H, H).transpose(1, 2).contiguous()
            dh = net.rnn_dh
            for g in range(3):
                out = dh if g == 0 else net.rnn_tmp
                call("harl_mlp_linear", ptr(dg[g]), M, H, H, ptr(WihT[g]), ptr(zero), ptr(out), s, tag="gru_gi_bwd")
                if g > 0:
                    dh[:M * H].add_(net.rnn_tmp[:M * H])
    # gradient into the last MLP layer: sum over the gates of  relu' . LNbwd(W_ig'^T d gi_g)  (linear in d gi)
    dg0 = net.rnn_dgate_l[0]
    Wih = net.gru_packs[0]["Wih"]
    tmp = net.rnn_tmp
    for g in range(3):
        out = net.dz[1] if g == 0 else tmp
        call("harl_mlp_bwd_dx", ptr(dg0[g]), ptr(net.xh[-1]), ptr(net.rmask[-1]), ptr(net.rstd[-1]), M, H, H,
             ptr(Wih[g * H * H:(g + 1) * H * H]), ptr(out), None, 0, None, 0, s, tag="bwd_dx")
        if g > 0:
            net.dz[1][:M * H].add_(tmp[:M * H])


def tangent(net, seq: dict, xdot: torch.Tensor, pack_d: torch.Tensor, ws: dict) -> torch.Tensor:
    """Forward-mode tangent through the composed GRU (HATRPO's Fisher-vector product, trpo_util.py:132-158): ``xdot`` = tangent
    of the last MLP layer's x_hat (ATL, L*m_pad rows), ``pack_d`` = tangent of the folded packs laid out like net.pack_arena
    (harl_fold_tangent_table) -> y_dot, the tangent of rnn.norm's output over all steps (ATL).  Uses the gates the forward pass
    saved (save=True).  Per layer: the input halves of the gate tangents for ALL steps as six raw GEMMs (W_i x_dot and
    W_i_dot x + b_i_dot per gate), then per step three or six GEMMs on the carried state (W_h h~_dot -- absent at the first
    step, h0 carries no tangent -- and W_h_dot h~ + b_h_dot) and one harl_gru_cell_tangent; layers run one after the other over
    the whole chunk as in ``forward``."""
    H = net.hidden_sizes[-1]
    L, mp = seq["L"], seq["m_pad"]
    M, n = L * mp, mp * H
    RN = net.recurrent_n
    s = stream()
    zero = net.rnn_zero_bias
    if ws.get("gru_rows", 0) < M or ws.get("gru_layers", 0) < RN:
        dev, f32 = net.device_, torch.float32
        e = lambda k: torch.empty(k, dtype=f32, device=dev)  # noqa: E731
        ws.update(gru_rows=M, gru_layers=RN, gia=[e(M * H) for _ in range(3)], gib=[e(M * H) for _ in range(3)],
                  gha=[e(n) for _ in range(3)], ghb=[e(n) for _ in range(3)], hdot=[e(M * H) for _ in range(RN)],
                  hpmd=[e(n) for _ in range(2)], ydot=e(M * H))
    gia, gib, gha, ghb, hpmd = ws["gia"], ws["gib"], ws["gha"], ws["ghb"], ws["hpmd"]
    mask_rows = seq["mask_rows"]
    n3 = 3 * H * H
    xin, xin_dot = net.xh[-1], xdot
    for layer in range(RN):
        gp, sv = net.gru_packs[layer], net.rnn_saved_l[layer]
        b0 = net._gru_pack_base + 2 * layer * (n3 + 3 * H)  # the same block of the tangent arena (nets._build_tables)
        Wihd, bihd = pack_d[b0:b0 + n3], pack_d[b0 + n3:b0 + n3 + 3 * H]
        Whhd, bhhd = pack_d[b0 + n3 + 3 * H:b0 + 2 * n3 + 3 * H], pack_d[b0 + 2 * n3 + 3 * H:b0 + 2 * n3 + 6 * H]
        blk = [slice(g * H * H, (g + 1) * H * H) for g in range(3)]
        bb = [slice(g * H, (g + 1) * H) for g in range(3)]
        _lin3([xin_dot] * 3, M, H, [gp["Wih"][b_] for b_ in blk], [zero] * 3, gia, s, "gru_gi_tan")
        _lin3([xin] * 3, M, H, [Wihd[b_] for b_ in blk], [bihd[b_] for b_ in bb], gib, s, "gru_gi_tan")
        hdot = ws["hdot"][layer]
        for l in range(L):
            lo, hi = l * n, (l + 1) * n
            first, last = l == 0, l == L - 1
            cur, nxt = hpmd[l & 1], hpmd[(l + 1) & 1]
            hpm_l = sv[0][lo:hi]
            _lin3([hpm_l] * 3, mp, H, [Whhd[b_] for b_ in blk], [bhhd[b_] for b_ in bb], ghb, s, "gru_gh_tan")
            if not first:
                _lin3([cur] * 3, mp, H, [gp["Whh"][b_] for b_ in blk], [zero] * 3, gha, s, "gru_gh_tan")
            ha = [None] * 3 if first else [ptr(t) for t in gha]
            call("harl_gru_cell_tangent", ptr(gia[0][lo:hi]), ptr(gia[1][lo:hi]), ptr(gia[2][lo:hi]), ptr(gib[0][lo:hi]),
                 ptr(gib[1][lo:hi]), ptr(gib[2][lo:hi]), ha[0], ha[1], ha[2], ptr(ghb[0]), ptr(ghb[1]), ptr(ghb[2]),
                 ptr(sv[1][lo:hi]), ptr(sv[2][lo:hi]), ptr(sv[3][lo:hi]), ptr(sv[4][lo:hi]), ptr(hpm_l),
                 None if first else ptr(cur), None if last else ptr(mask_rows[(l + 1) * mp:(l + 2) * mp]), H, mp,
                 ptr(hdot[lo:hi]), None if last else ptr(nxt), s, tag="gru_cell_tangent")
        xin, xin_dot = net.rnn_hraw_l[layer], hdot
    call("harl_act_ln_tangent", ptr(xin_dot), None, ptr(net.rnn_y), None, ptr(net.rnn_rstd), M, H, 0, ptr(ws["ydot"]), s,
         tag="gru_norm_tangent")
    return ws["ydot"]
