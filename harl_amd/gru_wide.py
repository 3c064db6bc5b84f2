"""GRU policies with 128-wide hidden layers (harl/models/base/rnn.py:8-81 on the reference's default
``hidden_sizes: [128, 128]``), composed from verified layer kernels.

The fused recurrence kernels of csrc/gru.hip keep the three bf16 images of W_hh resident in LDS, which stops at H = 64.  Here
a time step is four launches instead: the three gate products ``gh_g = W_hg h~ + b_hg`` through ``harl_mlp_linear`` (the
hidden-layer GEMM without an epilogue) and one element-wise cell kernel (csrc/gru_cell.hip); the input halves of the gates
for ALL steps are three more GEMM launches up front, the output LayerNorm one ``harl_rownorm``.  The backward pass mirrors
it (cell backward + ``W_hg^T dgh_g`` GEMMs per step) and ends in the tensors the fused ``harl_gru_bwd`` produces -- the gate
gradients ``net.rnn_dgate`` and the gradient into the last MLP layer ``net.dz[1]`` -- so that ``_FlatNet.backward_trunk``
continues unchanged (weight gradients of the six gate blocks, MLP layers).

Coverage path: launch-bound (4 L small launches per pass), not tuned.  Parity-green on hardware since round 3 (the
reference goldens ``rnn_box_h128`` / ``rnn_disc_h128_mb2``, and the same composition on 64-wide GRUs against the goldens of
the fused kernels): the default for 128-wide GRUs; ``HARL_GRU128=0`` makes ``_FlatNet`` refuse them as before.
"""
from __future__ import annotations

import torch

from ._lib import call, ptr, stream


def ensure_ws(net, mp_rows: int) -> None:
    """Extra workspaces next to the ones of ``_FlatNet._ensure_ws`` (``mp_rows`` = padded row count of the largest batch)."""
    H = net.hidden_sizes[-1]
    dev = net.device_
    f32 = torch.float32
    net.rnn_hraw = torch.empty(mp_rows * H, dtype=f32, device=dev)       # h_l of every step (input of the output LayerNorm)
    net.rnn_gh = torch.empty(3 * mp_rows * H, dtype=f32, device=dev)     # per-step gate products / backward GEMM outputs
    net.rnn_gz = torch.empty(mp_rows * H, dtype=f32, device=dev)         # G_l * z_l carried to the step before
    net.rnn_zero_bias = torch.zeros(H, dtype=f32, device=dev)


def forward(net, seq: dict, save: bool) -> None:
    """x_hat of the last MLP layer (net.xh[-1], L*m_pad rows) -> net.rnn_y / net.rnn_rstd (+ saved gates when ``save``)."""
    H = net.hidden_sizes[-1]
    L, mp = seq["L"], seq["m_pad"]
    M, n = L * mp, mp * H
    gp, sv, s = net.gru_pack, net.rnn_saved, stream()
    gi = [net.rnn_gi[g * M * H:(g + 1) * M * H] for g in range(3)]
    gh = [net.rnn_gh[g * n:(g + 1) * n] for g in range(3)]
    Wih, bih, Whh, bhh = gp["Wih"], gp["bih"], gp["Whh"], gp["bhh"]
    for g in range(3):  # input halves of the gates, all steps at once
        call("harl_mlp_linear", ptr(net.xh[-1]), M, H, H, ptr(Wih[g * H * H:(g + 1) * H * H]), ptr(bih[g * H:(g + 1) * H]),
             ptr(gi[g]), s, tag="gru_gi")
    hpm, mask_rows = sv[0], seq["mask_rows"]
    call("harl_gru_cell_init", ptr(seq["h0"]), ptr(mask_rows), H, mp, ptr(hpm[:n]), s)
    for l in range(L):
        lo, hi = l * n, (l + 1) * n
        last = l == L - 1
        for g in range(3):
            call("harl_mlp_linear", ptr(hpm[lo:hi]), mp, H, H, ptr(Whh[g * H * H:(g + 1) * H * H]), ptr(bhh[g * H:(g + 1) * H]),
                 ptr(gh[g]), s, tag="gru_gh")
        sl = (lambda t: ptr(t[lo:hi])) if save else (lambda t: None)  # noqa: E731
        call("harl_gru_cell_fwd", ptr(gi[0][lo:hi]), ptr(gi[1][lo:hi]), ptr(gi[2][lo:hi]), ptr(gh[0]), ptr(gh[1]), ptr(gh[2]),
             ptr(hpm[lo:hi]), None if last else ptr(mask_rows[(l + 1) * mp:(l + 2) * mp]), H, mp,
             sl(sv[1]), sl(sv[2]), sl(sv[3]), sl(sv[4]), ptr(net.rnn_hraw[lo:hi]),
             None if last else ptr(hpm[hi:hi + n]), ptr(seq.get("h_last")) if last else None, s, tag="gru_cell_fwd")
    call("harl_rownorm", ptr(net.rnn_hraw), M, H, ptr(net.rnn_y), ptr(net.rnn_rstd), s, tag="gru_norm")


def backward(net, seq: dict) -> None:
    """d(loss)/d(h_l) of every step (net.dz[0]) -> gate gradients net.rnn_dgate [dr, dz, dn, dhn] and net.dz[1] = the
    gradient at the last MLP layer's pre-activation (what ``harl_gru_bwd`` leaves behind)."""
    H = net.hidden_sizes[-1]
    L, mp = seq["L"], seq["m_pad"]
    M, n = L * mp, mp * H
    gp, sv, dg, s = net.gru_pack, net.rnn_saved, net.rnn_dgate, stream()
    mask_rows = seq["mask_rows"]
    WhhT = gp["Whh"].view(3, H, H).transpose(1, 2).contiguous()  # t_g = W_hg^T dgh_g as a forward GEMM with the transposed block
    t = [net.rnn_gh[g * n:(g + 1) * n] for g in range(3)]
    gz, zero = net.rnn_gz[:n], net.rnn_zero_bias
    for l in range(L - 1, -1, -1):
        lo, hi = l * n, (l + 1) * n
        nxt = l < L - 1
        call("harl_gru_cell_bwd", ptr(net.dz[0][lo:hi]), ptr(t[0]) if nxt else None, ptr(t[1]) if nxt else None,
             ptr(t[2]) if nxt else None, ptr(mask_rows[(l + 1) * mp:(l + 2) * mp]) if nxt else None,
             ptr(sv[1][lo:hi]), ptr(sv[2][lo:hi]), ptr(sv[3][lo:hi]), ptr(sv[4][lo:hi]), ptr(sv[0][lo:hi]), H, mp, ptr(gz),
             ptr(dg[0][lo:hi]), ptr(dg[1][lo:hi]), ptr(dg[2][lo:hi]), ptr(dg[3][lo:hi]), s, tag="gru_cell_bwd")
        if l > 0:
            for g, src in enumerate((dg[0], dg[1], dg[3])):  # d gh = [dr, dz, dhn]
                call("harl_mlp_linear", ptr(src[lo:hi]), mp, H, H, ptr(WhhT[g]), ptr(zero), ptr(t[g]), s, tag="gru_gh_bwd")
    # gradient into the last MLP layer: sum over the gates of  relu' . LNbwd(W_ig'^T d gi_g)  (linear in d gi)
    Wih = gp["Wih"]
    tmp = net.rnn_hraw
    for g in range(3):
        out = net.dz[1] if g == 0 else tmp
        call("harl_mlp_bwd_dx", ptr(dg[g]), ptr(net.xh[-1]), ptr(net.rmask[-1]), ptr(net.rstd[-1]), M, H, H,
             ptr(Wih[g * H * H:(g + 1) * H * H]), ptr(out), None, 0, None, 0, s, tag="bwd_dx")
        if g > 0:
            net.dz[1][:M * H].add_(tmp[:M * H])
