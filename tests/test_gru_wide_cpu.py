"""128-wide GRU policies (harl_amd/gru_wide.py), host side without a GPU: the composed launch sequence of a whole train() with
the C-ABI calls recorded instead of executed, and the opt-in gate (experimental until the GPU parity tests have run)."""
import numpy as np
import pytest
import torch

from harl_amd.synthetic import Shapes, make_buffers
from tests.gpu_checks import Box, default_args
from tests.test_multidiscrete_cpu import stub_kernels  # noqa: F401  (fixture)


def _runner(hidden, T=10, N=6, A=2):
    from harl_amd.runner import RUNNER_REGISTRY
    a = default_args(hidden, ppo_epoch=2, critic_epoch=2, use_recurrent_policy=True, data_chunk_length=5)
    train = dict(n_rollout_threads=N, episode_length=T, use_valuenorm=True, use_linear_lr_decay=False,
                 use_proper_time_limits=True, model_dir=None, eval_interval=25, use_eval=False, log_interval=1,
                 num_env_steps=T * N * 2)
    return RUNNER_REGISTRY["happo"](dict(algo="happo"), dict(train=train, model=dict(a), algo=dict(a)), dict(state_type="EP"),
                                    obs_spaces=[Box((19,))] * A, share_obs_space=Box((11,)), act_spaces=[Box((3,))] * A,
                                    device=torch.device("cpu"))


def test_gru128_is_default_and_can_be_refused(stub_kernels, monkeypatch):  # noqa: F811
    monkeypatch.setenv("HARL_GRU128", "0")
    with pytest.raises(NotImplementedError):
        _runner([128, 128])
    monkeypatch.delenv("HARL_GRU128", raising=False)
    r = _runner([128, 128])
    assert r.actor[0].actor.gru_wide and r.critic.critic.gru_wide
    from harl_amd.hatrpo import HATRPO  # (round 4: HATRPO takes 128-wide and stacked GRUs through gru_wide.tangent)
    t = HATRPO(default_args([128, 128], use_recurrent_policy=True, kl_threshold=0.01, ls_step=10, accept_ratio=0.5,
                            backtrack_coeff=0.8), Box((19,)), Box((3,)), device=torch.device("cpu"))
    assert t.actor.gru_wide


def test_gru128_launch_sequence(stub_kernels, monkeypatch):  # noqa: F811
    monkeypatch.setenv("HARL_GRU128", "1")
    calls = stub_kernels
    T, N, A, H = 10, 6, 2, 128
    r = _runner([128, 128], T, N, A)
    sh = Shapes(T=T, N=N, A=A, obs_dim=19, share_obs_dim=11, act_dim=3, hidden_sizes=[128, 128])
    d = make_buffers(sh, 4, inactive_p=0.1, rnn=True)
    for ag in range(A):
        b = r.actor_buffer[ag]
        b.obs.copy_(torch.from_numpy(d.obs[ag]))
        b.actions.copy_(torch.from_numpy(d.actions[ag]))
        b.action_log_probs.copy_(torch.from_numpy(d.action_log_probs[ag]))
        b.rnn_states.copy_(torch.from_numpy(np.zeros((T + 1, N, 1, H), np.float32)))
    calls.clear()
    r.prep_training()
    infos, cinfo = r.train()
    assert len(infos) == A
    assert "harl_gru_fwd" not in calls and "harl_gru_bwd" not in calls      # the 64-wide fused kernels are not on this path
    L_full, L_chunk, n_upd = T, 5, 2
    # per agent: two full-length log-prob passes + n_upd chunked forward/backward passes; critic: n_upd chunked passes
    n_fwd_steps = A * (2 * L_full + n_upd * L_chunk) + n_upd * L_chunk
    n_bwd_steps = (A + 1) * n_upd * L_chunk
    assert len(calls["harl_gru_cell_fwd"]) == n_fwd_steps
    assert len(calls["harl_gru_cell_bwd"]) == n_bwd_steps
    n_passes_fwd = A * (2 + n_upd) + n_upd
    assert len(calls["harl_gru_cell_init"]) == n_passes_fwd and len(calls["harl_rownorm"]) == n_passes_fwd
    # gate GEMMs: ONE three-product launch (harl_mlp_linear3, round 6; three launches each before) per pass for the input
    # halves, per forward step, and per backward step except the first time step of a chunk
    n_bwd_passes = (A + 1) * n_upd
    assert len(calls["harl_mlp_linear3"]) == n_passes_fwd + n_fwd_steps + (n_bwd_steps - n_bwd_passes)
    assert "harl_mlp_linear" not in calls
    # every cell launch works on one time step of m_pad = 32 sequences at width 128; saved gates only in training passes
    assert all(c[8] == H and c[9] == 32 for c in calls["harl_gru_cell_fwd"])
    n_saving = sum(1 for c in calls["harl_gru_cell_fwd"] if c[10] is not None)
    assert n_saving == (A + 1) * n_upd * L_chunk
    # the last step of a pass has no successor: no next-step h~ / masks
    assert sum(1 for c in calls["harl_gru_cell_fwd"] if c[15] is None) == n_passes_fwd
    assert sum(1 for c in calls["harl_gru_cell_bwd"] if c[1] is None) == n_bwd_passes
    # weight gradients of the six gate blocks: ONE multi-problem launch of the two-operand kernel per backward pass (HO = K = 128)
    gate_dw = calls["harl_mlp_dw_partials_multi"]
    assert len(gate_dw) == n_bwd_passes and all(c[0] == 6 and c[4] == H and c[5] == H for c in gate_dw)


# ------------------------------------------------------------------------------------------------
# Functional check of the COMPOSITION on the CPU: the entry points gru_wide.py calls are emulated with torch ops on
# row-major [rows, H] arrays (the ATL layout is opaque to the host code: every emulated kernel uses the same one), and the
# composed forward / BPTT is compared with autograd through a plain torch GRU.  What this cannot see is the kernels' own
# indexing -- that is what the gated GPU tests are for.
# ------------------------------------------------------------------------------------------------
class _Arena:
    """address -> tensor view, for the raw pointers the host code hands to the C ABI"""

    def __init__(self):
        self.regs = []

    def add(self, t):
        flat = t.reshape(-1)
        self.regs.append((flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size(), flat))

    def view(self, p, n):
        for lo, hi, flat in self.regs:
            if lo <= p < hi:
                off = (p - lo) // flat.element_size()
                assert off + n <= flat.numel(), "emulated kernel would run past the end of a buffer"
                return flat[off:off + n]
        raise AssertionError(f"pointer {p:#x} is not inside any registered buffer")


def _emulate(arena, H):
    def linear(xin, M, HI, HO, Wp, bp, xout, s):
        x = arena.view(xin, M * HI).view(M, HI)
        W = arena.view(Wp, HO * HI).view(HO, HI)
        arena.view(xout, M * HO).view(M, HO).copy_(x @ W.t() + arena.view(bp, HO))

    def linear3(x0, x1, x2, M, HI, HO, W0, W1, W2, b0, b1, b2, o0, o1, o2, s):  # ONE launch for the three gate products (round 6)
        for x_, W_, b_, o_ in ((x0, W0, b0, o0), (x1, W1, b1, o1), (x2, W2, b2, o2)):
            linear(x_, M, HI, HO, W_, b_, o_, s)

    def cell_init(h0, mask_rows, H_, mp, hpm0, s):
        arena.view(hpm0, mp * H).view(mp, H).copy_(arena.view(h0, mp * H).view(mp, H) * arena.view(mask_rows, mp).view(mp, 1))

    def cell_fwd(gi_r, gi_z, gi_n, gh_r, gh_z, gh_n, hpm, mask_next, H_, mp, r, z, n, hn, h, hpm_next, h_last, s):
        v = lambda p: arena.view(p, mp * H).view(mp, H)  # noqa: E731
        rr = torch.sigmoid(v(gi_r) + v(gh_r))
        zz = torch.sigmoid(v(gi_z) + v(gh_z))
        nn_ = torch.tanh(v(gi_n) + rr * v(gh_n))
        hh = (1 - zz) * nn_ + zz * v(hpm)
        v(h).copy_(hh)
        if r is not None:
            v(r).copy_(rr); v(z).copy_(zz); v(n).copy_(nn_); v(hn).copy_(v(gh_n))
        if hpm_next is not None:
            v(hpm_next).copy_(hh * arena.view(mask_next, mp).view(mp, 1))
        if h_last is not None:
            v(h_last).copy_(hh)

    def cell_bwd(dh_out, t_r, t_z, t_n, mask_next, r, z, n, hn, hpm, H_, mp, gz, dr, dz, dn, dhn, s):
        v = lambda p: arena.view(p, mp * H).view(mp, H)  # noqa: E731
        G = v(dh_out).clone()
        if t_r is not None:
            G += arena.view(mask_next, mp).view(mp, 1) * (v(gz) + v(t_r) + v(t_z) + v(t_n))
        rr, zz, nn_, hn_, hp = v(r), v(z), v(n), v(hn), v(hpm)
        dpre_n = G * (1 - zz) * (1 - nn_ * nn_)
        v(dz).copy_(G * (hp - nn_) * zz * (1 - zz))
        v(dn).copy_(dpre_n)
        v(dr).copy_(dpre_n * hn_ * rr * (1 - rr))
        v(dhn).copy_(dpre_n * rr)
        v(gz).copy_(G * zz)

    def rownorm(x, M, H_, y, rstd, s):
        xx = arena.view(x, M * H).view(M, H)
        mu = xx.mean(-1, keepdim=True)
        rs = 1.0 / torch.sqrt(((xx - mu) ** 2).mean(-1, keepdim=True) + 1e-5)
        arena.view(y, M * H).view(M, H).copy_((xx - mu) * rs)
        arena.view(rstd, M).copy_(rs[:, 0])

    def bwd_dx(dz, xprev, mask_prev, rstd_prev, M, HO, HI, Wp, dz_prev, x0n, kp0, dw_part, n_wg, s):
        # emulated WITHOUT the LayerNorm / ReLU backward of the MLP layer (identity there): d x_hat = dz W
        d = arena.view(dz, M * HO).view(M, HO)
        arena.view(dz_prev, M * HI).view(M, HI).copy_(d @ arena.view(Wp, HO * HI).view(HO, HI))

    def cell_tangent(gia_r, gia_z, gia_n, gib_r, gib_z, gib_n, gha_r, gha_z, gha_n, ghb_r, ghb_z, ghb_n, r, z, n, hn, hpm,
                     hpm_dot, mask_next, H_, mp, h_dot, hpm_dot_next, s):
        v = lambda p: arena.view(p, mp * H).view(mp, H)  # noqa: E731
        o = lambda p: 0.0 if p is None else v(p)  # noqa: E731
        rr, zz, nn_ = v(r), v(z), v(n)
        rd = rr * (1 - rr) * (v(gia_r) + v(gib_r) + o(gha_r) + v(ghb_r))
        zd = zz * (1 - zz) * (v(gia_z) + v(gib_z) + o(gha_z) + v(ghb_z))
        nd = (1 - nn_ * nn_) * (v(gia_n) + v(gib_n) + rd * v(hn) + rr * (o(gha_n) + v(ghb_n)))
        hd = (1 - zz) * nd + zd * (v(hpm) - nn_) + zz * o(hpm_dot)
        v(h_dot).copy_(hd)
        if hpm_dot_next is not None:
            v(hpm_dot_next).copy_(hd * arena.view(mask_next, mp).view(mp, 1))

    def ln_tangent(zd1, zd2, xhat, mean, rstd, M, H_, act, out, s):
        assert act == 0 and zd2 is None and mean is None
        d, x = arena.view(zd1, M * H).view(M, H), arena.view(xhat, M * H).view(M, H)
        rs = arena.view(rstd, M).view(M, 1)
        arena.view(out, M * H).view(M, H).copy_(rs * (d - d.mean(-1, keepdim=True) - x * (x * d).mean(-1, keepdim=True)))

    return dict(harl_mlp_linear=linear, harl_mlp_linear3=linear3, harl_gru_cell_init=cell_init, harl_gru_cell_fwd=cell_fwd, harl_gru_cell_bwd=cell_bwd,
                harl_rownorm=rownorm, harl_mlp_bwd_dx=bwd_dx, harl_gru_cell_tangent=cell_tangent, harl_act_ln_tangent=ln_tangent)


@pytest.mark.parametrize("H,L,m,RN", [(128, 7, 40, 1), (64, 5, 32, 1), (64, 6, 40, 2), (128, 4, 32, 3)])
def test_composition_matches_autograd_gru(stub_kernels, monkeypatch, H, L, m, RN):  # noqa: F811
    """RN stacked layers (rnn.py:14: nn.GRU(num_layers=recurrent_n)); every layer's carried state is multiplied by the mask of
    the step (rnn.py:27,67)."""
    from harl_amd import _lib, gru_wide

    torch.manual_seed(3)
    mp = ((m + 31) // 32) * 32
    M = L * mp
    f = lambda *s_: torch.randn(*s_, dtype=torch.float64)  # noqa: E731  (fp64: the comparison is about logic, not rounding)

    class Net:  # the attributes gru_wide.py reads from a _FlatNet
        hidden_sizes = [H]
        device_ = torch.device("cpu")
        recurrent_n = RN
    net = Net()
    z = lambda n_: torch.zeros(n_, dtype=torch.float64)  # noqa: E731
    net.gru_packs = [dict(Wih=0.3 * f(3 * H * H), bih=0.1 * f(3 * H), Whh=0.3 * f(3 * H * H), bhh=0.1 * f(3 * H))
                     for _ in range(RN)]
    net.xh, net.rmask, net.rstd = [f(M * H)], [torch.zeros(8, dtype=torch.int32)], [z(M)]
    net.rnn_saved_l = [[z(M * H) for _ in range(5)] for _ in range(RN)]
    net.rnn_dgate_l = [[z(M * H) for _ in range(4)] for _ in range(RN)]
    net.rnn_gi, net.rnn_y, net.rnn_rstd = z(3 * M * H), z(M * H), z(M)
    net.dz = [f(M * H), z(M * H)]
    net.rnn_hraw_l = [z(M * H) for _ in range(RN)]
    net.rnn_gh, net.rnn_gz, net.rnn_zero_bias, net.rnn_tmp, net.rnn_dh = z(3 * M * H), z(M * H), z(H), z(M * H), z(M * H)
    mask_rows = (torch.rand(M, dtype=torch.float64) > 0.2).to(torch.float64)
    seq = dict(L=L, m_pad=mp, m=m, h0=0.5 * f(mp, RN * H), mask_rows=mask_rows, h_last=z(mp * RN * H).view(mp, RN * H))

    arena = _Arena()
    for t in (*[v for gp in net.gru_packs for v in gp.values()], net.xh[0], net.rstd[0],
              *[t_ for l_ in net.rnn_saved_l for t_ in l_], *[t_ for l_ in net.rnn_dgate_l for t_ in l_],
              net.rnn_gi, net.rnn_y, net.rnn_rstd, *net.dz, *net.rnn_hraw_l, net.rnn_gh, net.rnn_gz, net.rnn_zero_bias,
              net.rnn_tmp, net.rnn_dh, mask_rows, seq["h0"], seq["h_last"]):
        arena.add(t)
    emu = _emulate(arena, H)
    keep = []  # tensors created inside gru_wide (transposed weight blocks, per-layer state slices) must stay alive and addressable

    real_ptr = _lib.ptr

    def ptr(t):
        if t is None:
            return None
        if not any(lo <= t.data_ptr() < hi for lo, hi, _ in arena.regs):
            keep.append(t)
            arena.add(t if t.is_contiguous() else t.contiguous())
        return real_ptr(t)

    def call(name, *args, tag=None):
        emu[name](*args)

    monkeypatch.setattr(gru_wide, "call", call)
    monkeypatch.setattr(gru_wide, "ptr", ptr)
    monkeypatch.setattr(gru_wide, "stream", lambda: 0)

    # ---- reference: plain torch GRU with autograd (torch.nn.GRU's equations, gate order r, z, n; masks reset the carried state)
    x = net.xh[0].view(L, mp, H).clone().requires_grad_(True)
    Wi = [gp["Wih"].view(3, H, H).clone().requires_grad_(True) for gp in net.gru_packs]
    Wh = [gp["Whh"].view(3, H, H).clone().requires_grad_(True) for gp in net.gru_packs]
    h = [seq["h0"].view(mp, RN, H)[:, k] for k in range(RN)]
    hs = []
    for l in range(L):
        inp = x[l]
        for k in range(RN):
            gp = net.gru_packs[k]
            ht = h[k] * mask_rows[l * mp:(l + 1) * mp].view(mp, 1)
            gi = [inp @ Wi[k][g].t() + gp["bih"][g * H:(g + 1) * H] for g in range(3)]
            gh = [ht @ Wh[k][g].t() + gp["bhh"][g * H:(g + 1) * H] for g in range(3)]
            r_ = torch.sigmoid(gi[0] + gh[0])
            z_ = torch.sigmoid(gi[1] + gh[1])
            n_ = torch.tanh(gi[2] + r_ * gh[2])
            h[k] = (1 - z_) * n_ + z_ * ht
            inp = h[k]
        hs.append(inp)
    hraw = torch.stack(hs)                                   # [L, mp, H]: the top layer
    mu = hraw.mean(-1, keepdim=True)
    y_ref = (hraw - mu) / torch.sqrt(((hraw - mu) ** 2).mean(-1, keepdim=True) + 1e-5)

    gru_wide.forward(net, seq, save=True)
    assert torch.allclose(net.rnn_y.view(L, mp, H), y_ref.detach(), atol=1e-12)
    assert torch.allclose(seq["h_last"].view(mp, RN, H), torch.stack([hk.detach() for hk in h], 1), atol=1e-12)

    G = net.dz[0].view(L, mp, H).clone()                     # d(loss)/d(h_l) of the top layer
    (hraw * G).sum().backward()
    gru_wide.backward(net, seq)
    assert torch.allclose(net.dz[1].view(L, mp, H), x.grad, atol=1e-9)                # gradient into the MLP output
    # weight gradients as backward_trunk forms them from the gate gradients: dW_ig = dgi_g^T (layer input), dW_hg = dgh_g^T h~
    for k in range(RN):
        xin = (net.xh[0] if k == 0 else net.rnn_hraw_l[k - 1]).view(M, H)
        hpm = net.rnn_saved_l[k][0].view(M, H)
        for g, (gi_g, gh_g) in enumerate(zip((0, 1, 2), (0, 1, 3))):
            dWi = net.rnn_dgate_l[k][gi_g].view(M, H).t() @ xin
            dWh = net.rnn_dgate_l[k][gh_g].view(M, H).t() @ hpm
            assert torch.allclose(dWi, Wi[k].grad[g], atol=1e-8), (k, g)
            assert torch.allclose(dWh, Wh[k].grad[g], atol=1e-8), (k, g)


@pytest.mark.parametrize("H,L,m,RN", [(128, 5, 40, 1), (64, 4, 32, 2)])
def test_tangent_composition_matches_jvp(stub_kernels, monkeypatch, H, L, m, RN):  # noqa: F811
    """gru_wide.tangent (HATRPO's forward-mode pass through the composed GRU, incl. stacked layers) against
    torch.autograd.functional.jvp through a plain torch GRU + LayerNorm, entry points emulated in float64."""
    from harl_amd import _lib, gru_wide

    torch.manual_seed(5)
    mp = ((m + 31) // 32) * 32
    M = L * mp
    f = lambda *s_: torch.randn(*s_, dtype=torch.float64)  # noqa: E731
    z = lambda n_: torch.zeros(n_, dtype=torch.float64)  # noqa: E731
    per = 2 * (3 * H * H + 3 * H)

    class Net:
        hidden_sizes = [H]
        device_ = torch.device("cpu")
        recurrent_n = RN
        _gru_pack_base = 7  # (an arbitrary offset: the blocks are addressed relative to it)
    net = Net()
    net.pack_arena = torch.cat([z(7), 0.3 * f(RN * per)])
    pack_d = torch.cat([z(7), 0.2 * f(RN * per)])

    def views(arena_):
        out = []
        for k in range(RN):
            b0, n3 = 7 + k * per, 3 * H * H
            out.append(dict(Wih=arena_[b0:b0 + n3], bih=arena_[b0 + n3:b0 + n3 + 3 * H],
                            Whh=arena_[b0 + n3 + 3 * H:b0 + 2 * n3 + 3 * H], bhh=arena_[b0 + 2 * n3 + 3 * H:b0 + 2 * n3 + 6 * H]))
        return out
    net.gru_packs = views(net.pack_arena)
    tan = views(pack_d)
    net.xh, net.rmask, net.rstd = [f(M * H)], [torch.zeros(8, dtype=torch.int32)], [z(M)]
    xdot = f(M * H)
    net.rnn_saved_l = [[z(M * H) for _ in range(5)] for _ in range(RN)]
    net.rnn_dgate_l = [[z(M * H) for _ in range(4)] for _ in range(RN)]
    net.rnn_gi, net.rnn_y, net.rnn_rstd = z(3 * M * H), z(M * H), z(M)
    net.dz = [f(M * H), z(M * H)]
    net.rnn_hraw_l = [z(M * H) for _ in range(RN)]
    net.rnn_gh, net.rnn_gz, net.rnn_zero_bias, net.rnn_tmp, net.rnn_dh = z(3 * M * H), z(M * H), z(H), z(M * H), z(M * H)
    mask_rows = (torch.rand(M, dtype=torch.float64) > 0.25).to(torch.float64)
    seq = dict(L=L, m_pad=mp, m=m, h0=0.5 * f(mp, RN * H), mask_rows=mask_rows, h_last=None)

    arena = _Arena()
    for t in (net.pack_arena, pack_d, net.xh[0], xdot, net.rstd[0], *[t_ for l_ in net.rnn_saved_l for t_ in l_], net.rnn_gi, net.rnn_y,
              net.rnn_rstd, *net.rnn_hraw_l, net.rnn_gh, net.rnn_gz, net.rnn_zero_bias, net.rnn_tmp, mask_rows, seq["h0"]):
        arena.add(t)
    emu = _emulate(arena, H)
    keep = []
    real_ptr = _lib.ptr

    def ptr(t):
        if t is None:
            return None
        if not any(lo <= t.data_ptr() < hi for lo, hi, _ in arena.regs):
            base = t._base if t._base is not None else t  # (a slice seen first must not shadow the workspace it belongs to)
            keep.append(base)
            arena.add(base)
        return real_ptr(t)

    monkeypatch.setattr(gru_wide, "call", lambda name, *args, tag=None: emu[name](*args))
    monkeypatch.setattr(gru_wide, "ptr", ptr)
    monkeypatch.setattr(gru_wide, "stream", lambda: 0)
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{**k, "dtype": torch.float64}) if k.get("dtype") == torch.float32 else real_empty(*a, **k))

    def fwd(x, *ws_):  # the reference function: stacked GRU with mask resets on every layer's state, then LayerNorm without affine
        h = [seq["h0"].view(mp, RN, H)[:, k] for k in range(RN)]
        ys = []
        for l in range(L):
            inp = x.view(L, mp, H)[l]
            for k in range(RN):
                Wi, bi, Wh, bh = ws_[4 * k:4 * k + 4]
                ht = h[k] * mask_rows[l * mp:(l + 1) * mp].view(mp, 1)
                gi = [inp @ Wi.view(3, H, H)[g].t() + bi[g * H:(g + 1) * H] for g in range(3)]
                gh = [ht @ Wh.view(3, H, H)[g].t() + bh[g * H:(g + 1) * H] for g in range(3)]
                r_ = torch.sigmoid(gi[0] + gh[0])
                z_ = torch.sigmoid(gi[1] + gh[1])
                n_ = torch.tanh(gi[2] + r_ * gh[2])
                h[k] = (1 - z_) * n_ + z_ * ht
                inp = h[k]
            ys.append(inp)
        hraw = torch.stack(ys)
        mu = hraw.mean(-1, keepdim=True)
        return (hraw - mu) / torch.sqrt(((hraw - mu) ** 2).mean(-1, keepdim=True) + 1e-5)

    prim = (net.xh[0].clone(),) + tuple(gp[k_].clone() for gp in net.gru_packs for k_ in ("Wih", "bih", "Whh", "bhh"))
    tang = (xdot.clone(),) + tuple(tp[k_].clone() for tp in tan for k_ in ("Wih", "bih", "Whh", "bhh"))
    y_ref, ydot_ref = torch.autograd.functional.jvp(fwd, prim, tang)

    gru_wide.forward(net, seq, save=True)
    assert torch.allclose(net.rnn_y.view(L, mp, H), y_ref, atol=1e-12)
    ydot = gru_wide.tangent(net, seq, xdot, pack_d, {})
    assert torch.allclose(ydot.view(L, mp, H), ydot_ref, atol=1e-10), float((ydot.view(L, mp, H) - ydot_ref).abs().max())
