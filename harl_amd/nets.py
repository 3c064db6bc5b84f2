"""Device-side actor / critic networks: parameter containers + kernel orchestration.

``StochasticPolicy`` / ``VNet`` are ``nn.Module``s whose ``state_dict()`` has exactly the
reference's keys, shapes and order (harl/models/policy_models/stochastic_policy.py:11-54,
harl/models/value_function_models/v_net.py:10-46), so ``save()/restore()``
checkpoints are interchangeable.  All parameters are views into ONE flat fp32 arena per
network (so grad-norm/clip/Adam are a single fused launch and the data-parallel gradient
all-reduce is a single RCCL call); gradients are views into a second arena of the same layout.

The arithmetic is in libharl_hip.so (include/harl_hip.h); torch is used here only for device
memory and the stream.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import DHEAD_LD, PS_STRIDE, SLAB, call, ptr, stream

SUPPORTED_WIDTHS = (64, 128, 256)
# models_tools.py:28-50; ids of csrc/elementwise.hip (0 = the fused ReLU kernels)
ACT_IDS = {"relu": 0, "leaky_relu": 1, "tanh": 2, "sigmoid": 3, "selu": 4}


_FUSED_UPDATE_MODES = ("hybrid", "logp", "1", "actor", "0")


def _fused_update_mode() -> str:
    """``HARL_FUSED_UPDATE`` (see _FlatNet.fused_update_ok), validated: an unknown value used to behave like "1" silently."""
    mode = os.environ.get("HARL_FUSED_UPDATE", "hybrid")
    if mode not in _FUSED_UPDATE_MODES:
        raise ValueError(f"HARL_FUSED_UPDATE={mode!r}: expected one of {_FUSED_UPDATE_MODES}")
    return mode


BWD_FUSED_MIN_ROWS = 400_000


def _bwd_fused_mode(M: int = 0) -> str:
    """``HARL_BWD_FUSED``: "auto" (default) = 128 x 128 hidden layers take harl_mlp_bwd_dx_dw (dx + the layer's weight gradient +
    the fused first-layer one in ONE launch, operand splits interleaved with the MFMAs: dz and x_hat cross HBM once) when the
    minibatch has at least BWD_FUSED_MIN_ROWS rows, the layer kernels (harl_mlp_dw_partials + harl_mlp_bwd_dx) below that;
    "1" / "0" force one or the other; "nofill" = the one-launch kernel with the splits in separate phases (A/B).
    Measured on MI355X (profiles/r05_bwd_fused_ab.md): 17 % fewer HBM bytes per MPE step at the SAME step time (17.09 ms both),
    -2.5 % on the 6-agent three-layer workload (819 200 rows per launch); at 204 800 rows (6.25 super-rounds per workgroup, the
    weight staging and the pipeline's first round amortised over too few slabs) it is 1.5 % slower -- hence the threshold."""
    m = os.environ.get("HARL_BWD_FUSED", "auto")
    if m not in ("0", "1", "nofill", "auto"):
        raise ValueError(f"HARL_BWD_FUSED={m!r}: expected auto, 0, 1 or nofill")
    if m == "auto":
        return "1" if M >= BWD_FUSED_MIN_ROWS else "0"
    return m


def _bwd_streams() -> bool:
    """``HARL_BWD_STREAMS=1``: the hidden layers' weight-gradient launches go to a second stream next to the backward-dx
    launches (nets.backward_trunk; the C side picks the register-lean dx kernel under the same variable)."""
    return os.environ.get("HARL_BWD_STREAMS", "0") == "1"


def _space_shape(space) -> Tuple[int, ...]:
    """Duck-typed like the reference (harl/utils/envs_tools.py:15-29)."""
    name = space.__class__.__name__
    if name == "Box":
        return tuple(space.shape)
    if name == "list":
        return tuple(space)
    raise NotImplementedError(f"observation space {name}")


class _Node(nn.Module):
    """Empty container used to reproduce the reference's dotted parameter names."""


def _register(root: nn.Module, dotted: str, p: nn.Parameter) -> None:
    parts = dotted.split(".")
    mod = root
    for name in parts[:-1]:
        if name not in mod._modules:
            mod.add_module(name, _Node())
        mod = mod._modules[name]
    mod.register_parameter(parts[-1], p)


class _FlatNet(nn.Module):
    """nn.Module whose parameters are views into one flat arena, plus the MLP trunk kernels."""

    def __init__(self, args: dict, in_dim: int, device: torch.device):
        super().__init__()
        _lib.require_gpu(device)
        self.device_ = device
        self.hidden_sizes = list(args["hidden_sizes"])
        self.use_feature_normalization = bool(args["use_feature_normalization"])
        # activation (models_tools.py:28-50).  ReLU runs on the fused kernels (one mask bit per element); leaky_relu / tanh /
        # sigmoid / selu -- the other functions nn.init.calculate_gain accepts, i.e. the ones the reference's MLPLayer can be
        # built with (mlp.py:19-23; "hardswish" and "identity" fail there) -- take a composed coverage path: the GEMM kernels in
        # raw mode + element-wise activation / LayerNorm launches (harl_act_ln_fwd, harl_act_bwd; see forward_trunk)
        self.activation_func = args.get("activation_func", "relu")
        if self.activation_func not in ACT_IDS:
            raise NotImplementedError(f"activation_func {self.activation_func!r}: supported are {sorted(ACT_IDS)} (the functions "
                                      "torch.nn.init.calculate_gain knows, as in the reference's MLPLayer)")
        self.act_id = ACT_IDS[self.activation_func]
        self.recurrent = bool(args.get("use_recurrent_policy", False) or args.get("use_naive_recurrent_policy", False))
        self.recurrent_n = int(args.get("recurrent_n", 1))
        # a 128-wide GRU is composed from layer GEMMs + element-wise cell kernels (harl_amd/gru_wide.py; parity-green on
        # hardware since round 3: goldens rnn_box_h128, rnn_disc_h128_mb2).  HARL_GRU128=0 restores the refusal.
        # Stacked layers (recurrent_n > 1, models/base/rnn.py:14) run on the same composition, layer after layer, for 64- and
        # 128-wide GRUs (round 4: goldens rnn2_box_h64, rnn2_disc_h128_naive_mb2); the fused kernels of csrc/gru.hip are
        # single-layer.
        hl = self.hidden_sizes[-1]
        self.gru_wide = self.recurrent and ((hl == 128 and os.environ.get("HARL_GRU128", "1") != "0")
                                            or (hl in (64, 128) and self.recurrent_n > 1))
        # HARL_GRU_COMPOSED=1 sends 64-wide GRUs through the same composition: a cross-check of gru_wide.py against the
        # fused kernels and their goldens (tests only)
        if self.recurrent and hl == 64 and os.environ.get("HARL_GRU_COMPOSED") == "1":
            self.gru_wide = True
        if self.recurrent and self.recurrent_n < 1:
            raise ValueError("recurrent_n must be >= 1")
        if self.recurrent and not self.gru_wide and (hl != 64 or self.recurrent_n != 1):
            raise NotImplementedError("GRU kernels: hidden width 64 (fused; composed for recurrent_n > 1) or 128 (composed)")
        for h in self.hidden_sizes:
            if h not in SUPPORTED_WIDTHS:
                raise NotImplementedError(f"hidden width {h}: kernels are instantiated for {SUPPORTED_WIDTHS}")
        # width 256 runs on the panel kernels (csrc/panel.hip): every layer 256 wide, feed-forward, inputs up to 512
        self.panel = 256 in self.hidden_sizes
        if self.panel and (any(h != 256 for h in self.hidden_sizes) or self.recurrent or in_dim > 512):
            raise NotImplementedError("hidden width 256: all layers must be 256 wide, feed-forward, inputs <= 512")
        if self.act_id and (self.recurrent or self.panel or in_dim > 512):
            raise NotImplementedError("activation functions other than relu: feed-forward MLPs of width 64 / 128 with inputs <= 512")
        self.in_dim = in_dim
        self.wide = 32 < in_dim <= 512  # first layer through the cached x0n image (csrc/wide.hip); <= 32: fused 2-layer kernel
        self._x0n_key = None
        self.md = False  # MultiDiscrete heads (StochasticPolicy)
        self._md_sp: List[int] = []
        self._cpu_params: List[Tuple[str, torch.Tensor]] = []
        self._build_trunk_params(args)

    # ---- parameter construction: same torch calls in the same order as the reference, so that the
    # global RNG stream (and therefore the initial weights) match for a given seed.
    def _build_trunk_params(self, args: dict) -> None:
        init = getattr(nn.init, args["initialization_method"])
        gain = nn.init.calculate_gain(self.activation_func)  # mlp.py:21
        d = self.in_dim
        if self.use_feature_normalization:  # MLPBase.feature_norm (mlp.py:57-58)
            self._cpu_params += [("base.feature_norm.weight", torch.ones(d)), ("base.feature_norm.bias", torch.zeros(d))]
        for i, h in enumerate(self.hidden_sizes):  # MLPLayer (mlp.py:25-38): [Linear, act, LayerNorm] x k
            lin = nn.Linear(d, h)
            init(lin.weight.data, gain=gain)
            nn.init.constant_(lin.bias.data, 0)
            self._cpu_params += [(f"base.mlp.fc.{3*i}.weight", lin.weight.data), (f"base.mlp.fc.{3*i}.bias", lin.bias.data),
                                 (f"base.mlp.fc.{3*i+2}.weight", torch.ones(h)), (f"base.mlp.fc.{3*i+2}.bias", torch.zeros(h))]
            d = h
        if self.recurrent:  # RNNLayer (rnn.py:8-21): nn.GRU default init, then bias = 0 / weight = init_method, then LayerNorm
            gru = nn.GRU(d, d, num_layers=self.recurrent_n)
            for name, param in gru.named_parameters():
                if "bias" in name:
                    nn.init.constant_(param, 0)
                elif "weight" in name:
                    init(param)
            sd = dict(gru.named_parameters())
            for l in range(self.recurrent_n):  # nn.GRU's registration order: per layer weight_ih, weight_hh, bias_ih, bias_hh
                self._cpu_params += [(f"rnn.rnn.{k}", sd[k].data) for k in
                                     (f"weight_ih_l{l}", f"weight_hh_l{l}", f"bias_ih_l{l}", f"bias_hh_l{l}")]
            self._cpu_params += [("rnn.norm.weight", torch.ones(d)), ("rnn.norm.bias", torch.zeros(d))]

    def _finalize_params(self) -> None:
        """Move the collected CPU tensors into the flat device arena and register views as nn.Parameters."""
        dev = self.device_
        total = sum(t.numel() for _, t in self._cpu_params)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        hidden = getattr(self, "_hidden_blocks", ())
        for name, t in self._cpu_params:
            n = t.numel()
            view = self.flat_param[off:off + n].view(t.shape)
            view.copy_(t.to(dev))
            if name not in hidden:
                p = nn.Parameter(view, requires_grad=True)
                p.grad = self.flat_grad[off:off + n].view(t.shape)
                _register(self, name, p)
            self.offsets[name] = (off, tuple(t.shape))
            off += n
        # parameters that are row ranges of a hidden block (the MultiDiscrete heads: one contiguous [sum n_k, H] matrix per
        # group in the arena, registered under the reference's per-head names and in its order)
        for name, base, lo, hi in getattr(self, "_alias_params", ()):
            boff, bshape = self.offsets[base]
            rowlen = math.prod(bshape[1:])
            o, shape = boff + lo * rowlen, (hi - lo,) + tuple(bshape[1:])
            n = math.prod(shape)
            p = nn.Parameter(self.flat_param[o:o + n].view(shape), requires_grad=True)
            p.grad = self.flat_grad[o:o + n].view(shape)
            _register(self, name, p)
            self.offsets[name] = (o, shape)
        self.n_params = total
        del self._cpu_params
        # folded weights (W' = W diag(gamma_prev), b' = b + W beta_prev), one pack per Linear incl. the head
        self._packs: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self._max_rows = 0

    def flat_reference(self) -> torch.Tensor:
        """All parameters as one vector in the REFERENCE's ``parameters()`` order.  Equal to ``flat_param`` except for
        MultiDiscrete policies, whose heads are stored group-contiguously ([W_0; W_1; ..][b_0; b_1; ..]) in the arena."""
        return torch.cat([p.detach().reshape(-1) for p in self.parameters()])

    def pview(self, name: str) -> torch.Tensor:
        off, shape = self.offsets[name]
        return self.flat_param[off:off + math.prod(shape)].view(shape)

    def gview(self, name: str) -> torch.Tensor:
        off, shape = self.offsets[name]
        return self.flat_grad[off:off + math.prod(shape)].view(shape)

    # ---- layer table: (W name, b name, gamma name | None, beta name | None, out, in)
    def _layers(self) -> List[Tuple[str, str, Optional[str], Optional[str], int, int]]:
        out = []
        d = self.in_dim
        g = ("base.feature_norm.weight", "base.feature_norm.bias") if self.use_feature_normalization else (None, None)
        for i, h in enumerate(self.hidden_sizes):
            out.append((f"base.mlp.fc.{3*i}.weight", f"base.mlp.fc.{3*i}.bias", g[0], g[1], h, d))
            g = (f"base.mlp.fc.{3*i+2}.weight", f"base.mlp.fc.{3*i+2}.bias")
            d = h
        if self.recurrent:
            g = ("rnn.norm.weight", "rnn.norm.bias")
        for hw, hb, hdim in self._head_layers():
            out.append((hw, hb, g[0], g[1], hdim, d))
        return out

    def _head_layers(self) -> List[Tuple[str, str, int]]:
        """The Linear(s) on top of the trunk: one head, or one per MultiDiscrete group (StochasticPolicy)."""
        return [self._head_names()]

    def _entries(self) -> List[Tuple[int, int, int, int, int, int]]:
        """Every Linear-like block in TABLE order as (w_off, b_off, gamma_off, beta_off, out, in) offsets into the flat
        arena: MLP layers, [GRU W_ih gate blocks r,z,n (folded with the last MLP LayerNorm), GRU W_hh gate blocks], head."""
        off = lambda n: self.offsets[n][0] if n else -1  # noqa: E731
        layers = self._layers()
        nh = len(self._head_layers())
        ents = [(off(w), off(b), off(g), off(be), o, k) for (w, b, g, be, o, k) in layers[:-nh]]
        if self.recurrent:
            H = self.hidden_sizes[-1]
            i = len(self.hidden_sizes) - 1
            for l in range(self.recurrent_n):
                # layer 0 reads the last MLP layer's LayerNorm output (folded into W_ih_l0); layer l > 0 the raw output of layer l - 1
                g, be = (off(f"base.mlp.fc.{3*i+2}.weight"), off(f"base.mlp.fc.{3*i+2}.bias")) if l == 0 else (-1, -1)
                for gate in range(3):
                    ents.append((off(f"rnn.rnn.weight_ih_l{l}") + gate * H * H, off(f"rnn.rnn.bias_ih_l{l}") + gate * H, g, be, H, H))
                for gate in range(3):
                    ents.append((off(f"rnn.rnn.weight_hh_l{l}") + gate * H * H, off(f"rnn.rnn.bias_hh_l{l}") + gate * H, -1, -1, H, H))
        for (w, b, g, be, o, k) in layers[-nh:]:
            ents.append((off(w), off(b), off(g), off(be), o, k))
        return ents

    def _head_names(self) -> Tuple[str, str, int]:
        raise NotImplementedError

    def _build_tables(self) -> None:
        """Folded-weight arena, dense folded-gradient arena and the device layer table (include/harl_hip.h,
        HARL_TABLE_STRIDE) used by harl_reduce_partials_multi / harl_adam_fold."""
        ents = self._entries()
        dev = self.device_
        L = len(self.hidden_sizes)
        rows, pack_off, dwp_off = [], 0, 0
        pack_slots = []
        self._dwp_offs, self._elems = [], []
        if self.recurrent:  # GRU: the three gate blocks of each matrix must be contiguous ([3H][H], then [3H] biases)
            H = self.hidden_sizes[-1]
        for ei, (wo, bo, go, beo, o, k) in enumerate(ents):
            gru_i = ei - L if (self.recurrent and L <= ei < L + 6 * self.recurrent_n) else -1
            if gru_i >= 0:
                layer, within = divmod(gru_i, 6)
                mat, gate = divmod(within, 3)
                base = self._gru_pack_base + (2 * layer + mat) * (3 * H * H + 3 * H)
                pw, pb = base + gate * H * H, base + 3 * H * H + gate * H
            else:
                if self.recurrent and ei == L:
                    pass
                # MultiDiscrete group: the GEMM kernels read a full [sp][k] matrix (zero rows past the last head)
                orows = self._md_sp[ei - (len(ents) - len(self._md_sp))] if (self.md and ei >= len(ents) - len(self._md_sp)) else o
                pw, pb = pack_off, pack_off + orows * k
                pack_off += orows * k + orows
                if self.recurrent and ei == L - 1:  # reserve the GRU block right after the last MLP layer
                    self._gru_pack_base = pack_off
                    pack_off += 2 * self.recurrent_n * (3 * H * H + 3 * H)
            kp, op = ((k + 31) // 32) * 32, ((o + 31) // 32) * 32
            if self.md and ei >= len(ents) - len(self._md_sp):
                op = self._md_sp[ei - (len(ents) - len(self._md_sp))]  # partial layout of harl_mlp_dw_partials(HO = sp)
            elems = op * kp + op
            rows.append([wo, bo, go, beo, o, k, pw, pb, dwp_off, kp, op, 0])
            pack_slots.append((pw, pb, o, k))
            self._dwp_offs.append(dwp_off)
            self._elems.append(elems)
            dwp_off += elems
        self._table_rows = rows
        self.n_entries = len(rows)
        self.pack_arena = torch.zeros(pack_off, dtype=torch.float32, device=dev)
        # the dense folded gradients live at the head of the data-parallel all-reduce message: [dwp | 4 fixed-grid scalar pieces] (dist.py)
        self.dwp_msg = torch.zeros(dwp_off + 4 * PS_STRIDE, dtype=torch.float32, device=dev)
        self.dwp = self.dwp_msg[:dwp_off]
        self.total_dwp = dwp_off
        self._pack_slots = pack_slots
        views = [(self.pack_arena[pw:pw + o * k], self.pack_arena[pb:pb + o]) for (pw, pb, o, k) in pack_slots]
        self._packs = views[:L] + [views[-1]]  # MLP layers by index, head last (callers use [l] and [-1])
        nh = len(self._head_layers())
        self._head_packs = views[-nh:]          # every head entry (MultiDiscrete: one per group)
        if self.recurrent:
            n = 3 * H * H
            self.gru_packs = []  # per GRU layer: the folded [3H, H] gate matrices and [3H] biases
            for l in range(self.recurrent_n):
                b0 = self._gru_pack_base + 2 * l * (n + 3 * H)
                self.gru_packs.append(dict(Wih=self.pack_arena[b0:b0 + n], bih=self.pack_arena[b0 + n:b0 + n + 3 * H],
                                           Whh=self.pack_arena[b0 + n + 3 * H:b0 + 2 * n + 3 * H],
                                           bhh=self.pack_arena[b0 + 2 * n + 3 * H:b0 + 2 * n + 6 * H]))
            self.gru_pack = self.gru_packs[0]
        # harl_adam_fold updates parameters by table entry (rows of W with their bias, the LayerNorm in front, log_std): every
        # parameter must be reachable that way
        covered = sum(o * k + o for (_, _, _, _, o, k) in ents)
        covered += sum(k for go, k in {(go, k) for (_, _, go, _, _, k) in ents if go >= 0}) * 2
        if isinstance(self, StochasticPolicy) and not self.discrete:
            covered += self.act_dim
        assert covered == self.n_params, f"layer table covers {covered} of {self.n_params} parameters"
        self.table = None  # device table is finalised in _ensure_ws (part offsets depend on n_wg)

    def fold(self) -> None:
        """Recompute the folded weights from the current parameters (after init / load_state_dict; the optimiser
        step re-folds inside harl_adam_fold)."""
        if not self._packs:
            self._build_tables()
        # nothing touched the parameters through torch since the packs were last made consistent with them INSIDE this update
        # (the optimiser kernel updates parameters AND packs together, without going through torch): the ten launches would
        # rewrite the same values -- 200 such launches per update in the 8-agent recurrent configuration.  Every train() entry
        # point drops the marker (invalidate_caches), so writes behind torch's back between updates are still picked up.
        ver = self.flat_param._version
        if getattr(self, "_fold_version", None) == ver and os.environ.get("HARL_ALWAYS_FOLD", "0") != "1":
            return
        s = stream()
        fp, pa = self.flat_param, self.pack_arena
        if self.table is not None and os.environ.get("HARL_FOLD_TABLE", "1") != "0":  # every entry in one launch
            call("harl_fold_table", ptr(fp), ptr(pa), ptr(self.table), self.n_entries, sum(r[4] for r in self._table_rows), s)
            self._fold_version = ver
            return
        for (wo, bo, go, beo, o, k), (pw, pb, _, _) in zip(self._entries(), self._pack_slots):
            call("harl_fold_linear", ptr(fp[wo:]), ptr(fp[bo:]), ptr(fp[go:]) if go >= 0 else None,
                 ptr(fp[beo:]) if beo >= 0 else None, ptr(pa[pw:]), ptr(pa[pb:]), o, k, s)
        self._fold_version = ver

    # ---- workspaces -------------------------------------------------------------------------
    def _ensure_ws(self, M: int) -> None:
        if M <= self._max_rows:
            return
        dev = self.device_
        n_slabs = (M + SLAB - 1) // SLAB
        mp = n_slabs * SLAB
        f32, u32 = torch.float32, torch.int32
        self.xh = [torch.empty(mp * h, dtype=f32, device=dev) for h in self.hidden_sizes]      # x_hat_l, ATL
        self.rmask = [torch.empty(n_slabs * max(h // 64, 1) * 64, dtype=u32, device=dev) for h in self.hidden_sizes]
        self.rstd = [torch.empty(mp, dtype=f32, device=dev) for _ in self.hidden_sizes]
        self.mu0 = torch.empty(mp, dtype=f32, device=dev)
        self.rstd0 = torch.empty(mp, dtype=f32, device=dev)
        # narrow inputs: the forward pass leaves the normalised inputs behind as an ATL image for the dW_1 kernel
        self.kp0 = ((self.in_dim + 31) // 32) * 32
        # inputs wider than 32 (csrc/wide.hip): x0n is the operand of the first-layer GEMM itself
        self.wide = 32 < self.in_dim <= 512
        self._x0n_key = None
        self.x0n = torch.empty(mp * self.kp0, dtype=f32, device=dev) if (self.in_dim <= 64 or self.wide) else None
        self.w1img = (torch.empty(3 * self.hidden_sizes[0] * self.kp0 // 2, dtype=f32, device=dev)
                      if (self.wide or self.act_id) else None)
        hmax = max(self.hidden_sizes)
        if self.act_id:  # composed activation path: raw pre-activations, mean(act(z)) per layer, an all-ones "ReLU mask"
            self.zraw = torch.empty(mp * hmax, dtype=f32, device=dev)
            self.amean = [torch.empty(mp, dtype=f32, device=dev) for _ in self.hidden_sizes]
            self.ones_mask = torch.full((n_slabs * 64 * 2,), -1, dtype=u32, device=dev)
        # ping-pong, ATL; a third one where the one-launch trunk backward keeps every layer's dz for the weight-gradient launch
        self.dz = [torch.empty(mp * hmax, dtype=f32, device=dev) for _ in range(3 if len(self.hidden_sizes) == 3 and self.trunk_fused() else 2)]
        self._trunk_cache = {}
        # head gradients for the separate dW pass: row-major [mp][32], or the ATL(64) image of a 33..64-way Categorical head
        self.wide_head = (not self.md) and self._layers()[-1][4] > 32
        self.dhead = torch.zeros(mp * (64 if self.wide_head else DHEAD_LD), dtype=f32, device=dev)
        # MultiDiscrete: one logits image per group; the loss kernel overwrites it with d(loss)/d(logits)
        self.md_z = [torch.zeros(mp * sp, dtype=f32, device=dev) for sp in self._md_sp]
        n_iter = (n_slabs + 1) // 2
        # rows of the per-workgroup partial arena = grid of the weight-gradient kernels (two workgroups per CU: the measured
        # optimum at 819 200 rows).  HARL_NWG overrides it (A/B measurements).
        # ... and fewer rows for small minibatches: ~10 slabs per workgroup keep 256+ workgroups busy from 2 560 slabs on, below that
        # the per-workgroup partial rows (written by every weight-gradient launch, read back by the combine) cost more than the
        # parallelism returns -- 2 560 slabs (the 8-agent recurrent workload): 256 rows instead of 512
        nwg_env = os.environ.get("HARL_NWG")
        nwg_cap = int(nwg_env) if nwg_env else max(256, min(512, n_slabs // 10))
        # ... and 256 rows when every weight gradient of this network comes out of a one-workgroup-per-CU launch (the fused forward's
        # head gradient, harl_mlp_bwd_dx_dw for the trunk): the other 256 rows would only be cleared by them and read back by
        # the combine (~22 MB each way per optimiser step at the 3-agent headline shapes)
        if (not nwg_env and _bwd_fused_mode(M) == "1" and len(self.hidden_sizes) >= 2 and all(h == 128 for h in self.hidden_sizes)
                and ((self.in_dim + 31) // 32) * 32 == 32 and not (self.recurrent or self.md or self.act_id or self.panel)):
            nwg_cap = 256
        self.n_wg = max(1, min(nwg_cap, n_iter))
        part_off, rows = 0, [list(r) for r in self._table_rows]
        self._part_offs = []
        for r, elems in zip(rows, self._elems):
            r[11] = part_off
            self._part_offs.append(part_off)
            part_off += self.n_wg * elems
        self.part = torch.empty(part_off, dtype=f32, device=dev)
        self.table = torch.tensor(rows, dtype=torch.int32, device=dev).reshape(-1).contiguous()
        if self.recurrent:
            H = self.hidden_sizes[-1]
            a = lambda: torch.empty(mp * H, dtype=f32, device=dev)  # noqa: E731
            self.rnn_y, self.rnn_rstd = a(), torch.empty(mp, dtype=f32, device=dev)
            # per GRU layer: h~ (= h*mask), r, z, n, hn  and  dr, dz, dn, dhn
            self.rnn_saved_l = [[a() for _ in range(5)] for _ in range(self.recurrent_n)]
            self.rnn_dgate_l = [[a() for _ in range(4)] for _ in range(self.recurrent_n)]
            self.rnn_saved, self.rnn_dgate = self.rnn_saved_l[0], self.rnn_dgate_l[0]
            # all-ones "relu mask" for rnn.norm: (H/2 + 31) / 32 words per lane (two at H = 128 -- sized for one, the head kernels
            # read the second word past the end: the first hardware run of the 128-wide GRU, round 3)
            self.rnn_ones = torch.full((n_slabs * 64 * ((H // 2 + 31) // 32),), -1, dtype=u32, device=dev)
            self.rnn_gi = torch.empty(3 * mp * H, dtype=f32, device=dev)  # input half of the gates, all steps (gru.hip)
            if self.gru_wide:
                from . import gru_wide
                gru_wide.ensure_ws(self, mp)
        self.n_head_blocks = max(_lib.load().harl_head_blocks(M), self.n_wg)
        self.part_scalars = torch.zeros(self.n_head_blocks * PS_STRIDE, dtype=f32, device=dev)
        self.scalars = torch.zeros(PS_STRIDE, dtype=torch.float64, device=dev)
        self._max_rows = M

    # ---- trunk forward: X[rows, D] (gathered by idx) -> x_hat_L in self.xh[-1] ------------------
    # head input: (ATL activation, relu mask, rstd, width) -- the last MLP layer, or the GRU's normalised output
    def feat(self):
        if self.recurrent:
            return self.rnn_y, self.rnn_ones, self.rnn_rstd, self.hidden_sizes[-1]
        if self.act_id:  # the loss kernels then leave the LayerNorm backward WITHOUT an activation derivative (backward_trunk)
            return self.xh[-1], self.ones_mask, self.rstd[-1], self.hidden_sizes[-1]
        return self.xh[-1], self.rmask[-1], self.rstd[-1], self.hidden_sizes[-1]

    def forward_rnn(self, seq: dict, save: bool, gates_done: bool = False) -> None:
        """GRU over a recurrent batch (seq: L, m_pad, h0 [m_pad, H], mask_rows [L*m_pad], optional h_last out).
        ``gates_done``: self.rnn_gi already holds the input half of the gates (harl_mlp_fwd_trunk, forward_trunk)."""
        if self.gru_wide:
            from . import gru_wide
            return gru_wide.forward(self, seq, save)
        gp = self.gru_pack
        sv = self.rnn_saved
        call("harl_gru_fwd", ptr(self.xh[-1]), ptr(seq["mask_rows"]), ptr(seq["h0"]), ptr(gp["Wih"]), ptr(gp["bih"]),
             ptr(gp["Whh"]), ptr(gp["bhh"]), self.hidden_sizes[-1], seq["L"], seq["m_pad"], ptr(self.rnn_y),
             ptr(self.rnn_rstd), ptr(sv[0]), ptr(sv[1]), ptr(sv[2]), ptr(sv[3]), ptr(sv[4]), ptr(seq.get("h_last")),
             int(save) | (2 if gates_done else 0), ptr(self.rnn_gi), stream(), tag="gru_fwd")

    # ---- the whole 64-wide trunk in one launch per direction (csrc/trunk.hip, round 6)
    def trunk_fused(self) -> bool:
        """Wide first layer + one or two more 64-wide ReLU layers (the SMAC shapes: obs 128 / 216 -> [64, 64, 64] -> GRU):
        harl_mlp_fwd_trunk / harl_mlp_bwd_trunk / harl_mlp_dw_partials_multi_v replace the layer launches one for one,
        bit-identically.  HARL_TRUNK_FUSED=0 keeps the layer-by-layer launches (A/B, tests/gpu_checks.check_trunk_fused)."""
        hs = self.hidden_sizes
        return (os.environ.get("HARL_TRUNK_FUSED", "1") != "0" and self.wide and not self.act_id and not self.panel
                and 2 <= len(hs) <= 3 and all(h == 64 for h in hs))

    def _trunk_ptrs(self, save: bool, gates: bool):
        """ctypes pointer arrays of harl_mlp_fwd_trunk for the current workspaces (cached: they change with _ensure_ws only)."""
        key = (save, gates, self._max_rows)
        hit = self._trunk_cache.get(key)
        if hit is None:
            import ctypes as C
            L = len(self.hidden_sizes)
            keep = [save or (l == L - 1 and not gates) for l in range(L)]  # activation records a later kernel reads
            vp = lambda ts: (C.c_void_p * len(ts))(*[ptr(t) for t in ts])  # noqa: E731
            hit = self._trunk_cache[key] = (
                vp([self._packs[l][0] for l in range(1, L)]), vp([self._packs[l][1] for l in range(1, L)]),
                vp([self.xh[l] if keep[l] else None for l in range(L)]), vp([self.rmask[l] if keep[l] else None for l in range(L)]),
                vp([self.rstd[l] if keep[l] else None for l in range(L)]))
        return hit

    def _x0n_image(self, X: torch.Tensor, M: int, s, idx: Optional[torch.Tensor] = None) -> None:
        """Normalised inputs of rows X[idx] as the ATL(kp0) image self.x0n (csrc/wide.hip).  It depends on the rows only:
        every epoch / line-search step / log-prob pass over the same (unmodified) tensor reuses it -- torch's version
        counter catches in-place writes; identity row order only; the cache keeps a reference to the source tensor, so its
        storage cannot have been recycled for different data at the same address."""
        # ... a GATHERED image is reused only when the row-index tensor is the very same object, unmodified (the recurrent
        # samplers hand out one table per update when there is a single minibatch, buffers._recurrent_seqs): the cache keeps
        # both tensors alive, so an equal address cannot belong to a recycled allocation with other rows in it
        key = ((X.data_ptr(), X._version, tuple(X.shape), M) if idx is None
               else (X.data_ptr(), X._version, tuple(X.shape), M, idx.data_ptr(), idx._version, idx.numel()))
        if key != self._x0n_key:
            call("harl_mlp_x0n_wide", ptr(X), X.shape[1], ptr(idx), M, self.in_dim, int(self.use_feature_normalization),
                 ptr(self.x0n), ptr(self.mu0), ptr(self.rstd0), s, tag="x0n_wide")
            self._x0n_key, self._x0n_src = key, (X, idx)

    def invalidate_caches(self) -> None:
        """Drop the cached normalised-input image.  The cache key is (data_ptr, torch version counter, shape, rows): it
        sees in-place torch writes but NOT writes through ``.data``, DLPack / NumPy-shared memory or raw-pointer kernels.
        Contract: every ``train()`` entry point (runner, HAPPO/HATRPO/MAPPO, VCritic) calls this first, so an image never
        outlives the update it was built for; callers that overwrite observations behind torch's back inside one update
        (between log-prob passes over the same tensor) must call it themselves."""
        self._x0n_key = None
        self._x0n_src = None
        # ... and the first fold() of every update re-folds unconditionally (see fold(): the version counter does not see
        # parameter writes through ``.data`` either)
        self._fold_version = None

    def forward_trunk(self, X: torch.Tensor, idx: Optional[torch.Tensor], M: int, for_backward: bool = True,
                      seq: Optional[dict] = None, upto: Optional[int] = None) -> None:
        """``upto``: stop after hidden layer ``upto`` (1-based; fused_last_ok: the last layer runs inside the loss launch)."""
        assert X.dim() == 2 and X.shape[1] == self.in_dim and X.is_contiguous()
        rnn_save = for_backward  # inference passes save no GRU internals (and take the latency variant of the kernel)
        if self.recurrent:
            assert seq is not None and seq["L"] * seq["m_pad"] == M, "recurrent nets need the sequence layout"
            for_backward = True  # the fused 2-layer path may skip x_hat_1; keep it simple for recurrent nets
        self._ensure_ws(M)
        s = stream()
        hs = self.hidden_sizes
        first_hidden = 1
        if self.act_id:
            # activation other than ReLU: [Linear (raw GEMM) -> act + LayerNorm (element-wise)] per layer (mlp.py:25-38)
            self._x0n_image(X, M, s, idx)
            for l, h in enumerate(hs):
                Wp, bp = self._packs[l]
                if l == 0:
                    call("harl_mlp_linear_wide", ptr(self.x0n), M, self.kp0, ptr(Wp), self.in_dim, ptr(bp), h, ptr(self.w1img),
                         ptr(self.zraw), s, tag="linear_wide")
                else:
                    call("harl_mlp_linear", ptr(self.xh[l - 1]), M, hs[l - 1], h, ptr(Wp), ptr(bp), ptr(self.zraw), s, tag="linear")
                call("harl_act_ln_fwd", ptr(self.zraw), M, h, self.act_id, ptr(self.xh[l]), ptr(self.amean[l]), ptr(self.rstd[l]),
                     s, tag="act_ln_fwd")
            return
        if self.panel:  # width 256: x0n image -> panel GEMMs (csrc/panel.hip)
            self._x0n_image(X, M, s, idx)
            for l in range(len(hs)):
                Wp, bp = self._packs[l]
                xin, kp, d = (self.x0n, self.kp0, self.in_dim) if l == 0 else (self.xh[l - 1], 256, 256)
                call("harl_mlp_panel_fwd", ptr(xin), M, kp, ptr(Wp), d, ptr(bp), 256, ptr(self.xh[l]), ptr(self.rmask[l]),
                     ptr(self.rstd[l]), s, tag="fwd_panel")
            return
        if upto is None and self.trunk_fused():
            # every layer (+ the input half of a fused GRU's gates) in ONE launch; forward-only passes write nothing but what
            # the next kernel reads (the three gate images, or x_hat_L)
            gates = self.recurrent and not self.gru_wide
            Wp, bp = self._packs[0]
            self._x0n_image(X, M, s, idx)
            pW, pb, pX, pM, pR = self._trunk_ptrs(bool(rnn_save), gates)
            gp = self.gru_pack if gates else None
            call("harl_mlp_fwd_trunk", ptr(self.x0n), M, self.kp0, ptr(Wp), self.in_dim, ptr(bp), 64, ptr(self.w1img),
                 len(hs) - 1, pW, pb, pX, pM, pR, ptr(gp["Wih"]) if gates else None, ptr(gp["bih"]) if gates else None,
                 ptr(gp["bhh"]) if gates else None, ptr(self.rnn_gi) if gates else None, s, tag="fwd_trunk")
            if self.recurrent:
                self.forward_rnn(seq, save=rnn_save, gates_done=gates)
            return
        if len(hs) >= 2 and hs[0] == hs[1] and self.in_dim <= 64 and idx is None:
            # layers 1+2 fused, from the x0n image: the rows are gathered and normalised ONCE per buffer (every epoch, log-prob
            # pass and line-search step over the same unmodified tensor reuses the image)
            (W1, b1), (W2, b2) = self._packs[0], self._packs[1]
            self._x0n_image(X, M, s)
            call("harl_mlp_fwd_fused2x", ptr(self.x0n), M, ptr(W1), self.in_dim, ptr(b1), ptr(W2), ptr(b2), hs[0],
                 int(for_backward), ptr(self.xh[0]), ptr(self.rmask[0]), ptr(self.rstd[0]), ptr(self.xh[1]),
                 ptr(self.rmask[1]), ptr(self.rstd[1]), s, tag="fwd_fused2" if self.kp0 == 32 else "fwd_fused2_k64")
            first_hidden = 2
        elif len(hs) >= 2 and hs[0] == hs[1] and self.in_dim <= 32:
            # layers 1+2 fused: x_hat_1 stays in registers; it is written out only if a backward pass follows
            (W1, b1), (W2, b2) = self._packs[0], self._packs[1]
            self._x0n_key = None  # this kernel writes the gathered minibatch's image into self.x0n
            call("harl_mlp_fwd_fused2", ptr(X), X.shape[1], ptr(idx), M, self.in_dim, ptr(W1), ptr(b1),
                 int(self.use_feature_normalization), ptr(W2), ptr(b2), hs[0], int(for_backward), ptr(self.xh[0]),
                 ptr(self.rmask[0]), ptr(self.rstd[0]), ptr(self.mu0), ptr(self.rstd0), ptr(self.xh[1]),
                 ptr(self.rmask[1]), ptr(self.rstd[1]), ptr(self.x0n) if for_backward else None, s, tag="fwd_fused2")
            first_hidden = 2
        elif self.wide:
            Wp, bp = self._packs[0]
            self._x0n_image(X, M, s, idx)
            call("harl_mlp_fwd_wide", ptr(self.x0n), M, self.kp0, ptr(Wp), self.in_dim, ptr(bp), hs[0], ptr(self.w1img),
                 ptr(self.xh[0]), ptr(self.rmask[0]), ptr(self.rstd[0]), s, tag="fwd_wide")
        else:
            Wp, bp = self._packs[0]
            self._x0n_key = None  # (may write self.x0n)
            call("harl_mlp_fwd_input", ptr(X), X.shape[1], ptr(idx), M, self.in_dim, ptr(Wp), ptr(bp),
                 int(self.use_feature_normalization), hs[0], ptr(self.xh[0]), ptr(self.rmask[0]), ptr(self.rstd[0]),
                 ptr(self.mu0), ptr(self.rstd0), ptr(self.x0n) if for_backward else None, s, tag="fwd_input")
        assert upto is None or upto >= first_hidden
        for l in range(first_hidden, len(hs) if upto is None else upto):
            Wp, bp = self._packs[l]
            call("harl_mlp_fwd_hidden", ptr(self.xh[l - 1]), M, hs[l - 1], hs[l],
                 ptr(Wp), ptr(bp), ptr(self.xh[l]), ptr(self.rmask[l]), ptr(self.rstd[l]), s, tag="fwd_hidden")
        if self.recurrent:
            self.forward_rnn(seq, save=rnn_save)

    # ---- fused optimiser-step path (csrc/update.hip): two equal hidden layers, inputs <= 64 wide, identity row order
    def fused_update_ok(self, idx: Optional[torch.Tensor], seq: Optional[dict] = None, train: bool = True) -> bool:
        """Route this pass through csrc/update.hip?  ``HARL_FUSED_UPDATE``: "hybrid" (default) = forward-only passes
        (log-probs, factor product, values: one launch instead of two, x_hat_2 never written) AND the forward half of the
        optimiser steps -- forward + head + loss + head gradient + dz_2 in one launch that also leaves layer 1's activation
        record in HBM, followed by the layer-by-layer backward (x_hat_2 / mask_2 / rstd_2 never cross HBM: ~3.5 KB per
        sample instead of ~4.5 KB, one launch less); "logp" = forward-only passes only; "1" = optimiser steps in three
        launches with x_hat_1 recomputed in the backward (~1.9 KB per sample -- fewer bytes but more VALU work than the
        layer kernels and slower end to end on MI355X: DESIGN.md section 3); "actor" = "1" for actors only; "0" = never."""
        hs = self.hidden_sizes
        mode = _fused_update_mode()
        if self.act_id:
            return False
        if mode == "0" or (train and mode == "logp") or (train and mode == "actor" and isinstance(self, VNet)):
            return False
        if not (not self.recurrent and not self.md and idx is None and seq is None
                and len(hs) == 2 and hs[0] == hs[1] and hs[0] in (64, 128) and self.in_dim <= 64 and self._layers()[-1][4] <= 8):
            return False
        # ... and the launch must fit the LDS of one workgroup (the ACTOR step of a 128-wide network with 33..64 inputs does not:
        # the layer kernels take it)
        kind = 0 if not train else (2 if isinstance(self, VNet) else 1)
        return bool(_lib.load().harl_update_supported(self.in_dim, hs[0], self._layers()[-1][4], kind))

    def fused_last_ok(self, idx: Optional[torch.Tensor], seq: Optional[dict] = None) -> bool:
        """Optimiser steps of networks the two-layer launch does not cover (three or more hidden layers, wide first layers,
        minibatches): run the LAST hidden layer inside the loss launch (harl_update_last_*: x_hat_L never written)?
        Needs ReLU, equal 64 / 128-wide last two layers, a head of <= 8 outputs, and a layer kernel in front that stops
        before the last layer."""
        hs = self.hidden_sizes
        if _fused_update_mode() != "hybrid" or self.act_id or self.recurrent or self.md or self.panel:
            return False
        if seq is not None or len(hs) < 2 or hs[-1] != hs[-2] or hs[-1] not in (64, 128) or self._layers()[-1][4] > 8:
            return False
        two_fused = hs[0] == hs[1] and ((self.in_dim <= 64 and idx is None) or self.in_dim <= 32)  # forward_trunk's first launch
        return len(hs) >= (3 if two_fused else 2) and bool(_lib.load().harl_update_supported(0, hs[-1], self._layers()[-1][4], 2 if isinstance(self, VNet) else 1))

    def fused_hybrid(self) -> bool:
        """Optimiser steps as fused forward + layer-by-layer backward (see fused_update_ok)?"""
        return _fused_update_mode() == "hybrid"

    def hybrid_outputs(self):
        """(xh1, rmask1, rstd1) arguments of harl_update_fwd_*: layer 1's activation record for the layer-by-layer backward
        (hybrid), or three NULLs (harl_update_bwd recomputes it)."""
        if not self.fused_hybrid():
            return (None, None, None)
        return (ptr(self.xh[0]), ptr(self.rmask[0]), ptr(self.rstd[0]))

    def backward_after_fused(self, X: torch.Tensor, M: int) -> None:
        if self.fused_hybrid():
            self.backward_trunk(X, None, M, head_dw_done=True)
        else:
            self.backward_fused(M)

    def fused_args(self, X: torch.Tensor, M: int):
        """(x0n, M, D, H, W1', b1', W2', b2', Wh', bh') -- the leading arguments of every harl_update_* entry point.
        Builds / reuses the normalised-input image of X."""
        self._ensure_ws(M)
        self._x0n_image(X, M, stream())
        (W1, b1), (W2, b2), (Wh, bh) = self._packs[0], self._packs[1], self._packs[-1]
        return (ptr(self.x0n), M, self.in_dim, self.hidden_sizes[0], ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(Wh), ptr(bh))

    def _bwd_side(self):
        """The second stream of the layer-by-layer backward (one per network: the critic's chain has its own) + its two events."""
        if getattr(self, "_bwd_side_stream", None) is None:
            self._bwd_side_stream = torch.cuda.Stream(device=self.device_)
            self._bwd_ev = [torch.cuda.Event(), torch.cuda.Event()]
        return self._bwd_side_stream

    def _combine_partials(self, s) -> None:
        """The deterministic split-K combine of every layer's per-workgroup weight-gradient partials into self.dwp, one launch.
        (Round 4 measured it as a first phase of harl_adam_fold instead: slower, csrc/elementwise.hip.)"""
        call("harl_reduce_partials_multi", ptr(self.part), ptr(self.table), self.n_entries, self.n_wg, self.total_dwp,
             ptr(self.dwp), s, tag="reduce_partials")

    def backward_fused(self, M: int) -> None:
        """dz_2 (self.dz[0], written by harl_update_fwd_*) + x0n -> dense folded gradients self.dwp (UNSCALED sums);
        the head's partial rows were written by the forward launch."""
        s = stream()
        (W1, b1), (W2, _) = self._packs[0], self._packs[1]
        po = self._part_offs
        call("harl_update_bwd", ptr(self.x0n), ptr(self.dz[0]), M, self.in_dim, self.hidden_sizes[0], ptr(W1), ptr(b1),
             ptr(W2), ptr(self.part[po[0]:]), ptr(self.part[po[1]:]), self.n_wg, s, tag="update_bwd")
        self._combine_partials(s)

    # ---- backward: dz_L (in self.dz[0]) and dhead -> dense folded gradients self.dwp (UNSCALED sums over samples)
    def backward_trunk(self, X: torch.Tensor, idx: Optional[torch.Tensor], M: int, seq: Optional[dict] = None,
                       head_dw_done: bool = False) -> None:
        """``head_dw_done``: the loss kernel already wrote the head's weight-gradient partials (fused path)."""
        s = stream()
        L = len(self.hidden_sizes)
        nwg = self.n_wg
        po = self._part_offs
        hdim = self._layers()[-1][4]
        fx, _, _, fh = self.feat()
        # head: dW_head' = dhead^T x_hat_L   (x_hat_L = GRU output for recurrent nets)
        if self.md:
            self.md_backward(M)
        elif not head_dw_done:
            if self.wide_head:  # ATL(64) image: an ordinary two-operand weight-gradient GEMM (partial layout dWp[64][fh] | dbp[64])
                call("harl_mlp_dw_partials", ptr(self.dhead), 0, 0, 64, ptr(fx), 0, 0, None, None, None, fh, M,
                     ptr(self.part[po[-1]:]), nwg, s, tag="dw_head")
            elif self.panel:  # 256-wide trunk (HATRPO's separate head-gradient pass; HAPPO's loss kernel fuses it)
                call("harl_head_dw_rows256", ptr(self.dhead), M, hdim, ptr(fx), ptr(self.part[po[-1]:]), nwg, s, tag="dw_head")
            else:
                call("harl_mlp_dw_partials", ptr(self.dhead), 1, DHEAD_LD, hdim, ptr(fx), 0, 0, None, None, None, fh, M,
                     ptr(self.part[po[-1]:]), nwg, s, tag="dw_head")
        if self.trunk_fused() and not self.gru_wide:
            return self._backward_trunk_fused(M, seq, s)
        cur = 0  # self.dz[cur] holds dz of the last MLP layer (non-recurrent) / d(loss)/d(h) (recurrent)
        if self.recurrent:
            gp, sv, dg = self.gru_pack, self.rnn_saved, self.rnn_dgate
            H = self.hidden_sizes[-1]
            if self.gru_wide:
                from . import gru_wide
                gru_wide.backward(self, seq)
            else:
                call("harl_gru_bwd", ptr(self.dz[0]), ptr(seq["mask_rows"]), ptr(gp["Wih"]), ptr(gp["Whh"]), ptr(sv[0]),
                     ptr(sv[1]), ptr(sv[2]), ptr(sv[3]), ptr(sv[4]), H, seq["L"], seq["m_pad"], ptr(self.xh[-1]),
                     ptr(self.rmask[-1]), ptr(self.rstd[-1]), ptr(dg[0]), ptr(dg[1]), ptr(dg[2]), ptr(dg[3]), ptr(self.dz[1]), s,
                     tag="gru_bwd")
            # the six gate blocks in ONE launch: W_ih' blocks d gi_g^T x_hat_mlp (g = r, z, n), W_hh blocks d gh_g^T h~ (dhn for n)
            import ctypes as C
            for gl in range(self.recurrent_n):  # layer gl reads x_hat of the MLP (gl = 0) or the raw output of layer gl - 1
                dg_, sv_ = self.rnn_dgate_l[gl], self.rnn_saved_l[gl]
                xin = self.xh[-1] if gl == 0 else self.rnn_hraw_l[gl - 1]
                a6 = (C.c_void_p * 6)(*[ptr(t) for t in (dg_[0], dg_[1], dg_[2], dg_[0], dg_[1], dg_[3])])
                b6 = (C.c_void_p * 6)(*([ptr(xin)] * 3 + [ptr(sv_[0])] * 3))
                p6 = (C.c_void_p * 6)(*[ptr(self.part[po[L + 6 * gl + k]:]) for k in range(6)])
                call("harl_mlp_dw_partials_multi", 6, a6, b6, p6, H, H, M, nwg, s, tag="dw_gru")
            cur = 1
        if self.panel:  # width 256 (csrc/panel.hip): per layer dW = dz^T x_hat_prev, then dz_prev through the panel GEMM
            for l in range(L - 1, -1, -1):
                b_in, k_in = (self.x0n, self.kp0) if l == 0 else (self.xh[l - 1], 256)
                call("harl_mlp_dw_partials", ptr(self.dz[cur]), 0, 0, 256, ptr(b_in), 0, 0, None, None, None, k_in, M,
                     ptr(self.part[po[l]:]), nwg, s, tag="dw_hidden" if l else "dw_input")
                if l > 0:
                    Wp, _ = self._packs[l]
                    call("harl_mlp_panel_bwd", ptr(self.dz[cur]), ptr(self.xh[l - 1]), ptr(self.rmask[l - 1]),
                         ptr(self.rstd[l - 1]), M, 256, 256, ptr(Wp), ptr(self.dz[1 - cur]), s, tag="bwd_panel")
                    cur = 1 - cur
            self._combine_partials(s)
            return
        if self.act_id:
            # activation other than ReLU: the loss kernel / harl_mlp_bwd_dx were handed an all-ones mask, so self.dz holds the
            # LayerNorm backward d(loss)/d(act(z)); harl_act_bwd multiplies by act'(z) in place (csrc/elementwise.hip)
            hs = self.hidden_sizes
            call("harl_act_bwd", ptr(self.dz[cur]), ptr(self.xh[L - 1]), ptr(self.amean[L - 1]), ptr(self.rstd[L - 1]), M, hs[L - 1],
                 self.act_id, s, tag="act_bwd")
            for l in range(L - 1, 0, -1):
                ho, hi = hs[l], hs[l - 1]
                call("harl_mlp_dw_partials", ptr(self.dz[cur]), 0, 0, ho, ptr(self.xh[l - 1]), 0, 0, None, None, None, hi, M,
                     ptr(self.part[po[l]:]), nwg, s, tag="dw_hidden")
                Wp, _ = self._packs[l]
                call("harl_mlp_bwd_dx", ptr(self.dz[cur]), ptr(self.xh[l - 1]), ptr(self.ones_mask), ptr(self.rstd[l - 1]), M, ho,
                     hi, ptr(Wp), ptr(self.dz[1 - cur]), None, 0, None, 0, s, tag="bwd_dx")
                cur = 1 - cur
                call("harl_act_bwd", ptr(self.dz[cur]), ptr(self.xh[l - 1]), ptr(self.amean[l - 1]), ptr(self.rstd[l - 1]), M, hi,
                     self.act_id, s, tag="act_bwd")
            call("harl_mlp_dw_partials", ptr(self.dz[cur]), 0, 0, hs[0], ptr(self.x0n), 0, 0, None, None, None, self.kp0, M,
                 ptr(self.part[po[0]:]), nwg, s, tag="dw_input")
            self._combine_partials(s)
            return
        # first-layer weight gradient fused into the last bwd_dx (needs the ones column of x0n: in_dim < kp0)
        fuse_dw1 = L >= 2 and self.x0n is not None and self.in_dim < self.kp0 and self.kp0 <= 64
        bwd_mode = _bwd_fused_mode(M)
        side_pending = False
        if (bwd_mode != "0" and fuse_dw1 and self.kp0 == 64 and L >= 2 and self.hidden_sizes[0] == 128 and self.hidden_sizes[1] == 128
                and os.environ.get("HARL_BWD_K64", "0") == "1"):
            # 33..64 inputs (the MPE critic's 54): the first-layer gradient of a 64-wide image does not fit the register file next
            # to the layer's own gradient -- one-launch backward that WRITES dz_1, then the two-operand weight-gradient launch
            fuse_dw1 = False
        for l in range(L - 1, 0, -1):
            ho, hi = self.hidden_sizes[l], self.hidden_sizes[l - 1]
            Wp, _ = self._packs[l]
            dw1_here = l == 1 and fuse_dw1
            if bwd_mode != "0" and ho == 128 and hi == 128 and (not dw1_here or self.kp0 == 32):
                # the whole backward of this layer in ONE launch (round 5): dz and x_hat_{l-1} cross HBM once for dx AND dW'
                call("harl_mlp_bwd_dx_dw", ptr(self.dz[cur]), ptr(self.xh[l - 1]), ptr(self.rmask[l - 1]), ptr(self.rstd[l - 1]), M,
                     ho, hi, ptr(Wp), None if dw1_here else ptr(self.dz[1 - cur]), ptr(self.x0n) if dw1_here else None,
                     self.kp0 if dw1_here else 0, ptr(self.part[po[0]:]) if dw1_here else None, ptr(self.part[po[l]:]), nwg,
                     int(bwd_mode != "nofill"), s, tag="bwd_full_dw1" if dw1_here else "bwd_full")
                cur = 1 - cur
                continue
            # HARL_BWD_STREAMS=1: the layer's weight gradient goes to a SECOND stream, next to the backward-dx launch -- both read
            # dz_l and x_hat_{l-1}, neither reads what the other writes.  The dx launch is enqueued FIRST: one of its workgroups
            # per CU (96 KiB of LDS, the register-lean instantiation) leaves room for one weight-gradient workgroup, so the two
            # kernels share every CU and fill each other's stalls; the other order would park two weight-gradient workgroups on
            # every CU and the dx kernel behind them.
            two = (_bwd_streams() and ho == 128 and hi == 128 and self.device_.type == "cuda"
                   and (not (l == 1 and fuse_dw1) or self.kp0 == 32))
            if side_pending and not (l == 1 and fuse_dw1):
                # this layer's dx launch WRITES dz[1 - cur] -- the buffer the previous layer's weight-gradient launch on the side
                # stream still reads (L >= 3 with an unfused first-layer gradient): order it behind that launch
                torch.cuda.current_stream(self.device_).wait_event(self._bwd_ev[1])
            if two:
                main_s, side_s, e0 = torch.cuda.current_stream(self.device_), self._bwd_side(), self._bwd_ev[0]
                e0.record(main_s)
            else:
                call("harl_mlp_dw_partials", ptr(self.dz[cur]), 0, 0, ho, ptr(self.xh[l - 1]), 0, 0, None, None, None, hi, M,
                     ptr(self.part[po[l]:]), nwg, s, tag="dw_hidden")
            if l == 1 and fuse_dw1:
                call("harl_mlp_bwd_dx", ptr(self.dz[cur]), ptr(self.xh[0]), ptr(self.rmask[0]), ptr(self.rstd[0]), M, ho, hi,
                     ptr(Wp), None, ptr(self.x0n), self.kp0, ptr(self.part[po[0]:]), nwg, s, tag="bwd_dx_dw1")
            else:
                call("harl_mlp_bwd_dx", ptr(self.dz[cur]), ptr(self.xh[l - 1]), ptr(self.rmask[l - 1]),
                     ptr(self.rstd[l - 1]), M, ho, hi, ptr(Wp), ptr(self.dz[1 - cur]), None, 0, None, 0, s, tag="bwd_dx")
            if two:
                side_s.wait_event(e0)
                with torch.cuda.stream(side_s):
                    call("harl_mlp_dw_partials", ptr(self.dz[cur]), 0, 0, ho, ptr(self.xh[l - 1]), 0, 0, None, None, None, hi, M,
                         ptr(self.part[po[l]:]), nwg, side_s.cuda_stream, tag="dw_hidden")
                self._bwd_ev[1].record(side_s)
                side_pending = True
            cur = 1 - cur
        h0 = self.hidden_sizes[0]
        use_ln = self.use_feature_normalization
        if fuse_dw1:
            pass
        elif self.x0n is not None:  # B = normalised inputs as written by the forward pass (ATL, zero-padded to kp0)
            call("harl_mlp_dw_partials", ptr(self.dz[cur]), 0, 0, h0, ptr(self.x0n), 0, 0, None, None, None, self.kp0, M,
                 ptr(self.part[po[0]:]), nwg, s, tag="dw_input")
        else:
            call("harl_mlp_dw_partials", ptr(self.dz[cur]), 0, 0, h0, ptr(X), 1, X.shape[1], ptr(idx),
                 ptr(self.mu0) if use_ln else None, ptr(self.rstd0) if use_ln else None, self.in_dim, M,
                 ptr(self.part[po[0]:]), nwg, s, tag="dw_input")
        if side_pending:  # the weight gradients enqueued on the second stream
            e1 = self._bwd_ev[1]
            e1.record(self._bwd_side())
            torch.cuda.current_stream(self.device_).wait_event(e1)
        # deterministic fixed-order combine of every entry's per-workgroup partials, one launch
        self._combine_partials(s)

    def _backward_trunk_fused(self, M: int, seq: Optional[dict], s) -> None:
        """The 64-wide trunk's backward in three launches (csrc/trunk.hip): [the GRU's recurrence, harl_gru_bwd without its
        input side] -> harl_mlp_bwd_trunk (W_ih'^T d gates + every hidden Linear's dx, every dz written once) ->
        harl_mlp_dw_partials_multi_v (the weight gradients of all MLP layers and of the six gate blocks) -> the combine.
        Each stage is the layer kernel's arithmetic; only the first-layer gradient of 33..63-wide inputs differs from the
        layer path in summation order (that path folds it into harl_mlp_bwd_dx on the matrix-pipe transposes)."""
        import ctypes as C
        L, po, nwg = len(self.hidden_sizes), self._part_offs, self.n_wg
        spare = [self.dz[2]] if L == 3 else []
        if self.recurrent:
            gp, sv, dg = self.gru_pack, self.rnn_saved, self.rnn_dgate
            call("harl_gru_bwd", ptr(self.dz[0]), ptr(seq["mask_rows"]), ptr(gp["Wih"]), ptr(gp["Whh"]), ptr(sv[0]),
                 ptr(sv[1]), ptr(sv[2]), ptr(sv[3]), ptr(sv[4]), 64, seq["L"], seq["m_pad"], ptr(self.xh[-1]),
                 ptr(self.rmask[-1]), ptr(self.rstd[-1]), ptr(dg[0]), ptr(dg[1]), ptr(dg[2]), ptr(dg[3]), None, s, tag="gru_bwd")
            dzs = [self.dz[1], self.dz[0]] + spare  # dz of layers L-1 .. 0 (d(loss)/d(h) in dz[0] is dead after the recurrence)
        else:
            dzs = [self.dz[0], self.dz[1]] + spare
        args = self._trunk_cache.get("bwd")
        if args is None:
            vp = lambda ts: (C.c_void_p * len(ts))(*[ptr(t) for t in ts])  # noqa: E731
            top_down = range(L - 1, -1, -1)
            a, b, part, K, t0, nt = [], [], [], [], [], []

            def add(aa, bb, l, k, t, n):
                a.append(aa), b.append(bb), part.append(self.part[po[l]:]), K.append(k), t0.append(t), nt.append(n)
            if self.recurrent:  # W_ih' blocks d gi_g^T x_hat_mlp (g = r, z, n), W_hh blocks d gh_g^T h~ (dhn for n)
                for k, (aa, bb) in enumerate(zip((dg[0], dg[1], dg[2], dg[0], dg[1], dg[3]), [self.xh[-1]] * 3 + [sv[0]] * 3)):
                    add(aa, bb, L + k, 64, 0, 2)
            for j, l in enumerate(range(L - 1, 0, -1)):
                add(dzs[j], self.xh[l - 1], l, 64, 0, 2)
            for t in range(0, self.kp0 // 32, 4):  # the wide first layer in groups of <= 4 column tiles
                add(dzs[L - 1], self.x0n, 0, self.kp0, t, min(4, self.kp0 // 32 - t))
            def pack(lo, hi):  # problems lo .. hi - 1 as the argument arrays of one harl_mlp_dw_partials_multi_v launch
                n = hi - lo
                ci = lambda v: (C.c_int * n)(*v[lo:hi])  # noqa: E731
                return (n, vp(a[lo:hi]), vp(b[lo:hi]), vp(part[lo:hi]), ci(K), ci(t0), ci(nt))
            ng = 6 if self.recurrent else 0
            args = self._trunk_cache["bwd"] = (
                vp([self._packs[l][0] for l in range(L - 1, 0, -1)]), vp([self.xh[l] for l in top_down]),
                vp([self.rmask[l] for l in top_down]), vp([self.rstd[l] for l in top_down]),
                vp([dzs[0] if self.recurrent else None] + dzs[1:]), pack(0, len(a)), pack(0, ng) if ng else None, pack(ng, len(a)))
        pW, pX, pM, pR, pD, dw_all, dw_gates, dw_mlp = args
        # The gate blocks' weight gradients need nothing from the trunk's backward (d gates and the saved activations are there
        # once the recurrence is done): HARL_TRUNK_DW_STREAM=1 sends them to a SECOND stream next to harl_mlp_bwd_trunk (one
        # workgroup of each fits a CU: 120 + 37 KiB of LDS, 272 + 64 registers per SIMD lane).  MEASURED SLOWER and therefore
        # opt-in: 26.4 / 25.9 against 23.8 / 23.7 ms per SMAC 3s5z update, 116.1 / 116.3 against 114.0 / 114.1 ms at 4096
        # threads (gpurun_out/r06h, call 3) -- two cross-stream dependencies per optimiser step cost more than the ~40 us of
        # overlap return.  Default: one launch for all nine problems behind the trunk's backward.
        two = (self.recurrent and self.device_.type == "cuda" and os.environ.get("HARL_TRUNK_DW_STREAM", "0") == "1")
        if two:
            main_s, side_s, (e0, e1) = torch.cuda.current_stream(self.device_), self._bwd_side(), self._bwd_ev
            e0.record(main_s)
            side_s.wait_event(e0)
            with torch.cuda.stream(side_s):
                call("harl_mlp_dw_partials_multi_v", dw_gates[0], *dw_gates[1:4], 64, *dw_gates[4:], M, nwg, side_s.cuda_stream,
                     tag="dw_gru")
                e1.record(side_s)
        if self.recurrent:
            call("harl_mlp_bwd_trunk", M, 64, L - 1, ptr(gp["Wih"]), ptr(dg[0]), ptr(dg[1]), ptr(dg[2]), None, pW, pX, pM, pR, pD, s,
                 tag="bwd_trunk")
        else:
            call("harl_mlp_bwd_trunk", M, 64, L - 1, None, None, None, None, ptr(self.dz[0]), pW, pX, pM, pR, pD, s, tag="bwd_trunk")
        # the six gate blocks as ONE problem whose six operand images are each read once (harl_gru_dw6: 1 536 instead of 3 072 B
        # per row on a launch that is bound by its row traffic), then the MLP layers.  Bit-identical to the nine problems of the
        # generic launch; it is one launch more, which costs what it saves at 81 920 rows per minibatch (24.2 / 24.3 against
        # 23.8 / 24.1 ms per SMAC 3s5z update) and pays at 655 360 (112.2 / 112.6 against 116.7 / 116.7 ms; gpurun_out/r06h call
        # 6): taken from 160 000 rows.  HARL_GRU_DW6=1 / 0 force it on / off.
        g6 = os.environ.get("HARL_GRU_DW6", "auto")
        gru6 = self.recurrent and not two and (g6 == "1" or (g6 == "auto" and M >= 160000))
        if gru6:
            call("harl_gru_dw6", ptr(dg[0]), ptr(dg[1]), ptr(dg[2]), ptr(dg[3]), ptr(self.xh[-1]), ptr(sv[0]), dw_gates[3], M, nwg, s,
                 tag="dw_gru")
        dw = dw_mlp if (two or gru6) else dw_all
        call("harl_mlp_dw_partials_multi_v", dw[0], *dw[1:4], 64, *dw[4:], M, nwg, s, tag="dw_trunk")
        if two:
            main_s.wait_event(e1)
        self._combine_partials(s)

    # ---- MultiDiscrete heads (csrc/multihead.hip): logits of every group from the head input; backward of the groups
    def md_logits(self, M: int) -> None:
        fx, _, _, fh = self.feat()
        s = stream()
        for (Wp, bp), z, sp in zip(self._head_packs, self.md_z, self._md_sp):
            call("harl_mlp_linear", ptr(fx), M, fh, sp, ptr(Wp), ptr(bp), ptr(z), s, tag="md_linear")

    def md_layout(self):
        """(z pointer array, n_groups, sp[], n_heads, nvec[], head_group[]) -- the leading arguments of harl_md_head_*."""
        import ctypes as C
        G, K = len(self._md_sp), len(self.nvec)
        zs = (C.c_void_p * G)(*[z.data_ptr() for z in self.md_z])
        return (zs, G, (C.c_int * G)(*self._md_sp), K, (C.c_int * K)(*self.nvec), (C.c_int * K)(*self._md_head_group))

    def md_backward(self, M: int) -> None:
        """d(loss)/d(logits) images (self.md_z, written by harl_md_head_loss) -> the groups' weight-gradient partials and
        d(loss)/d(head input) in self.dz[0]: the layer kernels of a hidden Linear (sum over the groups)."""
        fx, fmask, frstd, fh = self.feat()
        s = stream()
        nh = len(self._md_sp)
        po = self._part_offs[-nh:]
        for g, ((Wp, _), dzg, sp) in enumerate(zip(self._head_packs, self.md_z, self._md_sp)):
            call("harl_mlp_dw_partials", ptr(dzg), 0, 0, sp, ptr(fx), 0, 0, None, None, None, fh, M,
                 ptr(self.part[po[g]:]), self.n_wg, s, tag="dw_head")
            call("harl_mlp_bwd_dx", ptr(dzg), ptr(fx), ptr(fmask), ptr(frstd), M, sp, fh, ptr(Wp),
                 ptr(self.dz[0 if g == 0 else 1]), None, 0, None, 0, s, tag="bwd_dx")
            if g > 0:
                n = ((M + SLAB - 1) // SLAB) * SLAB * fh
                self.dz[0][:n].add_(self.dz[1][:n])

    def unfold_grads(self) -> None:
        """self.dwp (dense folded gradients) -> self.flat_grad in the reference parameter layout (UNSCALED)."""
        s = stream()
        fp, fg = self.flat_param, self.flat_grad
        if self.table is not None and os.environ.get("HARL_UNFOLD_TABLE", "1") != "0":  # every entry in one launch
            call("harl_unfold_table", ptr(fp), ptr(fg), ptr(self.dwp), ptr(self.table), self.n_entries,
                 sum(r[5] for r in self._table_rows), s)
            return
        seen = set()
        for (wo, bo, go, beo, o, k), off, trow in zip(self._entries(), self._dwp_offs, self._table_rows):
            kp, op = trow[9], trow[10]
            dW = self.dwp[off:off + op * kp]
            db = self.dwp[off + op * kp:off + op * kp + op]
            acc = int(go in seen)  # Linears sharing one LayerNorm (GRU gate blocks) accumulate its gradients
            if go >= 0:
                seen.add(go)
            call("harl_unfold_linear_grads", ptr(dW), ptr(db), kp, ptr(fp[wo:]), ptr(fp[go:]) if go >= 0 else None,
                 ptr(fp[beo:]) if beo >= 0 else None, ptr(fg[wo:]), ptr(fg[bo:]), ptr(fg[go:]) if go >= 0 else None,
                 ptr(fg[beo:]) if beo >= 0 else None, o, k, acc, s)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):  # keep views, then refold
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        self._fold_version = None
        self.fold()
        return out


class StochasticPolicy(_FlatNet):
    """Actor network container (reference: models/policy_models/stochastic_policy.py:11-54)."""

    def __init__(self, args: dict, obs_space, action_space, device: torch.device):
        obs_shape = _space_shape(obs_space)
        if len(obs_shape) != 1:
            raise NotImplementedError("image observations (CNNBase) are outside the accelerated path")
        super().__init__(args, obs_shape[0], device)
        self.action_type = action_space.__class__.__name__
        self.std_x_coef = float(args["std_x_coef"])
        self.std_y_coef = float(args["std_y_coef"])
        init = getattr(nn.init, args["initialization_method"])
        d = self.hidden_sizes[-1]
        if self.action_type == "Discrete":  # Categorical (distributions.py:37-55)
            self.discrete, self.act_dim, self.act_w = True, int(action_space.n), 1
            lin = nn.Linear(d, self.act_dim)
            init(lin.weight.data, gain=args["gain"])
            nn.init.constant_(lin.bias.data, 0)
            self._cpu_params += [("act.action_out.linear.weight", lin.weight.data), ("act.action_out.linear.bias", lin.bias.data)]
        elif self.action_type == "Box":  # DiagGaussian (distributions.py:58-89); log_std precedes fc_mean in parameters()
            self.discrete, self.act_dim = False, int(action_space.shape[0])
            self.act_w = self.act_dim
            lin = nn.Linear(d, self.act_dim)
            init(lin.weight.data, gain=args["gain"])
            nn.init.constant_(lin.bias.data, 0)
            self._cpu_params += [("act.action_out.log_std", torch.ones(self.act_dim) * self.std_x_coef),
                                 ("act.action_out.fc_mean.weight", lin.weight.data), ("act.action_out.fc_mean.bias", lin.bias.data)]
        elif self.action_type == "MultiDiscrete":  # one Categorical per entry of nvec (act.py:35-43)
            self._init_multidiscrete(args, action_space, init, d)
        else:
            raise NotImplementedError(f"action space {self.action_type}")
        if not self.md and (self.act_dim > 64 or (self.act_dim > 32 and not self.discrete)):
            raise NotImplementedError("action heads wider than 32 (Categorical: 64) are not instantiated")
        self.wide_head = (not self.md) and self.act_dim > 32
        self._finalize_params()
        self.fold()

    def _init_multidiscrete(self, args: dict, action_space, init, d: int) -> None:
        """Heads in the reference's order (same RNG draws), packed first-fit IN ORDER into groups of <= 128 logits; a group is
        one contiguous [S_g, d] block of the arena so that fold / weight-gradient / Adam see it as a single Linear."""
        nvec = [int(n) for n in action_space.nvec]
        if self.panel:
            raise NotImplementedError("MultiDiscrete heads on 256-wide layers")
        if len(nvec) > 8 or max(nvec) > 128:
            raise NotImplementedError("MultiDiscrete: at most 8 heads of at most 128 actions each")
        if self.act_id:
            raise NotImplementedError("MultiDiscrete heads with an activation other than relu are not implemented")
        self.md, self.discrete = True, True
        self.nvec, self.n_heads = nvec, len(nvec)
        self.act_dim = sum(nvec)   # width of the concatenated normalised logits (head_out)
        self.act_w = 1             # log-probs are summed over the heads (act.py:124-137)
        groups: List[List[int]] = [[]]
        for k, n in enumerate(nvec):
            if sum(nvec[j] for j in groups[-1]) + n > 128:
                groups.append([])
            groups[-1].append(k)
        if len(groups) > 4:
            raise NotImplementedError("MultiDiscrete: more than 4 groups of 128 logits")
        self._md_groups = groups
        self._md_head_group = [g for g, ks in enumerate(groups) for _ in ks]
        self._md_sp = [64 if sum(nvec[k] for k in ks) <= 64 else 128 for ks in groups]
        lins = []
        for n in nvec:  # Categorical(inputs_dim, n): init_(nn.Linear) with gain (distributions.py:37-50)
            lin = nn.Linear(d, n)
            init(lin.weight.data, gain=args["gain"])
            nn.init.constant_(lin.bias.data, 0)
            lins.append(lin)
        hidden, alias = [], []
        for g, ks in enumerate(groups):
            self._cpu_params += [(f"__md{g}.weight", torch.cat([lins[k].weight.data for k in ks], 0)),
                                 (f"__md{g}.bias", torch.cat([lins[k].bias.data for k in ks], 0))]
            hidden += [f"__md{g}.weight", f"__md{g}.bias"]
        for g, ks in enumerate(groups):
            lo = 0
            for k in ks:
                alias.append((f"act.action_outs.{k}.linear.weight", f"__md{g}.weight", lo, lo + nvec[k]))
                alias.append((f"act.action_outs.{k}.linear.bias", f"__md{g}.bias", lo, lo + nvec[k]))
                lo += nvec[k]
        self._hidden_blocks, self._alias_params = tuple(hidden), alias

    def _head_layers(self):
        if self.md:
            return [(f"__md{g}.weight", f"__md{g}.bias", sum(self.nvec[k] for k in ks)) for g, ks in enumerate(self._md_groups)]
        return [self._head_names()]

    def _head_names(self):
        if self.discrete:
            return "act.action_out.linear.weight", "act.action_out.linear.bias", self.act_dim
        return "act.action_out.fc_mean.weight", "act.action_out.fc_mean.bias", self.act_dim

    def log_std(self) -> Optional[torch.Tensor]:
        return None if self.discrete else self.pview("act.action_out.log_std")


class VNet(_FlatNet):
    """Critic network container (reference: models/value_function_models/v_net.py:10-46)."""

    def __init__(self, args: dict, cent_obs_space, device: torch.device):
        shape = _space_shape(cent_obs_space)
        if len(shape) != 1:
            raise NotImplementedError("image observations (CNNBase) are outside the accelerated path")
        super().__init__(args, shape[0], device)
        init = getattr(nn.init, args["initialization_method"])
        lin = nn.Linear(self.hidden_sizes[-1], 1)
        init(lin.weight.data, gain=1)
        nn.init.constant_(lin.bias.data, 0)
        self._cpu_params += [("v_out.weight", lin.weight.data), ("v_out.bias", lin.bias.data)]
        self._finalize_params()
        self.fold()

    def _head_names(self):
        return "v_out.weight", "v_out.bias", 1


def consume_policy_init_rng(args: dict, obs_space, action_space) -> None:
    """Draw from the global CPU generator exactly what constructing a ``StochasticPolicy`` draws (same torch calls, same
    order) without allocating anything on the device.  HATRPO.update builds a fresh policy object as its "old actor"
    snapshot on every call (algorithms/actors/hatrpo.py:127-130); those draws are part of the RNG stream of train()
    and must happen for the later minibatch permutations / agent orders to match the reference."""
    from .buffers import rng_sync
    rng_sync()
    init = getattr(nn.init, args["initialization_method"])
    gain = nn.init.calculate_gain(args.get("activation_func", "relu"))

    def draw(lin, g):
        if args["initialization_method"] == "orthogonal_":
            # nn.init.orthogonal_ = normal_(0, 1) on a [rows, cols] tensor + a QR (no further draws): only the normal_
            # touches the generator, and the QR of a 393 x 128 matrix costs ~8 ms of host time per call
            w = lin.weight.data
            w.new_empty((w.size(0), w.numel() // w.size(0))).normal_(0, 1)
        else:
            init(lin.weight.data, gain=g)

    d = _space_shape(obs_space)[0]
    for h in args["hidden_sizes"]:
        draw(nn.Linear(d, h), gain)
        d = h
    if args.get("use_recurrent_policy") or args.get("use_naive_recurrent_policy"):
        # RNNLayer (rnn.py:8-21): nn.GRU's own uniform init of all four tensors, then init_method on the two weights
        gru = nn.GRU(d, d, num_layers=args.get("recurrent_n", 1))
        for pn, prm in gru.named_parameters():
            if "weight" in pn:
                if args["initialization_method"] == "orthogonal_":
                    prm.data.new_empty((prm.size(0), prm.numel() // prm.size(0))).normal_(0, 1)
                else:
                    init(prm.data)
    kind = action_space.__class__.__name__
    if kind == "MultiDiscrete":
        for n in action_space.nvec:
            draw(nn.Linear(d, int(n)), args["gain"])
        return
    n_out = int(action_space.n) if kind == "Discrete" else int(action_space.shape[0])
    draw(nn.Linear(d, n_out), args["gain"])


_FIRST_RING: dict = {}


def _upload_first_rows(first_rows: torch.Tensor, dev) -> torch.Tensor:
    """CPU int64 sequence starts -> device, asynchronously: copied into the next slot of a ring of pinned buffers (an event per
    slot guards its reuse: 32 slots = 32 minibatches back, long executed) and from there with a non-blocking copy on the current
    stream.  Device tensors pass through."""
    if first_rows.is_cuda:
        return first_rows
    n = first_rows.numel()
    ring = _FIRST_RING.get(dev)
    if ring is None or ring["cap"] < n:
        cap = max(4096, 2 * n)
        ring = dict(cap=cap, k=0, bufs=[torch.empty(cap, dtype=torch.int64, pin_memory=True) for _ in range(32)],
                    evs=[None] * 32)
        _FIRST_RING[dev] = ring
    k = ring["k"]
    ring["k"] = (k + 1) % 32
    if ring["evs"][k] is not None:
        ring["evs"][k].synchronize()
    stage = ring["bufs"][k][:n]
    # (NumPy for the host-side copy: ATen's parallel copy wakes its whole intra-op pool for >= 32 768 elements -- tens of
    # milliseconds per minibatch on a 256-thread host, buffers.draw_permutation)
    stage.numpy()[:] = first_rows.reshape(-1).numpy()
    out = stage.to(dev, non_blocking=True)
    ev = ring["evs"][k] or torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    ring["evs"][k] = ev
    return out


def build_seq(dev, L: int, m: int, H: int, *, first_rows: Optional[torch.Tensor] = None, stride: Optional[int] = None,
              h0: Optional[torch.Tensor] = None, h0_src: Optional[torch.Tensor] = None,
              masks_src: Optional[torch.Tensor] = None, want_h_last: bool = False) -> dict:
    """Layout of a recurrent batch for the GRU kernels (csrc/gru.hip): L time steps x m sequences, row (l, j) at
    l*m_pad + j with m_pad = m rounded up to the 32-sample slab (padding sequences replay sequence 0 and are ignored by
    the loss kernels through m_valid/m_pad).

    buffer mode  (first_rows [m] CPU/device int64, stride = N): sequence j covers source rows first_rows[j] + l*stride
                 of the t-major flattened buffers -- the reference's chunk slicing (on_policy_actor_buffer.py:255-322)
                 and naive whole-column sampling (:180-221); h0 = h0_src[first_rows], masks = masks_src[rows].
    direct mode  (first_rows None): the caller's arrays are already [L*m, .] l-major (evaluate_actions / get_actions,
                 rnn.py:36-42 convention); h0 [m, H] and masks_src [L*m] are given explicitly.
    Returns dict(L, m, m_pad, idx (None = identity), valid_idx, h0, mask_rows, h_last)."""
    m_pad = (m + 31) // 32 * 32
    direct = first_rows is None
    if not direct and h0 is None and torch.device(dev).type == "cuda":
        # buffer mode on the device: ONE launch (harl_build_seq) instead of eight small torch kernels per minibatch; the sequence
        # starts travel through a ring of pinned staging buffers, so that the upload is asynchronous (a pageable source made
        # every minibatch wait for its copy with the stream drained: 20 us of idle GPU x 45 minibatches per 8-agent update)
        first = _upload_first_rows(first_rows, dev)
        n = L * m_pad
        idx = torch.empty(n, dtype=torch.int64, device=dev)
        valid_idx = idx if m_pad == m else torch.empty(L * m, dtype=torch.int64, device=dev)
        mask_rows = torch.empty(n, dtype=torch.float32, device=dev)
        h0o = torch.empty(m_pad, H, dtype=torch.float32, device=dev)
        msrc, hsrc = masks_src.reshape(-1), h0_src.reshape(-1, H)
        assert msrc.is_contiguous() and hsrc.is_contiguous() and msrc.dtype == torch.float32 and hsrc.dtype == torch.float32
        call("harl_build_seq", ptr(first), m, m_pad, L, int(stride), ptr(msrc), ptr(hsrc), H, ptr(idx),
             None if m_pad == m else ptr(valid_idx), ptr(mask_rows), ptr(h0o), stream())
        return dict(L=L, m=m, m_pad=m_pad, idx=idx, valid_idx=valid_idx, h0=h0o, mask_rows=mask_rows,
                    h_last=torch.empty(m_pad, H, dtype=torch.float32, device=dev) if want_h_last else None)
    if direct:
        first, stride = torch.arange(m, device=dev), m
    else:
        first = first_rows.to(dev)
    if m_pad != m:
        first = torch.cat([first, first[:1].expand(m_pad - m)])
    if direct and m_pad == m:
        idx = valid_idx = None
        mask_rows = masks_src.reshape(-1).contiguous()
    else:
        rows = first[None, :] + torch.arange(L, device=dev)[:, None] * stride
        idx = rows.reshape(-1).contiguous()
        valid_idx = idx if m_pad == m else rows[:, :m].reshape(-1).contiguous()
        mask_rows = masks_src.reshape(-1)[idx].contiguous()
    if h0 is None:
        h0 = h0_src.reshape(-1, H)[first]
    elif m_pad != m:
        h0 = torch.cat([h0.reshape(m, H), torch.zeros(m_pad - m, H, dtype=h0.dtype, device=dev)])
    return dict(L=L, m=m, m_pad=m_pad, idx=idx, valid_idx=valid_idx, h0=h0.reshape(m_pad, H).contiguous(),
                mask_rows=mask_rows, h_last=torch.empty(m_pad, H, dtype=torch.float32, device=dev) if want_h_last else None)


def seq_compact(t: torch.Tensor, seq: dict) -> torch.Tensor:
    """[L*m_pad, w] padded rows -> [L*m, w] (drop the padding sequences)."""
    if seq["m_pad"] == seq["m"]:
        return t
    w = t.shape[1:] if t.dim() > 1 else ()
    return t.reshape(seq["L"], seq["m_pad"], *w)[:, :seq["m"]].reshape(seq["L"] * seq["m"], *w)


class FusedAdam:
    """torch.optim.Adam semantics (defaults betas=(0.9,0.999), amsgrad=False) over one flat arena, fused with
    the gradient-norm / clip step into a single launch (harl_gradnorm_clip_adam).  Exposes ``param_groups`` so the
    reference's ``update_linear_schedule`` (utils/models_tools.py:77-87) works unchanged."""

    def __init__(self, net: _FlatNet, lr: float, eps: float, weight_decay: float):
        self.net = net
        self.param_groups = [dict(lr=lr, betas=(0.9, 0.999), eps=eps, weight_decay=weight_decay)]
        self.exp_avg = torch.zeros_like(net.flat_param)
        self.exp_avg_sq = torch.zeros_like(net.flat_param)
        self.step_count = 0
        self._ws = torch.zeros(8192, dtype=torch.int32, device=net.flat_param.device)  # grid-barrier words + partials

    def zero_grad(self) -> None:  # gradients are overwritten, never accumulated
        return None

    def step(self, mode: int, const_scale: float, use_clip: bool, max_norm: float, info_out: Optional[torch.Tensor],
             logstd_off: int = -1, act_dim: int = 0, part_scalars: Optional[torch.Tensor] = None,
             n_scalar_blocks: int = 0, scalars_hilo: Optional[torch.Tensor] = None) -> None:
        """One fused launch: loss scalars -> scale/statistics, unfold, ||g||, clip, Adam, re-fold (harl_adam_fold).
        ``part_scalars``: the loss kernel's per-block partial sums, reduced inside the launch into ``net.scalars``
        (single-GPU path); ``scalars_hilo``: the all-reduced fixed-grid fp32 pieces behind the gradients in ``net.dwp_msg``
        (data-parallel path); neither = ``net.scalars`` already holds the sums."""
        g = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        n = self.net
        call("harl_adam_fold", ptr(n.flat_param), ptr(n.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), n.n_params,
             ptr(n.dwp), ptr(n.table), n.n_entries, ptr(n.pack_arena), ptr(n.scalars), ptr(part_scalars),
             int(n_scalar_blocks), ptr(scalars_hilo), int(mode), float(const_scale), int(logstd_off), int(act_dim), ptr(info_out),
             int(use_clip), float(max_norm), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
             float(g["weight_decay"]), bc1, bc2, ptr(self._ws), stream(), tag="adam_fold")

    def state_dict(self) -> dict:
        return dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                    param_groups=[dict(g) for g in self.param_groups])

    def load_state_dict(self, sd: dict) -> None:
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups = [dict(g) for g in sd["param_groups"]]
