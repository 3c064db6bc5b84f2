"""ctypes binding of libharl_hip.so (the C ABI declared in include/harl_hip.h).

The HIP library is the *only* compute backend of this package: if it is missing the import of
any compute class fails loudly (there is no CPU / PyTorch fallback -- the oracle under oracle/ is
test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from .traffic import algorithmic_bytes, algorithmic_flops

_HERE = os.path.dirname(os.path.abspath(__file__))
# HARL_LIB=<variant> selects harl_amd/lib/libharl_<variant>.so (A/B builds of the same sources, harl_amd/_build.py)
LIB_PATH = os.path.join(_HERE, "lib", "libharl_%s.so" % (os.environ.get("HARL_LIB") or "hip"))

PS_STRIDE = 48
DHEAD_LD = 32
SLAB = 32

_vp, _i, _l, _f, _d = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double

# name -> argtypes (restype is int unless noted); mirrors include/harl_hip.h line by line
SIGNATURES = {
    "harl_gae_returns": [_vp] * 8 + [_i, _i, _f, _f, _i, _i, _i, _vp],
    "harl_masked_moments": [_vp, _vp, _l, _vp, _vp, _vp],
    "harl_dist_rows": [_vp, _l, _i, _i, _vp, _f, _f, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_moments_mean": [_vp, _vp, _vp],
    "harl_clock_probe": [_vp, _l, _vp],
    "harl_build_seq": [_vp, _i, _i, _i, _l, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "harl_adv_normalize": [_vp, _vp, _vp, _l, _vp],
    "harl_factor_update": [_vp, _vp, _vp, _l, _i, _i, _vp],
    "harl_sum_sumsq": [_vp, _vp, _l, _vp, _vp],
    "harl_valuenorm_apply": [_vp, _vp, _d, _d, _vp],
    "harl_gradnorm_clip_adam": [_vp, _vp, _vp, _vp, _l, _vp, _i, _f, _d, _d, _d, _f, _f, _d, _d, _vp, _vp],
    "harl_fold_linear": [_vp] * 6 + [_i, _i, _vp],
    "harl_unfold_linear_grads": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "harl_mlp_fwd_input": [_vp, _l, _vp, _l, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_mlp_x0n_wide": [_vp, _l, _vp, _l, _i, _i, _vp, _vp, _vp, _vp],
    "harl_mlp_fwd_wide": [_vp, _l, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "harl_mlp_linear_wide": [_vp, _l, _i, _vp, _i, _vp, _i, _vp, _vp, _vp],
    "harl_act_ln_fwd": [_vp, _l, _i, _i, _vp, _vp, _vp, _vp],
    "harl_act_bwd": [_vp, _vp, _vp, _vp, _l, _i, _i, _vp],
    "harl_act_ln_tangent": [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp],
    "harl_mlp_tangent_wide": [_vp, _l, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_mlp_fwd_fused2x": [_vp, _l, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_mlp_fwd_fused2": [_vp, _l, _vp, _l, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp],
    "harl_mlp_fwd_hidden": [_vp, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_mlp_bwd_dx": [_vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "harl_mlp_bwd_dx_dw": [_vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp],
    "harl_mlp_dw_partials": [_vp, _i, _i, _i, _vp, _i, _l, _vp, _vp, _vp, _i, _l, _vp, _i, _vp],
    "harl_mlp_dw_partials_multi": [_i, _vp, _vp, _vp, _i, _i, _l, _i, _vp],
    "harl_mlp_dw_partials_multi_v": [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _l, _i, _vp],
    "harl_gru_dw6": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp],
    "harl_mlp_fwd_trunk": [_vp, _l, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_mlp_bwd_trunk": [_l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_reduce_partials": [_vp, _i, _l, _vp, _vp],
    "harl_reduce_partials_multi": [_vp, _vp, _i, _i, _l, _vp, _vp],
    "harl_adam_fold": [_vp, _vp, _vp, _vp, _l, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _f, _i, _i, _vp, _i, _f, _d, _d,
                       _d, _f, _f, _d, _d, _vp, _vp],
    "harl_pack_scalars_hilo": [_vp, _vp, _vp],
    "harl_reduce_pack_scalars": [_vp, _i, _vp, _vp, _vp],
    "harl_randperm_replay": [_vp, _l, _l, _vp, _vp, _vp],
    "harl_rng_advance": [_vp, _l, _l, _vp],
    "harl_rng_jump": [_vp, _l, _l, _vp],
    "harl_actor_head_logp": [_vp, _l, _i, _vp, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _l, _l, _vp],
    "harl_actor_head_loss": [_vp, _vp, _vp, _l, _i, _vp, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                             _vp, _vp, _d, _f, _i, _i, _l, _l, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "harl_critic_head_values": [_vp, _l, _i, _vp, _vp, _vp, _vp],
    "harl_critic_head_loss": [_vp, _vp, _vp, _l, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _f, _l, _l, _vp, _vp, _vp,
                              _vp, _i, _vp],
    "harl_gru_fwd": [_vp] * 7 + [_i, _i, _l] + [_vp] * 8 + [_i, _vp, _vp],
    "harl_gru_bwd": [_vp] * 9 + [_i, _i, _l] + [_vp] * 8 + [_vp],
    "harl_fold_linear_tangent": [_vp] * 9 + [_i, _i, _vp],
    "harl_fold_table": [_vp, _vp, _vp, _i, _i, _vp],
    "harl_unfold_table": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "harl_fold_tangent_table": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "harl_mlp_tangent_input": [_vp, _l, _vp, _l, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "harl_gru_gates": [_vp, _vp, _i, _l, _vp, _vp, _vp, _i, _vp],
    "harl_gru_tangent": [_vp] * 15 + [_i, _i, _l, _vp, _vp],
    "harl_mlp_tangent_hidden": [_vp, _vp, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_mlp_tangent_hidden2": [_vp, _vp, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_actor_head_fvp": [_vp, _vp, _vp, _vp, _l, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _vp, _l, _l, _vp, _vp, _vp],
    "harl_trpo_kl_sum": [_vp, _vp, _vp, _vp, _f, _f, _l, _i, _i, _vp, _vp],
    "harl_update_supported": [_i, _i, _i, _i],
    "harl_update_fwd_actor": [_vp, _l, _i, _i] + [_vp] * 7 + [_f, _f, _i, _i] + [_vp] * 7 + [_d, _f, _i, _i] + [_vp] * 4 + [_i] + [_vp] * 4,
    "harl_update_logp": [_vp, _l, _i, _i] + [_vp] * 7 + [_f, _f, _i, _i] + [_vp] * 5 + [_i, _vp, _vp],
    "harl_update_last_actor": [_vp, _l, _i] + [_vp] * 5 + [_f, _f, _i, _i] + [_vp] * 8 + [_d, _f, _i, _i] + [_vp] * 4 + [_i, _vp],
    "harl_update_last_critic": [_vp, _l, _i] + [_vp] * 8 + [_f, _i, _i, _f] + [_vp] * 3 + [_i, _vp],
    "harl_update_fwd_critic": [_vp, _l, _i, _i] + [_vp] * 9 + [_f, _i, _i, _f] + [_vp] * 3 + [_i] + [_vp] * 4,
    "harl_update_values": [_vp, _l, _i, _i] + [_vp] * 7 + [_vp],
    "harl_update_bwd": [_vp, _vp, _l, _i, _i] + [_vp] * 5 + [_i, _vp],
    "harl_head_blocks": [_l],
    "harl_trpo_fvp_finish": [_vp, _vp, _vp, _vp, _l, _f, _f, _l, _i, _f, _f, _vp],
    "harl_trpo_cg_step": [_vp, _vp, _vp, _vp, _l, _vp, _vp, _vp],
    "harl_mlp_panel_fwd": [_vp, _l, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "harl_mlp_panel_bwd": [_vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _vp],
    "harl_head_dw_rows256": [_vp, _l, _i, _vp, _vp, _i, _vp],
    "harl_mlp_panel_tangent": [_vp, _vp, _l, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_reduce_scalars": [_vp, _i, _vp, _vp],
    "harl_mlp_linear": [_vp, _l, _i, _i, _vp, _vp, _vp, _vp],
    "harl_mlp_linear3": [_vp, _vp, _vp, _l, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "harl_gru_cell_init": [_vp, _vp, _i, _l, _vp, _vp],
    "harl_gru_cell_fwd": [_vp] * 8 + [_i, _l] + [_vp] * 7 + [_vp],
    "harl_gru_cell_bwd": [_vp] * 10 + [_i, _l] + [_vp] * 5 + [_vp],
    "harl_rownorm": [_vp, _l, _i, _vp, _vp, _vp],
    "harl_gru_cell_tangent": [_vp] * 19 + [_i, _l] + [_vp] * 3,
    "harl_md_head_logp": [_vp, _i, _vp, _i, _vp, _vp, _l, _vp, _vp, _vp, _i, _vp, _i, _vp, _l, _l, _vp],
    "harl_md_head_loss": [_vp, _vp, _i, _vp, _i, _vp, _vp, _l, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _d, _f, _i, _i,
                          _l, _l, _vp, _vp, _i, _vp],
    "harl_trpo_begin": [_vp, _vp, _l, _i, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp],
    "harl_trpo_step": [_vp, _vp, _vp, _vp, _vp, _vp, _l, _f, _vp, _vp, _vp],
    "harl_trpo_ls_candidate": [_vp, _vp, _vp, _vp, _l, _vp],
    "harl_trpo_ls_test": [_vp, _vp, _d, _d, _d, _d, _vp, _vp],
    "harl_reduce_scalars_set": [_vp, _i, _vp, _vp],
    "harl_zero_bytes": [_vp, _l, _vp],
    "harl_comm_create": [_i, _i, _l, _i, _vp, _vp],
    "harl_comm_connect": [_vp, _vp],
    "harl_comm_allreduce": [_vp, _vp, _l, _i, _vp],
    "harl_comm_set_timeout": [_vp, _d],
    "harl_comm_status": [_vp],
    "harl_comm_destroy": [_vp],
    "harl_version": [],
}


class HipLibraryMissing(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library (built by ``__graft_entry__.build()`` / ``python -m harl_amd._build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m harl_amd._build` (hipcc, gfx950). "
            "harl_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.harl_last_error.argtypes = []
    lib.harl_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a (contiguous) tensor, None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "harl_amd kernels need contiguous tensors"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    """Raw hipStream_t of torch's current stream (PyTorch-ROCm names the HIP device 'cuda').  Every launch asks for it:
    ``torch.cuda.current_stream()`` builds a Stream object through three Python layers -- 8 us a call, 3.3 ms of host time per
    8-agent recurrent update (cProfile, round 6: the host was within 3 ms of the GPU there) -- the raw query behind it is one
    C call."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_timing_on = False
_timing_events: dict = {}


_timing_tags = None


_event_pool: list = []


def reserve_timing_events(n: int) -> None:
    """Create ``n`` timing events ahead of the launches that will use them."""
    while len(_event_pool) < n:
        _event_pool.append(torch.cuda.Event(enable_timing=True))


def enable_kernel_timing(on: bool, tags=None) -> None:
    """Bracket tagged launches (all, or only those whose tag is in ``tags``) with HIP events on the launch stream
    (torch's current stream, which is the stream handed to the kernels).  Used by bench.py for the roofline figures."""
    global _timing_on, _timing_tags
    _timing_on = bool(on)
    _timing_tags = set(tags) if tags else None
    if on:
        _timing_events.clear()


def collect_kernel_timing() -> dict:
    """{tag: {n, avg_ms, total_ms, bytes}} for the launches recorded since enable_kernel_timing(True)."""
    torch.cuda.synchronize()
    out = {}
    for tag, evs in _timing_events.items():
        ms = [e[0].elapsed_time(e[1]) for e in evs]
        nb = [e[2] for e in evs if e[2] is not None]
        out[tag] = dict(n=len(ms), avg_ms=sum(ms) / max(len(ms), 1), total_ms=sum(ms),
                        # algorithmic HBM bytes of the launches that ran (harl_amd/traffic.py), None if no model
                        bytes=sum(nb) if len(nb) == len(evs) and nb else None,
                        # executed Linear-layer FLOPs of the same launches (0 for kernels without a GEMM)
                        flops=sum(e[3] or 0.0 for e in evs))
    return out


_fn_cache: dict = {}


def call(name: str, *args, tag: Optional[str] = None) -> None:
    fn = _fn_cache.get(name)
    if fn is None:  # (one attribute lookup on the CDLL per entry point and process, not per launch)
        fn = _fn_cache[name] = getattr(load(), name)
    if _timing_on and tag is not None and (_timing_tags is None or tag in _timing_tags):
        # events come from a pool filled outside the timed region (hipEventCreate is not free on the launch path)
        a = _event_pool.pop() if _event_pool else torch.cuda.Event(enable_timing=True)
        b = _event_pool.pop() if _event_pool else torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        _timing_events.setdefault(tag, []).append((a, b, algorithmic_bytes(name, args), algorithmic_flops(name, args)))
    else:
        rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {load().harl_last_error().decode()}")


SCRATCH_BYTES = {"mm": 24640, "cg": 1088}  # include/harl_hip.h: HARL_MM_SCRATCH_BYTES, HARL_CG_SCRATCH_BYTES
_scratch: Dict[tuple, torch.Tensor] = {}


def scratch(kind: str) -> int:
    """Device pointer of the caller-owned scratch block of ``harl_masked_moments`` ("mm") / ``harl_trpo_cg_step`` ("cg") for the
    CURRENT device and stream: the C ABI allocates nothing, so the blocks are torch tensors, one per (device, stream, kind) --
    two streams may have such a launch in flight at the same time -- zero-filled on that stream when first asked for and kept
    for the life of the process (a few KiB each)."""
    st = torch.cuda.current_stream()
    key = (st.device_index, st.cuda_stream, kind)
    t = _scratch.get(key)
    if t is None:
        t = _scratch[key] = torch.zeros(SCRATCH_BYTES[kind] // 8, dtype=torch.float64, device=torch.device("cuda", st.device_index))
    return t.data_ptr()


def default_device() -> torch.device:
    """Device of objects constructed without an explicit ``device=`` (the reference builds its buffers that way,
    on_policy_base_runner.py:129-156): ``HARL_DEVICE`` if set, else ``cuda:<LOCAL_RANK>`` -- one process per GPU under
    torchrun, cuda:0 otherwise."""
    return torch.device(os.environ.get("HARL_DEVICE") or f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")


def require_gpu(device: torch.device) -> None:
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError(
            "harl_amd runs on MI355X (PyTorch-ROCm device 'cuda') only; there is no CPU path. "
            f"Got device={device}, torch.cuda.is_available()={torch.cuda.is_available()}")
    load()
