"""The algorithmic-byte models behind bench.py's roofline (harl_amd/traffic.py): argument positions against the C ABI
declarations in include/harl_hip.h, and a few hand-computed values."""
import os
import re

from harl_amd.traffic import ALGORITHMIC_BYTES, algorithmic_bytes

HERE = os.path.dirname(os.path.abspath(__file__))


def _params(name):
    src = open(os.path.join(HERE, "..", "include", "harl_hip.h")).read()
    m = re.search(r"int " + name + r"\((.*?)\);", src, re.S)
    assert m, name
    return [p.strip().split()[-1].lstrip("*") for p in " ".join(m.group(1).split()).split(",")]


def test_every_modelled_function_is_in_the_header():
    for name in ALGORITHMIC_BYTES:
        assert _params(name), name


def test_argument_positions_match_the_header():
    # (function, {argument name: index the model reads it at})
    expect = {
        "harl_mlp_fwd_fused2x": {"M": 1, "D": 3, "H": 7, "store1": 8},
        "harl_mlp_fwd_hidden": {"M": 1, "HI": 2, "HO": 3},
        "harl_mlp_fwd_wide": {"M": 1, "KP": 2, "H": 6},
        "harl_mlp_x0n_wide": {"idx": 2, "M": 3, "D": 4},
        "harl_mlp_bwd_dx": {"M": 4, "HO": 5, "HI": 6, "dz_prev": 8, "x0n": 9, "kp0": 10},
        "harl_mlp_bwd_dx_dw": {"M": 4, "HO": 5, "HI": 6, "dz_prev": 8, "x0n": 9, "kp0": 10},
        "harl_mlp_dw_partials": {"a_kind": 1, "lda": 2, "HO": 3, "K": 10, "M": 11},
        "harl_gru_fwd": {"H": 7, "L": 8, "m_pad": 9, "save": 18},
        "harl_gru_bwd": {"H": 9, "L": 10, "m_pad": 11},
        "harl_actor_head_loss": {"M": 3, "H": 4, "discrete": 10, "act_dim": 11, "avail": 14, "factor": 18, "active": 19,
                                 "logp_out": 26, "dhead": 28, "dw_part": 30},
        "harl_actor_head_logp": {"M": 1, "H": 2, "discrete": 8, "act_dim": 9, "avail": 11, "logp_out": 12, "old_logp": 13,
                                 "factor": 14, "head_out": 16},
        "harl_critic_head_loss": {"M": 3, "H": 4},
        "harl_update_fwd_actor": {"M": 1, "D": 2, "H": 3, "discrete": 13, "act_dim": 14, "avail": 16, "factor": 20,
                                  "active": 21, "logp_out": 26, "xh1": 31},
        "harl_update_fwd_critic": {"M": 1, "D": 2, "H": 3, "xh1": 21},
        "harl_update_last_actor": {"M": 1, "H": 2, "discrete": 10, "act_dim": 11, "avail": 14, "factor": 18, "active": 19,
                                   "logp_out": 24},
        "harl_update_last_critic": {"M": 1, "H": 2},
        "harl_update_logp": {"M": 1, "D": 2, "discrete": 13, "act_dim": 14, "avail": 16, "logp_out": 17, "old_logp": 18,
                             "factor": 19, "head_out": 21},
        "harl_update_bwd": {"M": 2, "D": 3, "H": 4},
        "harl_gae_returns": {"T": 8, "ncols": 9},
        "harl_mlp_panel_fwd": {"M": 1, "KP": 2, "HO": 6},
        "harl_mlp_panel_bwd": {"M": 4, "HO": 5, "HI": 6},
        "harl_mlp_tangent_hidden": {"M": 2, "HI": 3, "HO": 4},
    }
    for fn, pos in expect.items():
        names = _params(fn)
        for arg, i in pos.items():
            assert names[i] == arg, (fn, arg, i, names)


def test_hand_computed_values():
    B = 819200
    # hidden layer 128 -> 128: x_hat in (512 B) + x_hat out (512) + mask (16) + rstd (4) per row
    assert algorithmic_bytes("harl_mlp_fwd_hidden", (1, B, 128, 128, 1, 1, 1, 1, 1, 0)) == B * 1044.0
    # weight gradient of a hidden layer: dz (512 B) + x_hat (512 B)
    assert algorithmic_bytes("harl_mlp_dw_partials", (1, 0, 0, 128, 1, 0, 0, None, None, None, 128, B, 1, 512, 0)) == B * 1024.0
    # fused first two layers from the 128-byte input image, training mode: both activations + masks + statistics
    assert algorithmic_bytes("harl_mlp_fwd_fused2x", (1, B, 1, 18, 1, 1, 1, 128, 1, 1, 1, 1, 1, 1, 1, 0)) == B * (128 + 2 * 532.0)
    assert algorithmic_bytes("harl_no_such_kernel", ()) is None
