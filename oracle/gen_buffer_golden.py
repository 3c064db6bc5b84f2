#!/usr/bin/env python3
"""Golden SLOT CONTENTS of the rollout buffers, recorded from the REAL reference (test infrastructure; build container only).

For each case the reference's own ``OnPolicyBaseRunner.insert`` (harl/runners/on_policy_base_runner.py:342-460), called
unbound on a runner built with ``__new__`` around the reference's real buffers, is fed T + 2 seeded environment steps
(dones of single agents and of whole environments, ``bad_transition`` infos, availability masks, hidden states), with
``after_update()`` of every buffer (on_policy_actor_buffer.py:88-96, critic buffers :85-93) after step T as the training
loop does.  Stored: every buffer array right before after_update and at the end.  The inputs are regenerated from the
seed by ``step_inputs`` (shared with tests/gpu_checks.check_buffer_slots), so the fixtures hold outputs only.

    python oracle/gen_buffer_golden.py        # writes tests/golden/buffers/<case>.npz
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HARL_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

CASES = {
    # name: (state_type, discrete, recurrent_n, T, N, A, obs, sobs, act, hidden)
    "ep_box": ("EP", False, 1, 5, 6, 3, 7, 11, 2, 8),
    "ep_disc_avail": ("EP", True, 1, 4, 5, 2, 6, 9, 5, 8),
    "fp_box": ("FP", False, 1, 5, 6, 3, 7, 11, 2, 8),
    "fp_disc_avail": ("FP", True, 1, 4, 5, 2, 6, 9, 5, 8),
}
ACTOR_KEYS = ("obs", "rnn_states", "actions", "action_log_probs", "masks", "active_masks", "available_actions")
CRITIC_KEYS = ("share_obs", "rnn_states_critic", "value_preds", "rewards", "masks", "bad_masks")


def step_inputs(case: str, step: int):
    """The data tuple of environment step ``step`` (the reference's layout, :344-356), a pure function of (case, step)."""
    st, disc, rn, T, N, A, D, S, act, H = CASES[case]
    rng = np.random.default_rng(1000 * (sorted(CASES).index(case) + 1) + step)
    f32 = np.float32
    obs = rng.standard_normal((N, A, D)).astype(f32)
    share = rng.standard_normal((N, A, S)).astype(f32)
    rewards = rng.standard_normal((N, A, 1)).astype(f32)
    dones = rng.random((N, A)) < 0.3
    dones[step % N] = True             # one whole environment done every step
    dones[(step + 2) % N] = False      # and one fully alive
    infos = [[({"bad_transition": True} if rng.random() < 0.3 else ({"bad_transition": False} if rng.random() < 0.3 else {}))
              for _ in range(A)] for _ in range(N)]
    avail = (rng.random((N, A, act)) < 0.7).astype(f32) if disc else np.array([None] * N)
    values = rng.standard_normal((N, 1) if st == "EP" else (N, A, 1)).astype(f32)
    if disc:
        actions = rng.integers(0, act, (N, A, 1)).astype(f32)
        logp = rng.standard_normal((N, A, 1)).astype(f32)
    else:
        actions = rng.standard_normal((N, A, act)).astype(f32)
        logp = rng.standard_normal((N, A, act)).astype(f32)
    rnn = rng.standard_normal((N, A, rn, H)).astype(f32)
    rnn_c = rng.standard_normal((N, rn, H) if st == "EP" else (N, A, rn, H)).astype(f32)
    return obs, share, rewards, dones, infos, avail, values, actions, logp, rnn, rnn_c


def snapshot(actor_bufs, critic_buf, get=lambda x: np.asarray(x)) -> dict:
    out = {}
    for a, b in enumerate(actor_bufs):
        for k in ACTOR_KEYS:
            v = getattr(b, k, None)
            if v is not None:
                out[f"actor{a}_{k}"] = get(v).copy()
    for k in CRITIC_KEYS:
        out[f"critic_{k}"] = get(getattr(critic_buf, k)).copy()
    return out


def main():
    sys.path.insert(0, REF)
    for name, mod in (("absl", None), ("setproctitle", None)):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "absl":
                fl = types.ModuleType("absl.flags")
                fl.FLAGS = lambda argv: argv
                m.flags = fl
                sys.modules["absl.flags"] = fl
            else:
                m.setproctitle = lambda s: None
            sys.modules[name] = m
    from harl.common.buffers.on_policy_actor_buffer import OnPolicyActorBuffer
    from harl.common.buffers.on_policy_critic_buffer_ep import OnPolicyCriticBufferEP
    from harl.common.buffers.on_policy_critic_buffer_fp import OnPolicyCriticBufferFP
    from harl.runners.on_policy_base_runner import OnPolicyBaseRunner

    Box = type("Box", (), {"__init__": lambda self, shape: setattr(self, "shape", shape)})
    Discrete = type("Discrete", (), {"__init__": lambda self, n: setattr(self, "n", n)})
    out_dir = os.path.join(REPO, "tests", "golden", "buffers")
    os.makedirs(out_dir, exist_ok=True)
    for case, (st, disc, rn, T, N, A, D, S, act, H) in CASES.items():
        args = dict(episode_length=T, n_rollout_threads=N, hidden_sizes=[H, H], recurrent_n=rn, gamma=0.99, gae_lambda=0.95,
                    use_gae=True, use_proper_time_limits=True)
        space = Discrete(act) if disc else Box((act,))
        r = OnPolicyBaseRunner.__new__(OnPolicyBaseRunner)
        r.num_agents, r.recurrent_n, r.rnn_hidden_size, r.state_type = A, rn, H, st
        r.algo_args = {"train": {"n_rollout_threads": N}}
        r.actor_buffer = [OnPolicyActorBuffer(args, Box((D,)), space) for _ in range(A)]
        r.critic_buffer = (OnPolicyCriticBufferEP(args, Box((S,))) if st == "EP" else OnPolicyCriticBufferFP(args, Box((S,)), A))
        res = {}
        for step in range(T + 2):
            data = step_inputs(case, step)
            OnPolicyBaseRunner.insert(r, tuple(x.copy() if isinstance(x, np.ndarray) else x for x in data))
            if step == T - 1:
                res.update({f"full_{k}": v for k, v in snapshot(r.actor_buffer, r.critic_buffer).items()})
                for b in r.actor_buffer:
                    b.after_update()
                r.critic_buffer.after_update()
        res.update({f"end_{k}": v for k, v in snapshot(r.actor_buffer, r.critic_buffer).items()})
        np.savez_compressed(os.path.join(out_dir, f"{case}.npz"), **res)
        print(case, len(res), "arrays", sum(v.nbytes for v in res.values()), "bytes")


if __name__ == "__main__":
    main()
