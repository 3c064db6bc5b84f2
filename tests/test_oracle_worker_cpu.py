"""The worker-process path of the full-size parity checks (tests/gpu_checks._oracle_bench_runs) without a GPU: the same
oracle run in this process and in a `python -m tests.oracle_worker` child must agree bit for bit."""
import numpy as np
import torch


def _payload(n_threads=4):
    import bench
    from harl_amd.synthetic import Shapes, actor_param_shapes, critic_param_shapes, make_buffers, synthetic_state_dict
    w = bench.WORKLOADS["mpe"]
    sh = Shapes(T=w["T"], N=n_threads, A=w["A"], obs_dim=w["obs"], share_obs_dim=w["sobs"], act_dim=w["act"], discrete=False,
                hidden_sizes=w["hidden"])
    d = make_buffers(sh, seed=5)
    t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}  # noqa: E731
    return dict(workload="mpe", n_threads=n_threads,
                actor_sd=[t(synthetic_state_dict(actor_param_shapes(sh, True, False), 10 + a)) for a in range(w["A"])],
                critic_sd=t(synthetic_state_dict(critic_param_shapes(sh, True, False), 99)),
                abuf=[dict(obs=d.obs[a], actions=d.actions[a], logp=d.action_log_probs[a], masks=d.masks[a], active=d.active_masks[a])
                      for a in range(w["A"])],
                cbuf=dict(share_obs=d.share_obs, rewards=d.rewards, value_preds=d.value_preds, masks=d.critic_masks,
                          bad_masks=d.bad_masks),
                st0=np.zeros(3, dtype=np.float32), rng0=torch.get_rng_state(),
                next_value_hip=d.value_preds[-1].copy())


def test_worker_process_matches_in_process_run(monkeypatch):
    from tests import gpu_checks as G
    torch.manual_seed(3)
    pl = _payload()
    plan = [("f32", "f32", None), ("pert0", "f32", 977)]
    monkeypatch.setenv("HARL_ORACLE_THREADS", "2")  # two workers at once on a small CI host
    monkeypatch.setenv("HARL_ORACLE_PARALLEL", "0")
    a = G._oracle_bench_runs(pl, plan, False)
    monkeypatch.setenv("HARL_ORACLE_PARALLEL", "force")
    b = G._oracle_bench_runs(pl, plan, False)
    for tag in ("f32", "pert0"):
        assert np.array_equal(a[tag]["returns"], b[tag]["returns"])
        for x, y in zip(a[tag]["atr"], b[tag]["atr"]):
            assert np.array_equal(x, y)
        assert np.array_equal(a[tag]["cfin"], b[tag]["cfin"])
        assert torch.equal(a[tag]["rng"], b[tag]["rng"])
    assert not np.array_equal(a["f32"]["cfin"], a["pert0"]["cfin"])  # the one-ulp twin really is another run


def test_recurrent_payload_runs_through_the_oracle(monkeypatch):
    """The payload format of the recurrent full-size check (SMAC shape: available actions, GRU states of actors and critic) on
    a few rollout threads: one in-process oracle run produces per-update traces for every agent and a finite value of slot T."""
    import bench
    from harl_amd.synthetic import Shapes, actor_param_shapes, critic_param_shapes, make_buffers, synthetic_state_dict
    from tests import gpu_checks as G
    w = bench.WORKLOADS["smac3s5z"]
    n = 4
    sh = Shapes(T=w["T"], N=n, A=w["A"], obs_dim=w["obs"], share_obs_dim=w["sobs"], act_dim=w["act"], discrete=True,
                hidden_sizes=w["hidden"])
    d = make_buffers(sh, seed=6, unavailable_p=0.3, rnn=True)
    t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}  # noqa: E731
    torch.manual_seed(4)
    pl = dict(workload="smac3s5z", n_threads=n,
              actor_sd=[t(synthetic_state_dict(actor_param_shapes(sh, True, True), 10 + a)) for a in range(w["A"])],
              critic_sd=t(synthetic_state_dict(critic_param_shapes(sh, True, True), 99)),
              abuf=[dict(obs=d.obs[a], actions=d.actions[a], logp=d.action_log_probs[a], masks=d.masks[a], active=d.active_masks[a],
                         avail=d.available_actions[a], rnn=d.rnn["actor"][a]) for a in range(w["A"])],
              cbuf=dict(share_obs=d.share_obs, rewards=d.rewards, value_preds=d.value_preds, masks=d.critic_masks,
                        bad_masks=d.bad_masks, rnn=d.rnn["critic"]),
              st0=np.zeros(3, dtype=np.float32), rng0=torch.get_rng_state(), next_value_hip=d.value_preds[-1].copy())
    monkeypatch.setenv("HARL_ORACLE_THREADS", "2")
    run = G._oracle_bench_run(pl, "f32", "f32", None, False)
    assert len(run["atr"]) == w["A"] and all(x.shape == (5, 4) and np.isfinite(x).all() for x in run["atr"])
    assert run["ctr"].shape == (5, 2) and np.isfinite(run["nv"]).all() and run["nv"].shape == (n, 1)


def _generic_payload(name, n, seed):
    import bench
    from harl_amd.synthetic import Shapes, actor_param_shapes, critic_param_shapes, make_buffers, synthetic_state_dict
    w = bench.WORKLOADS[name]
    rnn = bool(w.get("rnn"))
    sh = Shapes(T=w["T"], N=n, A=w["A"], obs_dim=w["obs"], share_obs_dim=w["sobs"], act_dim=w["act"], discrete=w["disc"],
                hidden_sizes=w["hidden"])
    d = make_buffers(sh, seed=seed, unavailable_p=w.get("unavailable_p", 0.0), rnn=rnn)
    t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}  # noqa: E731
    return w, dict(workload=name, n_threads=n,
                   actor_sd=[t(synthetic_state_dict(actor_param_shapes(sh, True, rnn), 10 + a)) for a in range(w["A"])],
                   critic_sd=t(synthetic_state_dict(critic_param_shapes(sh, True, rnn), 99)),
                   abuf=[dict(obs=d.obs[a], actions=d.actions[a], logp=d.action_log_probs[a], masks=d.masks[a], active=d.active_masks[a],
                              avail=d.available_actions[a], rnn=d.rnn["actor"][a] if rnn else None) for a in range(w["A"])],
                   cbuf=dict(share_obs=d.share_obs, rewards=d.rewards, value_preds=d.value_preds, masks=d.critic_masks,
                             bad_masks=d.bad_masks, rnn=d.rnn["critic"] if rnn else None),
                   st0=np.zeros(3, dtype=np.float32), rng0=torch.get_rng_state(), next_value_hip=d.value_preds[-1].copy())


def test_hatrpo_payloads_run_through_the_oracle(monkeypatch):
    """The HATRPO bench workloads' payloads (humanoid17: 17 agents, obs 393; hatrpo_gru128: 128-wide GRU, unavailable actions) on
    a couple of rollout threads: the oracle run of the full-size checks returns one line-search record per agent -- decision,
    fraction, the five statistics -- and the one-ulp twin is another run with the same record layout."""
    from tests import gpu_checks as G
    monkeypatch.setenv("HARL_ORACLE_THREADS", "4")
    for name, n in (("humanoid17", 2), ("hatrpo_gru128", 2)):
        torch.manual_seed(4)
        w, pl = _generic_payload(name, n, 6)
        run = G._oracle_bench_run(pl, "f32", "f32", None, False)
        twin = G._oracle_bench_run(pl, "pert0", "f32", 977, False)
        assert len(run["atr"]) == w["A"] and all(len(x) == 1 for x in run["atr"])
        for a in range(w["A"]):
            u = run["atr"][a][0]
            assert set(u) == set(G.TRPO_TRACE_KEYS) and isinstance(u["accepted"], bool)
            assert all(np.isfinite(float(v)) for v in u.values())
        assert run["ctr"].shape == (5, 2) and len(run["fin"]) == w["A"]
        assert any(not np.array_equal(x, y) for x, y in zip(run["fin"], twin["fin"]))


def test_hatrpo_teacher_forced_pieces_and_the_scheduler(monkeypatch):
    """The pieces of the HATRPO full-size checks (gpu_checks._oracle_trpo_piece) on two rollout threads: one agent's step from a
    given input factor -- fp32 in this process against the same payload through the core-slot scheduler's worker process, bit for
    bit -- and the critic piece with its replay of the generator draws."""
    from tests import gpu_checks as G
    from tests import oracle_sched
    monkeypatch.setenv("HARL_ORACLE_THREADS", "2")
    torch.manual_seed(4)
    w, pl = _generic_payload("hatrpo_gru128", 2, 6)
    T, n = w["T"], 2
    rng = np.random.default_rng(0)
    factor_in = (1.0 + 0.1 * rng.standard_normal((T, n, 1))).astype(np.float32)
    small = {k: pl["cbuf"][k] for k in ("rewards", "value_preds", "masks", "bad_masks")}
    small["rnn"] = pl["cbuf"]["rnn"]
    common = dict(workload="hatrpo_gru128", n_threads=n, st0=pl["st0"], rng0=pl["rng0"], next_value_hip=pl["next_value_hip"])
    agent_pl = dict(common, mode="agent", agent=3, actor_sd=pl["actor_sd"][3], abuf=pl["abuf"][3], cbuf=small, factor_in=factor_in)
    here = G._oracle_bench_run(agent_pl, "f32", "f32", None, False)
    assert set(here["trace"]) == set(G.TRPO_TRACE_KEYS) and here["factor_out"].shape == (T, n, 1)
    assert np.isfinite(here["factor_out"]).all() and not np.array_equal(here["factor_out"], factor_in.astype(np.float64))
    sched = oracle_sched.Scheduler(list(range(4)), slot=2)  # two slots of two logical CPUs, three jobs: one has to queue
    monkeypatch.setattr(oracle_sched, "ACTIVE", sched)
    try:
        handle = G._oracle_launch(agent_pl, [("f32", "f32", None), ("f64", "f64", None), ("pert0", "f32", 977)], False)
        runs = G._oracle_collect(handle)
    finally:
        sched.shutdown()
    assert np.array_equal(runs["f32"]["factor_out"], here["factor_out"]) and np.array_equal(runs["f32"]["fin"], here["fin"])
    assert runs["f32"]["trace"] == here["trace"]
    assert not np.array_equal(runs["pert0"]["fin"], here["fin"])
    shapes = [(k, tuple(v.shape)) for k, v in pl["actor_sd"][0].items()]
    crit = G._oracle_bench_run(dict(common, mode="critic", critic_sd=pl["critic_sd"], cbuf=pl["cbuf"], actor_shapes=shapes),
                               "f32", "f32", None, False)
    assert crit["ctr"].shape == (5, 2) and crit["nv"].shape == (n, 1) and np.isfinite(crit["returns"]).all()
    assert not torch.equal(crit["rng"], pl["rng0"])


def test_teacher_forcing_hook_reproduces_a_free_run_when_fed_its_own_states(monkeypatch):
    """The forcing hook of the full-size HAPPO checks (gpu_checks._oracle_bench_run, payload['forced']): fed the states a
    free-running oracle run went through -- parameters and Adam moments in front of every optimiser step -- the forced run must
    reproduce that run bit for bit; fed states that are off by a factor, it must not."""
    from oracle import harl_oracle as O
    from tests import gpu_checks as G
    monkeypatch.setenv("HARL_ORACLE_THREADS", "2")
    torch.manual_seed(3)
    pl = _payload(2)
    taps = {}

    def record(stage, obj, sample, _vn):
        if stage != "pre":
            return
        flat = torch.cat([p.detach().reshape(-1) for p in obj.net.params()]).numpy().copy()
        st = obj.net.opt.state
        if len(st) == 0:
            m = v = np.zeros_like(flat)
            step = 0
        else:
            m = torch.cat([st[p]["exp_avg"].reshape(-1) for p in obj.net.params()]).numpy().copy()
            v = torch.cat([st[p]["exp_avg_sq"].reshape(-1) for p in obj.net.params()]).numpy().copy()
            step = int(st[obj.net.params()[0]]["step"])
        taps.setdefault(id(obj), []).append((flat, m, v, step))

    real_ha_train = O.ha_train
    order = []

    def spy(actors, critic, *a, **k):
        order[:] = [id(x) for x in actors] + [id(critic)]
        O.GRAD_HOOK = record
        try:
            return real_ha_train(actors, critic, *a, **k)
        finally:
            O.GRAD_HOOK = None

    monkeypatch.setattr(O, "ha_train", spy)
    free = G._oracle_bench_run(pl, "f32", "f32", None, False)
    monkeypatch.setattr(O, "ha_train", real_ha_train)
    snaps = [taps[i] for i in order]
    assert all(len(s) == 5 for s in snaps) and snaps[0][1][3] == 1
    pl["forced"] = dict(actor=snaps[:-1], critic=snaps[-1])
    forced = G._oracle_bench_run(pl, "f32", "f32", None, False)
    for x, y in zip(free["atr"], forced["atr"]):
        assert np.array_equal(x, y)
    assert np.array_equal(free["ctr"], forced["ctr"]) and np.array_equal(free["cfin"], forced["cfin"])
    assert all(np.array_equal(x, y) for x, y in zip(free["fin"], forced["fin"]))
    bad = dict(pl, forced=dict(actor=[[(f * 1.01, m, v, k) for f, m, v, k in s] for s in snaps[:-1]], critic=snaps[-1]))
    off = G._oracle_bench_run(bad, "f32", "f32", None, False)
    assert not np.array_equal(off["atr"][0], free["atr"][0])
