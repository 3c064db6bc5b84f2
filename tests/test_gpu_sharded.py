"""-m gpu: the data-parallel update path end to end.  Two ranks (gloo rendezvous, both on cuda:0 -- RCCL refuses two
ranks on one device, the collective semantics are the same) each own half of the rollout threads of a golden case; the
sharded train() must reproduce the reference's unsharded golden vectors: identical CPU-RNG stream, losses / grad-norms
/ final parameters within the end-of-train tolerance, and bit-identical parameters on both ranks."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, q, env=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    os.environ.update(env or {})
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from harl_amd.dist import Comm, shard_columns
        from harl_amd.runner import OnPolicyHARunner
        from tests import gpu_checks as G
        from tests.helpers import GoldenCase, rel_err, vec_rel_err

        case = GoldenCase(name)
        z, sh, d = case.z, case.shapes, case.data
        train, model, algo = case.reference_dicts()
        space = G.act_space_of(sh)
        comm = Comm()
        if (env or {}).get("HARL_ALLREDUCE") in ("oneshot", "auto"):  # (auto: all ranks share this GPU, every kind qualifies)
            assert comm.oneshot is not None and comm.oneshot_info["enabled"], comm.oneshot_info
        lo, hi = shard_columns(sh.N, rank, world)
        torch.manual_seed(case.seed)
        r = OnPolicyHARunner(dict(algo="happo"), dict(train=train, model=model, algo=algo), dict(state_type="EP"),
                             obs_spaces=[G.Box((sh.obs_dim,))] * sh.A, share_obs_space=G.Box((sh.share_obs_dim,)),
                             act_spaces=[space] * sh.A, device=G.DEV, comm=comm)
        assert (r.col_lo, r.col_hi) == (lo, hi)
        cut = lambda x: np.ascontiguousarray(x[:, lo:hi])  # noqa: E731
        for a in range(sh.A):
            r.actor[a].actor.load_state_dict({k: torch.from_numpy(v) for k, v in case.actor_sd[a].items()})
            b = r.actor_buffer[a]
            b.obs.copy_(G.dev(cut(d.obs[a])))
            b.actions.copy_(G.dev(cut(d.actions[a])))
            b.action_log_probs.copy_(G.dev(cut(d.action_log_probs[a])))
            b.masks.copy_(G.dev(cut(d.masks[a])))
            b.active_masks.copy_(G.dev(cut(d.active_masks[a])))
            if sh.discrete:
                b.available_actions.copy_(G.dev(cut(d.available_actions[a])))
            if d.rnn is not None:
                b.rnn_states.copy_(G.dev(cut(d.rnn["actor"][a])))
        if d.rnn is not None:
            r.critic_buffer.rnn_states_critic.copy_(G.dev(cut(d.rnn["critic"])))
        r.critic.critic.load_state_dict({k: torch.from_numpy(v) for k, v in case.critic_sd.items()})
        cb = r.critic_buffer
        for nm, arr in (("share_obs", d.share_obs), ("rewards", d.rewards), ("value_preds", d.value_preds),
                        ("masks", d.critic_masks), ("bad_masks", d.bad_masks)):
            getattr(cb, nm).copy_(G.dev(cut(arr)))
        if r.value_normalizer is not None:
            vi = case.vn_init
            r.value_normalizer.stats.copy_(G.dev(np.array([vi["running_mean"], vi["running_mean_sq"], vi["debiasing_term"]],
                                                          dtype=np.float32)))
        from harl_amd.buffers import _advance_matches_randperm
        assert _advance_matches_randperm()
        torch.manual_seed(case.seed + 12345)
        cb.compute_returns(cb.value_preds[-1].clone(), r.value_normalizer)
        ret_bad = float(np.sum(cb.returns.cpu().numpy()[:sh.T] != z["returns"][:sh.T, lo:hi]))
        r.prep_training()
        infos, cinfo = r.train()
        torch.cuda.synchronize()
        state_after = torch.get_rng_state()
        torch.manual_seed(case.seed + 12345)
        for g in case.perms():
            torch.randperm(len(g))
        res = dict(rank=rank, ret_bad=ret_bad, rng_bad=float(not torch.equal(state_after, torch.get_rng_state())))
        got = np.array([[i["policy_loss"], i["dist_entropy"], i["actor_grad_norm"], i["ratio"]] for i in infos])
        res["actor_rel"] = rel_err(got, z["actor_infos"])
        res["critic_rel"] = rel_err([cinfo["value_loss"], cinfo["critic_grad_norm"]], z["critic_info"])
        res["param_rel"] = max([vec_rel_err(r.actor[a].actor.flat_reference().cpu().numpy(), z[f"actor_final_{a}"])
                                for a in range(sh.A)] + [vec_rel_err(r.critic.critic.flat_param.cpu().numpy(), z["critic_final"])])
        # the bar of the unsharded golden tests (tests/helpers.excess): every entry within max(1e-5, 2 x the reference's own
        # measured fp32 uncertainty) of the reference's figure -- the sharded sums are another correct fp32 summation order
        from tests.helpers import excess, load_noise, vec_excess
        nz = load_noise(name)
        res["actor_excess"] = max(excess(got[:, c], z["actor_infos"][:, c], nz["actor_infos"][:, c], nz["sens_actor_infos"][:, c])
                                  for c in range(4))
        res["critic_excess"] = excess([cinfo["value_loss"], cinfo["critic_grad_norm"]], z["critic_info"], nz["critic_info"],
                                      nz["sens_critic_info"])
        res["param_excess"] = max([vec_excess(r.actor[a].actor.flat_reference().cpu().numpy(), z[f"actor_final_{a}"],
                                              nz[f"actor_final_{a}"], nz[f"sens_actor_final_{a}"]) for a in range(sh.A)]
                                  + [vec_excess(r.critic.critic.flat_param.cpu().numpy(), z["critic_final"], nz["critic_final"],
                                                nz["sens_critic_final"])])
        res["param_sum"] = float(sum(r.actor[a].actor.flat_param.double().sum().item() for a in range(sh.A)))
        res["oneshot_status"] = comm.oneshot_status()
        comm.close()
        q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["mpe_box_h128", "cheetah_h128x3_mb2", "rnn_disc_h64_mb2", "rnn_naive_h64", "md_h64_mb2",
                                  "md_lag_h128"])
def test_two_rank_sharded_train_matches_unsharded_golden(name):
    _run_sharded(name, None)


@pytest.mark.parametrize("name", ["mpe_box_h128", "rnn_disc_h64_mb2"])
def test_two_rank_sharded_train_through_oneshot_allreduce(name):
    """The same two-rank update with every exchange step on the hand-written one-hop all-reduce (csrc/comm.hip) instead of the
    backend's: each rank pushes its message into the other's hipIpc-mapped buffer (two processes sharing this GPU)."""
    _run_sharded(name, {"HARL_ALLREDUCE": "oneshot"})


def test_four_rank_sharded_train_matches_unsharded_golden_through_auto_exchange():
    """FOUR ranks (a quarter of the rollout threads each) with HARL_ALLREDUCE=auto -- the collectively decided one-hop exchange --
    against the reference's unsharded golden vectors; bit-identical replicated parameters on all four ranks."""
    _run_sharded("mpe_box_h128", {"HARL_ALLREDUCE": "auto"}, world=4)


def _run_sharded(name, env, world=2):
    import torch.multiprocessing as mp

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q, env)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r["ret_bad"] == 0.0 and r["rng_bad"] == 0.0 and r["oneshot_status"] == 0, r
        assert r["actor_excess"] <= 1.0 and r["critic_excess"] <= 1.0 and r["param_excess"] <= 1.0, r
    assert all(r["param_sum"] == res[0]["param_sum"] for r in res), "replicated parameters diverged between ranks"


def _oneshot_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from harl_amd.dist import Comm

        comm = Comm().enable_oneshot(cap_bytes=1 << 20)
        assert comm.oneshot is not None
        dev = torch.device("cuda:0")
        bad, cases = 0, 0
        # 25 153 floats = the MPE actor's message; 262 144 floats = the capacity; odd tails; fp64 moments (3 doubles)
        for it, (n, dt) in enumerate([(1, torch.float32), (3, torch.float64), (5, torch.float32), (1027, torch.float32),
                                      (25153, torch.float32), (262144, torch.float32), (131072, torch.float64),
                                      (20001, torch.float64)] * 3):
            g = torch.Generator().manual_seed(1000 * it + rank)
            x = (torch.randn(n, generator=g, dtype=torch.float64) * 10.0 ** float(torch.randint(-3, 4, (1,), generator=g))).to(dt)
            mine = x.to(dev)
            comm.all_reduce_sum(mine)
            parts = [torch.zeros_like(x) for _ in range(world)]
            dist.all_gather(parts, x)
            want = parts[0].clone()
            for p in parts[1:]:
                want += p  # rank order, the kernel's
            cases += 1
            bad += int(not torch.equal(mine.cpu(), want))
        # back-to-back launches with no host synchronisation in between (epochs alternate between the two buffer sets)
        y = torch.full((4096,), float(rank + 1), device=dev)
        reps = 200
        for _ in range(reps):
            comm.all_reduce_sum(y)
            y.mul_(1.0 / world)
        torch.cuda.synchronize()
        # launch-to-launch time of the MPE actor's message (25 153 floats) on this shared GPU: `world` processes time-share one
        # chip here, so this is an upper bound on the kernel's own cost, not an xGMI figure
        z = torch.zeros(25153 + 4 * 48, device=dev)
        for _ in range(10):
            comm.all_reduce_sum(z)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            comm.all_reduce_sum(z)
        e1.record()
        torch.cuda.synchronize()
        us_per_call = e0.elapsed_time(e1) * 10.0
        mean0 = sum(range(1, world + 1)) / world
        chain_ok = bool(torch.all(y == mean0).item())  # after the first launch every rank holds the mean, a fixed point
        # a tensor that is MISALIGNED on this rank only (a view one element into its storage on odd ranks) must still take the
        # exchange -- through an aligned staging copy -- not another backend (the path is chosen from rank-invariant properties)
        base = torch.arange(1030, dtype=torch.float32, device=dev) * (rank + 1)
        view = base[1:1025] if rank % 2 else base[:1024]
        want_v = sum((torch.arange(1030, dtype=torch.float32) * (q_ + 1))[1:1025] if q_ % 2 else
                     (torch.arange(1030, dtype=torch.float32) * (q_ + 1))[:1024] for q_ in range(world))
        comm.all_reduce_sum(view)
        bad += int(not torch.equal(view.cpu(), want_v))
        cases += 1
        st = comm.oneshot_status()
        kind = comm.oneshot[3]
        comm.close()
        q.put(dict(rank=rank, bad=bad, cases=cases, chain_ok=chain_ok, status=st, kind=kind, us_per_call=us_per_call))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_oneshot_allreduce_is_the_rank_ordered_sum_bit_for_bit(world):
    """harl_comm_allreduce among `world` processes sharing this GPU against the rank-ordered sum of the gathered messages: every
    bit, fp32 and fp64, lengths from 1 element to the capacity, and 200 unsynchronised launches in a row."""
    import torch.multiprocessing as mp

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_oneshot_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from tests.gpu_checks import dump_parity
    dump_parity(f"oneshot_allreduce_world{world}", dict(world=world, allocation_kind=res[0]["kind"],
                                                        us_per_call_max=max(r["us_per_call"] for r in res),
                                                        us_per_call_min=min(r["us_per_call"] for r in res),
                                                        mismatching_cases=sum(r["bad"] for r in res), cases=res[0]["cases"]))
    for r in res:
        assert r["bad"] == 0 and r["cases"] == 25 and r["chain_ok"] and r["status"] == 0, r


def _timeout_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HARL_ALLREDUCE="oneshot", HARL_ONESHOT_TIMEOUT_S="1.5")
    import time
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from harl_amd.dist import Comm
        comm = Comm()
        dev = torch.device("cuda:0")
        x = torch.full((4096,), float(rank + 1), device=dev)
        comm.all_reduce_sum(x)  # a healthy exchange first
        torch.cuda.synchronize()
        ok_first = bool(torch.all(x == 3.0).item()) and comm.oneshot_status() == 0
        res = dict(rank=rank, ok_first=ok_first)
        if rank == 0:  # rank 1 never launches the second exchange: rank 0 must give up after 1.5 s, not hang
            y = torch.ones(4096, device=dev)
            t0 = time.perf_counter()
            comm.all_reduce_sum(y)
            torch.cuda.synchronize()
            res.update(waited_s=time.perf_counter() - t0, all_nan=bool(torch.isnan(y).all().item()), status=comm.oneshot_status())
            try:
                comm.check()
                res["raised"] = False
            except RuntimeError as e:
                res["raised"] = "gave up waiting for rank 1" in str(e)
        dist.barrier()
        comm.close()
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_oneshot_timeout_is_loud_and_does_not_hang():
    """A peer that never shows up: after HARL_ONESHOT_TIMEOUT_S the waiting rank's launch ends, its result is NaN, the
    status word names the missing rank and Comm.check() -- what train() calls after its read-back -- raises."""
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_timeout_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r["rank"]: r for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["ok_first"] and res[1]["ok_first"], res
    r0 = res[0]
    assert r0["all_nan"] and r0["status"] == 2 and r0["raised"] is True and 1.0 < r0["waited_s"] < 30.0, r0
