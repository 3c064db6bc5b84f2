"""Host logic of the single-minibatch recurrent sampler (harl_amd/buffers._recurrent_seqs, round 6): with ONE minibatch the
sequences are taken in buffer order, the global CPU generator advances exactly as the reference's ``torch.randperm`` draw does
(on_policy_actor_buffer.py:180-326), and the sequence table is built once per update (cache keyed on the source tensors)."""
import numpy as np
import pytest
import torch

from harl_amd import buffers


def _run(monkeypatch, ordered, T, N, L, naive, shard=None, agents=1, cache=None, seed=7):
    monkeypatch.setenv("HARL_RNN_ORDERED", "1" if ordered else "0")
    buffers.rng_sync()
    torch.manual_seed(seed)
    n_local = N if shard is None else shard[2] - shard[1]
    H = 4
    h0_src = torch.arange((T + 1) * n_local * agents * H, dtype=torch.float32).reshape(-1, H)
    masks = (torch.arange((T + 1) * n_local * agents) % 3 != 0).float()
    out = list(buffers._recurrent_seqs(torch.device("cpu"), T, n_local, agents, H, 1, L, naive, shard, h0_src, masks, cache=cache))
    buffers.rng_sync()
    return out, torch.get_rng_state(), (h0_src, masks)


@pytest.mark.parametrize("naive", [False, True])
@pytest.mark.parametrize("shard", [None, (6, 2, 5)])
def test_ordered_minibatch_same_sequences_same_generator_state(monkeypatch, naive, shard):
    T, N, L = 20, 6, 5
    (a,), st_a, _ = _run(monkeypatch, True, T, N, L, naive, shard)
    (b,), st_b, _ = _run(monkeypatch, False, T, N, L, naive, shard)
    assert torch.equal(st_a, st_b), "the generator must advance exactly as the materialised permutation does"
    assert a["L"] == b["L"] and a["m"] == b["m"] and a["m_global"] == b["m_global"]
    m = a["m"]
    first = lambda s: np.sort(s["idx"].reshape(s["L"], s["m_pad"])[0, :m].numpy())  # noqa: E731
    assert np.array_equal(first(a), first(b)), "the same sequences, in another order"
    ia = a["idx"].reshape(a["L"], a["m_pad"])[0, :m].numpy()
    assert np.array_equal(ia, np.sort(ia)) or not naive, "buffer order"
    # every sequence starts from ITS stored state and carries ITS masks
    order = np.argsort(b["idx"].reshape(b["L"], b["m_pad"])[0, :m].numpy())
    assert torch.equal(a["h0"][:m][np.argsort(ia)], b["h0"][:m][order])
    ma = a["mask_rows"].reshape(a["L"], a["m_pad"])[:, :m][:, np.argsort(ia)]
    mb = b["mask_rows"].reshape(b["L"], b["m_pad"])[:, :m][:, order]
    assert torch.equal(ma, mb)


def test_sequence_table_is_built_once_per_update(monkeypatch):
    T, N, L = 20, 6, 5
    monkeypatch.setenv("HARL_RNN_ORDERED", "1")
    H = 4
    h0_src = torch.zeros((T + 1) * N, H)
    masks = torch.ones((T + 1) * N)
    cache = {}
    run = lambda: next(iter(buffers._recurrent_seqs(torch.device("cpu"), T, N, 1, H, 1, L, False, None, h0_src, masks, cache=cache)))  # noqa: E731
    s1 = run()
    s2 = run()
    assert s2 is s1 and s2["idx"] is s1["idx"], "second epoch: the same table object (the networks' input image is keyed on it)"
    masks.mul_(1.0)  # an in-place write bumps the version counter: the table is rebuilt
    s3 = run()
    assert s3 is not s1
    cache.clear()  # what every train() entry point does
    assert run() is not s3
    buffers.rng_sync()


def test_two_minibatches_still_materialise_the_permutation(monkeypatch):
    monkeypatch.setenv("HARL_RNN_ORDERED", "1")
    buffers.rng_sync()  # (deferred advances of an earlier test belong to ITS generator state)
    torch.manual_seed(3)
    T, N, L, H = 20, 6, 5, 4
    ref = torch.randperm((T * N) // L)
    torch.manual_seed(3)
    h0_src = torch.zeros((T + 1) * N, H)
    masks = torch.ones((T + 1) * N)
    seqs = list(buffers._recurrent_seqs(torch.device("cpu"), T, N, 1, H, 2, L, False, None, h0_src, masks, cache={}))
    assert len(seqs) == 2
    start = ref.numpy() * L
    want = (start % T) * N + start // T
    got = np.concatenate([s["idx"].reshape(s["L"], s["m_pad"])[0, :s["m"]].numpy() for s in seqs])
    assert np.array_equal(got, want)
