"""CPU checks of the arithmetic the four-waves-per-slab GRU kernels (csrc/gru.hip: k_gru_fwd_q, k_gru_bwd_q) rely on: the
four-way merge of the LayerNorm partials and the ownership map between accumulator registers, k-steps of the next step's B
operand and float4 pieces of an ATL(64) image.  No GPU, no library call: the kernels themselves are covered by the -m gpu tests."""
import numpy as np


def feat(R: int, h: int) -> int:
    """common.h: register R of lane half h of a width-64 activation <-> feature f(R, h)."""
    return 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2) + 4 * h


def test_four_way_layernorm_merge_matches_the_direct_statistics():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1000, 64)) * rng.uniform(0.1, 5.0, (1000, 1)) + rng.uniform(-3, 3, (1000, 1))
    groups = []
    for w in range(2):
        for kh in range(2):  # wave (w, kh): registers 16 w + 8 kh .. + 7 of both lane halves = 16 features
            groups.append([feat(16 * w + 8 * kh + rr, h) for rr in range(8) for h in range(2)])
    assert sorted(sum(groups, [])) == list(range(64))  # the four waves' features partition the hidden vector
    mk = np.stack([x[:, g].mean(1) for g in groups])
    m2 = np.stack([((x[:, g] - x[:, g].mean(1, keepdims=True)) ** 2).sum(1) for g in groups])
    mean = 0.25 * mk.sum(0)                                   # k_gru_fwd_q: mean = sum mean_k / 4
    M2 = m2.sum(0) + 16.0 * ((mk - mean) ** 2).sum(0)         # M2 = sum M2_k + 16 sum (mean_k - mean)^2
    assert np.allclose(mean, x.mean(1), rtol=0, atol=1e-12)
    assert np.allclose(M2 / 64.0, x.var(1), rtol=1e-12, atol=1e-12)


def test_register_kstep_piece_ownership_of_the_four_waves():
    for w in range(2):
        for kh in range(2):
            regs = [16 * w + 8 * kh + rr for rr in range(8)]
            # forward: the wave's 8 registers are exactly k-step 2 w + kh of the B operand (k-step j = registers 8 j .. 8 j + 7) ...
            assert {r // 8 for r in regs} == {2 * w + kh}
            # ... and the float4 pieces 4 w + 2 kh, 4 w + 2 kh + 1 of the ATL(64) image (piece q = registers 4 q .. 4 q + 3)
            assert sorted({r // 4 for r in regs}) == [4 * w + 2 * kh, 4 * w + 2 * kh + 1]
            # its rows of the partial tiles: registers 8 kh .. 8 kh + 7 of row tile w (tile register r <-> activation register 16 w + r)
            assert [r - 16 * w for r in regs] == list(range(8 * kh, 8 * kh + 8))
    # backward: wave q owns registers 8 q .. 8 q + 7 of dr, dz and dhn = k-steps q, 4 + q, 8 + q of the 192-wide operand
    for q in range(4):
        ks = sorted({(32 * ty + 8 * q + rr) // 8 for ty in range(3) for rr in range(8)})
        assert ks == [q, 4 + q, 8 + q]
    assert sorted(k for q in range(4) for k in (q, 4 + q, 8 + q)) == list(range(12))


def test_partial_sums_over_k_halves_add_up_to_the_full_product():
    """The k-split of the forward: a dot product over 64 inputs as the sum of the partial sums over k-steps {0, 1} and {2, 3}
    (in the kernels' register order) is the full product up to fp32 rounding of one extra addition."""
    rng = np.random.default_rng(1)
    W = rng.standard_normal((192, 64)).astype(np.float32)
    hvec = rng.standard_normal(64).astype(np.float32)
    order = {h: [feat(R, h) for R in range(32)] for h in range(2)}
    full = W.astype(np.float64) @ hvec.astype(np.float64)
    part = np.zeros((2, 192))
    for kh in range(2):
        cols = [order[h][R] for R in range(16 * kh, 16 * kh + 16) for h in range(2)]  # k-steps 2 kh, 2 kh + 1: registers 16 kh .. + 15
        part[kh] = W[:, cols].astype(np.float64) @ hvec[cols].astype(np.float64)
    assert np.allclose(part.sum(0), full, rtol=1e-12, atol=1e-12)
