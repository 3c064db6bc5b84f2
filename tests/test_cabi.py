"""C-ABI checks that run without a GPU: the in-tree library loads, exports every symbol that
include/harl_hip.h declares, and the ctypes signatures in harl_amd/_lib.py agree with the header."""
import ctypes
import os
import re

import pytest

from harl_amd import _lib


def _header_decls(repo_root):
    src = open(os.path.join(repo_root, "include", "harl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|const char \*)\s*(harl_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(3).strip()
        arglist = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
        decls[m.group(2)] = arglist
    return decls


def _ctype_of(arg: str):
    if "*" in arg:
        return ctypes.c_void_p
    t = arg.rsplit(" ", 1)[0].replace("const", "").strip()
    return {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double}[t]


def test_library_builds_and_exports_all_header_symbols(repo_root):
    from harl_amd._build import build

    path = build()
    lib = ctypes.CDLL(path)
    decls = _header_decls(repo_root)
    assert len(decls) >= 20
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/harl_hip.h but not exported"
    lib.harl_version.restype = ctypes.c_int
    assert lib.harl_version() >= 100  # host-only call, no GPU needed


def test_ctypes_signatures_match_header(repo_root):
    decls = _header_decls(repo_root)
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in decls, name
        want = [_ctype_of(a) for a in decls[name]]
        assert want == argtypes, f"{name}: header {decls[name]} vs ctypes {argtypes}"
    missing = set(decls) - set(_lib.SIGNATURES) - {"harl_last_error"}
    assert not missing, missing


def test_product_does_not_import_oracle(repo_root):
    """The oracle is test infrastructure: nothing under harl_amd/ may import it."""
    for dirpath, _, files in os.walk(os.path.join(repo_root, "harl_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)


def test_reference_checkpoint_fixture_keys_match_parameter_layout(repo_root):
    """tests/golden/ref_ckpt/*.pt were written by the reference's save(): their key order and shapes are the parameter
    layout harl_amd.synthetic / harl_amd.nets assume, for a GRU actor and critic."""
    import torch
    from harl_amd.synthetic import Shapes, actor_param_shapes, critic_param_shapes
    sh = Shapes(T=4, N=4, A=2, obs_dim=9, share_obs_dim=12, act_dim=4, discrete=True, hidden_sizes=[64, 64])
    d = os.path.join(repo_root, "tests", "golden", "ref_ckpt")
    sd = torch.load(os.path.join(d, "actor_agent0.pt"), map_location="cpu")
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == actor_param_shapes(sh, True, True)
    sd = torch.load(os.path.join(d, "critic_agent.pt"), map_location="cpu")
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == critic_param_shapes(sh, True, True)
    sd = torch.load(os.path.join(d, "value_normalizer.pt"), map_location="cpu")
    assert list(sd.keys()) == ["running_mean", "running_mean_sq", "debiasing_term"]


def test_randperm_replay_is_bit_exact_with_torch():
    """buffers.draw_permutation (C replay of ATen's randperm_cpu on a copy of the mt19937 state) returns exactly
    torch.randperm's permutation and leaves the global CPU generator in exactly the same state -- the integer side of
    the parity contract (minibatch sampling bit-exact) for the large-batch path that bypasses torch.randperm."""
    import torch
    from harl_amd import buffers
    assert buffers._replay_matches_randperm()
    for seed in (0, 1, 123456789):
        for n in (65536, 70001, 204800, 819200):
            torch.manual_seed(seed)
            torch.randperm(5)  # off the freshly seeded state
            st = torch.get_rng_state()
            want = torch.randperm(n)
            want_state = torch.get_rng_state()
            torch.set_rng_state(st)
            taps = []
            buffers.PERM_TAP = taps.append
            try:
                got = buffers.draw_permutation(n)
            finally:
                buffers.PERM_TAP = None
            assert got.dtype == torch.int64 and torch.equal(got, want), (seed, n)
            assert torch.equal(torch.get_rng_state(), want_state), (seed, n)
            assert len(taps) == 1 and torch.equal(taps[0], want)
            # and the next draw of the stream is unaffected
            assert torch.equal(torch.randperm(7), (torch.set_rng_state(want_state), torch.randperm(7))[1])


def test_policy_init_rng_replay_matches_real_construction():
    """HATRPO's old-actor snapshot draws (hatrpo.py:127-130) are replayed without the QR of orthogonal_: the global CPU
    generator must end exactly where constructing the layers for real leaves it."""
    import torch
    import torch.nn as nn
    from harl_amd.nets import consume_policy_init_rng

    class Box:
        def __init__(self, shape):
            self.shape = shape

    def real(d, hs, n):
        g = nn.init.calculate_gain("relu")
        for h in hs:
            lin = nn.Linear(d, h)
            nn.init.orthogonal_(lin.weight.data, gain=g)
            d = h
        lin = nn.Linear(d, n)
        nn.init.orthogonal_(lin.weight.data, gain=0.01)

    for d, hs, n in [(393, [128, 128, 128], 1), (18, [64], 5), (70, [128, 64], 3)]:
        torch.manual_seed(5)
        real(d, hs, n)
        a = torch.get_rng_state()
        torch.manual_seed(5)
        consume_policy_init_rng(dict(initialization_method="orthogonal_", hidden_sizes=hs, gain=0.01), Box((d,)), Box((n,)))
        assert bool((a == torch.get_rng_state()).all())


def test_rng_advance_matches_tensor_random():
    """harl_rng_advance (mt19937 skip-ahead, host code) leaves the CPU generator exactly where Tensor.random_ over as many
    int32 elements does; consume_randperm (deferred: summed, applied by rng_sync in one jump) equals torch.randperm's advance."""
    import torch
    from harl_amd import _lib, buffers as B

    lib = _lib.load()
    for seed, n in ((1, 1), (2, 623), (3, 624), (4, 625), (5, 70000), (6, 819199)):
        torch.manual_seed(seed)
        torch.randperm(11)
        mid = torch.get_rng_state()
        torch.empty(n, dtype=torch.int32).random_()
        want = torch.get_rng_state()
        out = torch.empty_like(mid)
        assert lib.harl_rng_advance(mid.data_ptr(), mid.numel(), n, out.data_ptr()) == 0
        assert bool(torch.equal(out, want)), (seed, n)
    torch.manual_seed(9)
    for _ in range(3):
        torch.randperm(819200)
    a = torch.randperm(9)
    torch.manual_seed(9)
    for _ in range(3):
        B.consume_randperm(819200)
    B.rng_sync()
    assert bool(torch.equal(a, torch.randperm(9)))


def test_rng_jump_ahead_is_bit_identical_and_constant_time(monkeypatch):
    """The GF(2) polynomial jump (harl_rng_jump; harl_rng_advance above ~1.5 M draws) leaves the generator exactly where drawing
    does -- every word of the state block, from mid-block starts, on all three instruction-set paths -- and a whole train()'s
    worth of deferred sampler advances at the 8-GPU global batch (20 x randperm(200 x 32768)) costs one jump: < 1 ms of host
    time once the polynomial of that total is cached (round 2: 20 x 0.8 ms on a replay thread)."""
    import time

    import torch
    from harl_amd import _lib, buffers as B

    lib = _lib.load()
    for isa in ("", "avx2", "base"):
        monkeypatch.setenv("HARL_RNG_ISA", isa)
        for seed, pre, n in ((1, 17, 624 * 3000 + 5), (5, 0, 1700000), (9, 623, 624 * 2600), (11, 624, 2000001)):
            torch.manual_seed(seed)
            if pre:
                torch.empty(pre, dtype=torch.int32).random_()
            mid = torch.get_rng_state()
            torch.empty(n, dtype=torch.int32).random_()
            want = torch.get_rng_state()
            for fn in (lib.harl_rng_jump, lib.harl_rng_advance):
                out = torch.empty_like(mid)
                assert fn(mid.data_ptr(), mid.numel(), n, out.data_ptr()) == 0
                assert bool(torch.equal(out, want)), (isa, seed, pre, n, fn.__name__)
    monkeypatch.setenv("HARL_RNG_ISA", "")
    n_global = 200 * 32768
    torch.manual_seed(3)
    torch.randperm(5)
    start = torch.get_rng_state()
    for _ in range(20):
        torch.empty(n_global - 1, dtype=torch.int32).random_()
    want = torch.get_rng_state()
    best = 1e9
    for rep in range(4):  # rep 0 builds the jump polynomial of this total (~20 ms, cached)
        torch.set_rng_state(start)
        t0 = time.perf_counter()
        for _ in range(20):
            B.consume_randperm(n_global)
        B.rng_sync()
        dt = time.perf_counter() - t0
        assert bool(torch.equal(torch.get_rng_state(), want))
        if rep:
            best = min(best, dt)
    print(f"20 deferred sampler advances of {n_global} draws + rng_sync: {best * 1e3:.3f} ms")
    assert best < 1e-3, best


def test_host_rng_instruction_set_paths_agree(monkeypatch):
    """The AVX-512 / AVX2 / baseline variants of the mt19937 host loops (HARL_RNG_ISA caps the dispatch) give the same
    permutation and generator state as torch for a block-crossing size."""
    import torch
    from harl_amd import _lib

    lib = _lib.load()
    n = 70001
    torch.manual_seed(21)
    torch.empty(33, dtype=torch.int32).random_()
    mid = torch.get_rng_state()
    want_perm = torch.randperm(n)
    want_state = torch.get_rng_state()
    for isa in ("", "avx2", "base"):
        monkeypatch.setenv("HARL_RNG_ISA", isa)
        out = torch.empty(n, dtype=torch.int32)
        scratch = torch.empty(n, dtype=torch.int32)
        st = torch.empty_like(mid)
        assert lib.harl_randperm_replay(mid.data_ptr(), mid.numel(), n, out.data_ptr(), scratch.data_ptr(), st.data_ptr()) == 0
        assert bool(torch.equal(out.long(), want_perm)) and bool(torch.equal(st, want_state)), isa
        st2 = torch.empty_like(mid)
        assert lib.harl_rng_advance(mid.data_ptr(), mid.numel(), n - 1, st2.data_ptr()) == 0
        assert bool(torch.equal(st2, want_state)), isa


def test_update_supported_accounts_for_lds():
    """harl_update_supported(D, H, act_dim, kind): the instantiated range AND the 160 KiB of LDS a workgroup can have
    (kind 0 forward-only, 1 actor step, 2 critic step).  No GPU needed: pure host arithmetic of the library."""
    from harl_amd import _lib
    f = _lib.load().harl_update_supported
    assert f(18, 128, 5, 1) == 1 and f(18, 128, 5, 0) == 1          # the MPE actor
    assert f(54, 128, 1, 2) == 1 and f(64, 128, 1, 2) == 1          # critics with up to 64 inputs
    assert f(40, 128, 5, 1) == 0 and f(40, 128, 4, 1) == 0          # actor step, 33..64 inputs, 128 wide: 183 / 165 KiB
    assert f(40, 128, 5, 0) == 1                                    # ... its log-prob passes fit
    assert f(40, 64, 8, 1) == 1 and f(0, 128, 8, 1) == 1 and f(0, 64, 3, 2) == 1
    assert f(65, 128, 5, 1) == 0 and f(18, 96, 5, 1) == 0 and f(18, 128, 9, 1) == 0


def test_comm_entry_points_validate_arguments_and_fail_loudly_without_a_gpu():
    """harl_comm_* (csrc/comm.hip): bad arguments are refused before anything is allocated; without a GPU the allocation fails
    with an error text instead of a crash (no compute call is made here)."""
    lib = _lib.load()
    handle = ctypes.create_string_buffer(64)
    ctx = ctypes.c_void_p()
    assert lib.harl_comm_create(17, 0, 1024, 8, handle, ctypes.byref(ctx)) == -2      # more ranks than slots
    assert b"harl_comm_create" in lib.harl_last_error()
    assert lib.harl_comm_create(2, 2, 1024, 8, handle, ctypes.byref(ctx)) == -2       # rank outside the world
    assert lib.harl_comm_create(2, 0, 1024, 64, handle, ctypes.byref(ctx)) == -2      # more blocks than flag columns
    assert lib.harl_comm_allreduce(None, None, 4, 0, None) == -2
    assert lib.harl_comm_connect(None, None) == -2
    assert lib.harl_comm_destroy(None) == 0
    import torch
    if not torch.cuda.is_available():
        assert lib.harl_comm_create(2, 0, 1024, 8, handle, ctypes.byref(ctx)) < 0 and ctx.value is None


def test_no_entry_point_but_create_allocates_device_memory(repo_root):
    """include/harl_hip.h promises "no allocation, no host synchronisation inside ... safe under hipGraph capture" (VERDICT r05
    weak 13: a lazily allocating scratch pool broke it).  Every device / pinned allocation call in the HIP sources must sit in
    the body of a `harl_*_create` entry point; scratch memory is the caller's (HARL_*_SCRATCH_BYTES)."""
    csrc = os.path.join(repo_root, "harl_amd", "csrc")
    alloc = re.compile(r"\b(hipMalloc\w*|hipExtMalloc\w*|hipHostMalloc|hipHostAlloc|hipMallocManaged|hipMemPoolCreate)\s*\(")
    offenders = []
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".h")):
            continue
        src = open(os.path.join(csrc, fn)).read()
        src = re.sub(r"//[^\n]*", "", src)
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        # function heads start at column 0 (bodies are indented): the allocation belongs to the nearest head in front of it
        heads = [(m.start(), m.group(1)) for m in re.finditer(r"^(?![\s#}])[^\n(]*?\b(\w+)\s*\(", src, flags=re.M)]
        for m in alloc.finditer(src):
            owner = next((name for pos, name in reversed(heads) if pos < m.start()), "?")
            if not (owner.startswith("harl_") and owner.endswith("_create")):
                offenders.append(f"{fn}: {m.group(1)} inside {owner}")
    assert not offenders, offenders
    hdr = open(os.path.join(repo_root, "include", "harl_hip.h")).read()
    assert int(re.search(r"#define HARL_MM_SCRATCH_BYTES (\d+)", hdr).group(1)) == _lib.SCRATCH_BYTES["mm"]
    assert int(re.search(r"#define HARL_CG_SCRATCH_BYTES (\d+)", hdr).group(1)) == _lib.SCRATCH_BYTES["cg"]
