"""Per-phase shader cycles of the persistent kernels' slab loops (debug library built with -DHARL_PHASE_TIMING):

    python -c "from harl_amd import _build as B; ex=dict(B.DEFAULT_EXTRA); [ex.__setitem__(f, ex.get(f, [])+['-DHARL_PHASE_TIMING']) for f in ('mlp.hip','wide.hip','heads.hip','update.hip')]; B.build(variant='phase', extra=ex)"
    HARL_LIB=phase python tools/phase_cycles.py            (on the MI355X box)

Every instrumented kernel sums s_memtime deltas per phase over the slabs of wave 0 of workgroup 0 (csrc/common.h PHASE macros);
this script runs one MPE update in the layer mode (HARL_FUSED_UPDATE=logp) and one in the default hybrid mode (fused forward,
layer backward) and prints the tables.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from harl_amd import _lib  # noqa: E402

NPH = 12
NAMES = {
    ("mlp", 0): ("k_bwd_dx<KT>0> (bwd_dx_dw1)", ["split dz", "issue loads", "GEMM", "LN bwd + store", "dW1"]),
    ("mlp", 1): ("k_bwd_dx<KT=0>", ["split dz", "issue loads", "GEMM", "LN bwd + store"]),
    ("mlp", 3): ("k_bwd_dx_dw (bwd_full[_dw1]: dx + dW' + dW1' in one launch)",
                 ["issue loads", "GEMM dx", "LN bwd (+ store)", "dW1", "D-part loads issue", "barrier", "MFMA rounds + fillers", "B stores"]),
    ("mlp", 2): ("k_dw_tr", ["barrier 1", "split + store", "barrier 2", "MFMA phase"]),
    ("wide", 0): ("k_fwd_fused2x", ["split x0n + load", "GEMM 1", "relu/LN 1 + store", "split x1", "GEMM 2", "relu/LN 2 + store"]),
    ("heads", 0): ("k_actor_head<TRAIN, FUSE>", ["row loads + x load issue", "head fwd", "sample / loss", "head dW", "head bwd + store"]),
    ("heads", 1): ("k_actor_head<other>", ["load", "head fwd", "sample", "-", "-"]),
    ("update", 0): ("k_upd_fwd actor TRAIN", ["rows + split x0n", "GEMM 1", "relu/LN 1 + split", "GEMM 2", "relu/LN 2", "head fwd",
                                                "sample / loss", "head dW", "head bwd + store"]),
    ("update", 1): ("k_upd_fwd critic TRAIN", ["rows + split x0n", "GEMM 1", "relu/LN 1 + split", "GEMM 2", "relu/LN 2", "head fwd",
                                                 "sample / loss", "head dW", "head bwd + store"]),
    ("update", 2): ("k_upd_fwd actor logp", ["rows + split x0n", "GEMM 1", "relu/LN 1 + split", "GEMM 2", "relu/LN 2", "head fwd", "sample"]),
}


def read(tu):
    lib = _lib.load()
    buf = (C.c_longlong * (8 * NPH))()
    fn = getattr(lib, "harl_phase_read_" + tu)
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    assert fn(buf) == 0
    return [[buf[s * NPH + k] for k in range(NPH)] for s in range(8)]


def show(tus):
    for tu in tus:
        tab = read(tu)
        for slot in range(8):
            if (tu, slot) not in NAMES or sum(tab[slot]) == 0:
                continue
            name, ph = NAMES[(tu, slot)]
            tot = sum(tab[slot])
            print(f"## {name}: {tot} cycles, wave 0 / workgroup 0, kernel entry to exit")
            for k, p in list(enumerate(ph)) + [(10, "PROLOGUE (before the loop)"), (11, "EPILOGUE (after the loop)")]:
                if tab[slot][k]:
                    print(f"   {p:28s} {tab[slot][k]:10d}  {100.0 * tab[slot][k] / tot:5.1f} %")


def show_wg(tus):
    """Per workgroup (wave 0): loop cycles, start skew, end spread, wall time of the slowest, shader clock under load."""
    lib = _lib.load()
    for tu in tus:
        buf = (C.c_longlong * (8 * 256 * 3))()
        fn = getattr(lib, "harl_phase_read_wg_" + tu)
        fn.argtypes = [C.c_void_p]
        fn.restype = C.c_int
        assert fn(buf) == 0
        for slot in range(8):
            if (tu, slot) not in NAMES:
                continue
            rows = [(buf[(slot * 256 + g) * 3], buf[(slot * 256 + g) * 3 + 1], buf[(slot * 256 + g) * 3 + 2]) for g in range(256)]
            rows = [r for r in rows if r[0] > 0]
            if not rows:
                continue
            cyc = sorted(r[0] for r in rows)
            t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
            dur = sorted((r[2] - r[1]) / 100.0 for r in rows)  # us (s_memrealtime: 100 MHz)
            clk = sorted(r[0] / ((r[2] - r[1]) * 10.0) for r in rows if r[2] > r[1])  # GHz
            print(f"## {NAMES[(tu, slot)][0]}: {len(rows)} workgroups (last launch)")
            print(f"   loop cycles   min {cyc[0]}  median {cyc[len(cyc) // 2]}  max {cyc[-1]}  (max/min {cyc[-1] / cyc[0]:.3f})")
            print(f"   loop time us  min {dur[0]:.1f}  median {dur[len(dur) // 2]:.1f}  max {dur[-1]:.1f}")
            print(f"   first loop start -> last loop end {(t1 - t0) / 100.0:.1f} us; start skew {(max(r[1] for r in rows) - t0) / 100.0:.1f} us; "
                  f"end spread {(t1 - min(r[2] for r in rows)) / 100.0:.1f} us")
            print(f"   shader clock  min {clk[0]:.2f}  median {clk[len(clk) // 2]:.2f}  max {clk[-1]:.2f} GHz")


def main():
    assert os.environ.get("HARL_LIB") == "phase", "run with HARL_LIB=phase"
    dev = torch.device("cuda:0")
    w = bench.WORKLOADS["mpe"]
    for mode in ("logp", "hybrid"):
        os.environ["HARL_FUSED_UPDATE"] = mode
        r = bench.build_gpu_runner(w, w["N"], 0, 1, dev)
        bench.one_step(r)
        torch.cuda.synchronize()
        print(f"# HARL_FUSED_UPDATE={mode}")
        show(("mlp", "wide", "heads") if mode == "logp" else ("update", "mlp"))
        if "--wg" in sys.argv:
            show_wg(("mlp",) if mode == "logp" else ("update", "mlp"))
        del r


if __name__ == "__main__":
    main()
