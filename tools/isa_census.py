"""Static instruction census of the gfx950 kernels (no GPU needed).

    python tools/isa_census.py [mlp heads gru ...]  > profiles/r02_isa_census.md

Compiles harl_amd/csrc/<name>.hip to gfx950 assembly with the library's own flags (`hipcc -S --cuda-device-only`) and, per
kernel, finds its STEADY-STATE LOOP -- the outermost backward branch that contains matrix instructions (the per-slab loop of
the persistent kernels; for kernels without MFMAs: the largest loop) -- and counts what one trip of it issues: MFMAs, VALU,
LDS, global memory, scalar, waits.  Cost model (round 3, s_memtime: tools/mfma_valu_overlap2.hip, tools/valu_cost.hip): a
32x32x16 bf16 MFMA holds the matrix pipe for 32.3 cycles (fp32 32x32x2: 64) but costs its wave ~12.5 cycles of issue, a plain
VALU / LDS instruction ~4.9 cycles (transcendentals ~9), and up to five of them issue for free in the shadow of an MFMA of the
same wave when they are INTERLEAVED with it.  Two floors per slab and wave:

    serial      = 32.3 n_bf16 + 64 n_f32 + 4.9 (n_valu + n_lds) + 9 n_transc      (phases one after the other: today's kernels)
    interleaved = max(32.3 n_bf16 + 64 n_f32,  12.5 (n_bf16 + n_f32) + 4.9 (n_valu + n_lds) + 9 n_transc)

which DESIGN.md section 3 compares with the measured cycles per slab (tools/phase_cycles.py).  Inner loops (weight staging, k-panels) are counted once
per textual occurrence, i.e. the floor is a LOWER bound for kernels with data-dependent inner trip counts; the report marks
them.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "harl_amd", "csrc")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")  # quarter-rate VALU


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        res = out.stdout.strip().split("\n")
        return res if len(res) == len(names) else names
    except OSError:
        return names


def parse_functions(path):
    """{symbol: [lines]} for every kernel entry (functions carrying .amdhsa_kernel descriptors)."""
    funcs, cur, name = {}, None, None
    kernels = set()
    with open(path) as f:
        for ln in f:
            s = ln.strip()
            m = re.match(r"\.amdhsa_kernel\s+(\S+)", s)
            if m:
                kernels.add(m.group(1))
            m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", s)
            if m and not s.startswith(".L"):
                name = m.group(1)
                cur = funcs.setdefault(name, [])
                continue
            if s.startswith(".Lfunc_end"):
                cur = None
                continue
            if cur is not None and s and not s.startswith(";"):
                cur.append(s)
    return {k: v for k, v in funcs.items() if k in kernels}


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma_f32" if ("f32_32x32x2" in op or "x2_f32" in op or "x1_f32" in op or op.endswith("_f32") and "bf16" not in op and "f16" not in op and "i8" not in op and "f8" not in op) else "mfma_bf16"
    if op.startswith("v_"):
        return "valu_t" if op.startswith(TRANS) else "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem_st" if ("store" in op or "atomic" in op) else "vmem_ld"
    if op.startswith("s_waitcnt") or op.startswith("s_wait"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def census(lines):
    """Steady-state loop = the widest backward branch containing an MFMA (else the widest backward branch)."""
    labels, instrs = {}, []
    for s in lines:
        m = re.match(r"^(\.L[\w$.]+):", s)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        tgt = None
        if op.startswith(("s_cbranch", "s_branch")):
            t = s.split()[-1]
            tgt = t if t.startswith(".L") else None
        instrs.append((op, tgt))
    kinds = [classify(op) for op, _ in instrs]
    loops = []
    for i, (op, tgt) in enumerate(instrs):
        if tgt is not None and tgt in labels and labels[tgt] <= i:
            lo = labels[tgt]
            n_m = sum(1 for k in kinds[lo:i + 1] if k.startswith("mfma"))
            loops.append((n_m > 0, i - lo, lo, i))
    total = {k: kinds.count(k) for k in set(kinds)}
    if not loops:
        return None, total, 0
    with_m = [lp for lp in loops if lp[0]]
    _, _, lo, hi = max(with_m or loops, key=lambda lp: lp[1])
    body = kinds[lo:hi + 1]
    inner = sum(1 for (_, _, a, b) in loops if a > lo and b < hi)
    return {k: body.count(k) for k in set(body)}, total, inner


def issue_slots(c):
    g = lambda k: c.get(k, 0)  # noqa: E731
    return 8 * g("mfma_bf16") + 16 * g("mfma_f32") + g("valu") + 4 * g("valu_t") + g("lds")


def main():
    names = sys.argv[1:] or ["mlp", "heads", "gru", "wide", "panel", "update", "multihead", "elementwise"]
    print("# Static instruction census of the gfx950 kernels (`tools/isa_census.py`, hipcc -O3 + the library's per-file flags, no GPU)\n")
    print("One trip of each kernel's steady-state loop (one 32-sample slab per wave unless noted).  `serial` / `interleaved` = "
          "cycles per trip under the round-3 cost model (module docstring: MFMA 32.3 / 64 cycles of pipe and 12.5 of issue, VALU / "
          "LDS 4.9, transcendentals 9; up to five VALU free per MFMA when interleaved).  `inner` = loops nested inside (counted "
          "once: lower bound).\n")
    sys.path.insert(0, REPO)
    from harl_amd._build import EXTRA_FLAGS
    with tempfile.TemporaryDirectory() as td:
        for n in names:
            src, asm = os.path.join(CSRC, n + ".hip"), os.path.join(td, n + ".s")
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                                src, "-o", asm] + EXTRA_FLAGS.get(n + ".hip", []), capture_output=True, text=True)
            if r.returncode != 0:
                print(f"## {n}.hip: hipcc failed\n```\n{r.stderr[-400:]}\n```")
                continue
            funcs = parse_functions(asm)
            syms = sorted(funcs)
            pretty = dict(zip(syms, demangle(syms)))
            print(f"## {n}.hip\n")
            print("| kernel | mfma bf16 | mfma f32 | valu | transc. | lds | vmem ld / st | waits | inner | serial cycles | interleaved cycles | matrix share of serial |")
            print("|---|---|---|---|---|---|---|---|---|---|---|---|")
            rows = []
            for sym in syms:
                body, total, inner = census(funcs[sym])
                if body is None:
                    continue
                g = lambda k: body.get(k, 0)  # noqa: E731
                sl = issue_slots(body)
                if sl < 50:
                    continue
                nm = re.sub(r"\(anonymous namespace\)::", "", pretty[sym])
                nm = re.sub(r"^void ", "", nm)
                nm = nm.split("(")[0][:70]
                pipe = 32.3 * g("mfma_bf16") + 64.0 * g("mfma_f32")
                other = 4.9 * (g("valu") + g("lds")) + 9.0 * g("valu_t")
                serial = pipe + other
                inter = max(pipe, 12.5 * (g("mfma_bf16") + g("mfma_f32")) + other)
                rows.append((nm, g("mfma_bf16"), g("mfma_f32"), g("valu"), g("valu_t"), g("lds"), f"{g('vmem_ld')} / {g('vmem_st')}",
                             g("wait"), inner, serial, inter, pipe / max(serial, 1.0)))
            for r_ in sorted(rows, key=lambda x: (-x[9])):
                print("| `%s` | %d | %d | %d | %d | %d | %s | %d | %d | %.0f | %.0f | %.0f %% |" % (
                    r_[0], r_[1], r_[2], r_[3], r_[4], r_[5], r_[6], r_[7], r_[8], r_[9], r_[10], 100 * r_[11]))
            print()


if __name__ == "__main__":
    main()
