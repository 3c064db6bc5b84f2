"""Static instruction census of the gfx950 kernels (no GPU needed).

    python tools/isa_census.py [mlp heads gru ...]  > profiles/r02_isa_census.md

Compiles harl_amd/csrc/<name>.hip to gfx950 assembly with the library's own flags (`hipcc -S --cuda-device-only`) and, per
kernel, finds its STEADY-STATE LOOP -- the outermost backward branch that contains matrix instructions (the per-slab loop of
the persistent kernels; for kernels without MFMAs: the largest loop) -- and counts what one trip of it issues: MFMAs, VALU,
LDS, global memory, scalar, waits.  With the cost model measured in tools/mfma_valu_overlap.hip (a 32x32x16 bf16 MFMA holds
the SIMD for 8 issue slots of 4 cycles, a 32x32x2 f32 MFMA for 16, and nothing else issues in their shadow; a VALU / LDS
instruction takes one slot) that gives the ISSUE FLOOR of one slab per wave:

    slots = 8 * n_mfma_bf16 + 16 * n_mfma_f32 + n_valu + n_lds (+ transcendental VALU ops counted 4x)

which DESIGN.md section 3 compares with the measured time per slab.  Inner loops (weight staging, k-panels) are counted once
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
    print("# Static instruction census of the gfx950 kernels (`tools/isa_census.py`, hipcc -O3, no GPU)\n")
    print("One trip of each kernel's steady-state loop (one 32-sample slab per wave unless noted).  `slots` = issue slots of 4 "
          "cycles: 8 per bf16 MFMA (32x32x16), 16 per fp32 MFMA, 1 per VALU / LDS instruction, 4 per transcendental; "
          "`mfma share` = the part of them that is matrix work.  `inner` = loops nested inside (counted once: lower bound).\n")
    with tempfile.TemporaryDirectory() as td:
        for n in names:
            src, asm = os.path.join(CSRC, n + ".hip"), os.path.join(td, n + ".s")
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                                src, "-o", asm], capture_output=True, text=True)
            if r.returncode != 0:
                print(f"## {n}.hip: hipcc failed\n```\n{r.stderr[-400:]}\n```")
                continue
            funcs = parse_functions(asm)
            syms = sorted(funcs)
            pretty = dict(zip(syms, demangle(syms)))
            print(f"## {n}.hip\n")
            print("| kernel | mfma bf16 | mfma f32 | valu | transc. | lds | vmem ld / st | waits | inner | slots | mfma share | us @2.4 GHz |")
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
                share = (8 * g("mfma_bf16") + 16 * g("mfma_f32")) / sl
                rows.append((nm, g("mfma_bf16"), g("mfma_f32"), g("valu"), g("valu_t"), g("lds"), f"{g('vmem_ld')} / {g('vmem_st')}",
                             g("wait"), inner, sl, share, sl * 4 / 2400.0))
            for r_ in sorted(rows, key=lambda x: (-x[9])):
                print("| `%s` | %d | %d | %d | %d | %d | %s | %d | %d | %d | %.0f %% | %.2f |" % (
                    r_[0], r_[1], r_[2], r_[3], r_[4], r_[5], r_[6], r_[7], r_[8], r_[9], 100 * r_[10], r_[11]))
            print()


if __name__ == "__main__":
    main()
