"""Build libharl_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libharl_hip.so")
SOURCES = ["elementwise.hip", "mlp.hip", "wide.hip", "trunk.hip", "panel.hip", "heads.hip", "multihead.hip", "update.hip", "gru.hip", "gru_cell.hip", "host_rng.hip", "comm.hip"]
HEADERS = ["common.h", "split_mfma.h", "mfma_transpose.h", "heads_common.h", "dw_common.h", "fwd_epilogue.h", os.path.join("..", "..", "include", "harl_hip.h")]


# MFMA results in VGPRs instead of AGPRs: the epilogues (LayerNorm / ReLU, operand splits, transposes) consume every
# accumulator element on the VALU, and each element held in an AGPR costs a v_accvgpr_read first -- 5.7 % of the issue slots of
# k_bwd_dx<128,128,1>, 17 % of <64,64,2> (tools/isa_census.py).  Per file: the same flag crashes the compiler on heads.hip and
# changes nothing in wide.hip / update.hip.  Adopted in round 3 after the full GPU parity suite passed on the variant
# (gpurun_out/r3_vgpr_suite.log: 107 passed; A/B on one box 21.53 / 21.69 -> 20.93 / 21.18 ms per MPE update).
# HARL_HIPCC_EXTRA="mlp.hip:;gru.hip:..." overrides per file for A/B builds.
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# No packed fp32 VALU (v_pk_add / v_pk_mul / v_pk_fma_f32) in the kernels that interleave their epilogues with MFMAs (round 6,
# session 3): the LayerNorm / ReLU epilogues and the operand splits were written on register pairs (f32x2, common.h) to halve
# their instruction count, but beside matrix instructions a packed fp32 op costs MORE than the two scalar ops it replaces
# (MI355X_MICROARCH.md, "price of one filler beside MFMAs": 2 v_pk_add_f32 per gap +26 cycles against 2 v_fma_f32).  -DHARL_NO_PK
# turns f32x2 into a pair of scalars (same arithmetic, same bits), -fno-slp-vectorize keeps hipcc from re-packing them.  A/B on
# one box, three interleaved runs each (profiles/r06s3_packed_valu_ab.md): MPE 16.77 -> 16.36-16.51 ms, HalfCheetah-6x1
# 50.27 -> 49.43 ms, SMAC 3s5z unchanged (20.5 ms; the GRU chain kernels of gru.hip, which have no MFMA shadow to lose, keep
# their packed epilogues: unpacked they measured 0.1 - 0.2 ms slower).  One setting for every file whose kernels must agree bit
# for bit with each other (trunk.hip <-> mlp.hip / wide.hip layer launches: tests/gpu_checks.check_trunk_fused).
NO_PK = ["-DHARL_NO_PK", "-fno-slp-vectorize"]
DEFAULT_EXTRA: dict = {"mlp.hip": VGPR_FORM + NO_PK, "gru.hip": VGPR_FORM, "panel.hip": VGPR_FORM + NO_PK, "trunk.hip": VGPR_FORM + NO_PK,
                       "update.hip": NO_PK, "heads.hip": NO_PK, "wide.hip": NO_PK, "multihead.hip": NO_PK,
                       # ... and the element-wise kernels (the composed GRU cell of gru_cell.hip, the activation / LayerNorm launches and
                       # the optimiser kernels of elementwise.hip): hatrpo_gru128 256 -> 230 ms, SMAC 3s5z 20.02 -> 19.83 ms
                       # (three interleaved runs each, profiles/r06s3_packed_valu_ab.md); gru.hip alone measured neutral and
                       # keeps its packed epilogues
                       "gru_cell.hip": NO_PK, "elementwise.hip": NO_PK}


def _extra_flags() -> dict:
    """Per-file extra hipcc flags; HARL_HIPCC_EXTRA="mlp.hip:-mllvm -amdgpu-mfma-vgpr-form;gru.hip:" overrides the defaults
    above for A/B builds (an empty flag list switches a file back; rebuild with ``python -m harl_amd._build``)."""
    out = {k: list(v) for k, v in DEFAULT_EXTRA.items()}
    for item in filter(None, os.environ.get("HARL_HIPCC_EXTRA", "").split(";")):
        name, _, flags = item.partition(":")
        out[name.strip()] = flags.split()
    return out


EXTRA_FLAGS = _extra_flags()


def _flag_stamp(extra: dict) -> str:
    """The effective per-file flag set as text; stored next to the library so that a build made with other flags counts as
    stale (A/B measurements must not be attributed to the wrong variant)."""
    return ";".join("%s:%s" % (k, " ".join(extra[k])) for k in sorted(extra) if extra[k])


def _stale(lib: str, extra: dict) -> bool:
    if not os.path.exists(lib):
        return True
    try:
        with open(lib + ".flags") as f:
            if f.read() != _flag_stamp(extra):
                return True
    except OSError:
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, variant: str = "", extra: dict = None) -> str:
    """Compile every HIP source into harl_amd/lib/libharl_hip.so for gfx950 (``variant``: libharl_<variant>.so with its own
    object directory, for A/B builds selected at run time with HARL_LIB)."""
    extra = EXTRA_FLAGS if extra is None else extra
    lib = LIB if not variant else os.path.join(LIBDIR, "libharl_%s.so" % variant)
    if not force and not _stale(lib, extra):
        return lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = LIBDIR if not variant else os.path.join(LIBDIR, "obj_" + variant)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in HEADERS)
    try:
        with open(lib + ".flags") as f:
            same_flags = f.read() == _flag_stamp(extra)
    except OSError:
        same_flags = False
    for src in SOURCES:  # one hipcc per translation unit, in parallel
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        # an object newer than its source, every header and made with the same flags is reused (a one-file edit recompiles one file)
        if (not force and same_flags and os.path.exists(obj)
                and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src)))):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj]
        cmd += extra.get(src, [])
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    subprocess.check_call(cmd)
    with open(lib + ".flags", "w") as f:
        f.write(_flag_stamp(extra))
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
