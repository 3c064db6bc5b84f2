#!/usr/bin/env python3
"""HBM traffic per launch of the update's streaming kernels, measured on bench.py ITSELF (same commit, same workload).

    tools/pmc_traffic.sh            # on the GPU box: for every --config one plain run (per-tag algorithmic bytes) and two
                                    # PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only), CSVs under gpurun_out/pmc_traffic/
    python tools/pmc_traffic.py report gpurun_out/pmc_traffic profiles/r03_hbm_traffic   # -> .json + .md

FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md / HBM);
WRITE_SIZE is taken as reported (calibrated 1:1 on fwd_hidden, which writes exactly x_hat + mask + rstd); counter unit
KB = 1024 B.  Kernels are matched to bench.py's tags by name; tags sharing one kernel template are reported as a group.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

# kernel-name prefix (after stripping the anonymous namespace) -> tags of bench.py's `kernels` that launch it
GROUPS = [
    (r"void k_fwd_fused2x<", ("fwd_fused2", "fwd_fused2_k64")),
    (r"void k_fwd_fused2<", ("fwd_fused2",)),
    (r"void k_fwd_hidden<\d+, \d+, 0>", ("fwd_hidden",)),
    (r"void k_fwd_hidden<\d+, \d+, [12]>", ("tangent_hidden",)),
    (r"void k_panel<false>", ("fwd_panel",)),
    (r"void k_panel<true>", ("bwd_panel",)),
    (r"void k_bwd_dx_dw<0", ("bwd_full",)),
    (r"void k_bwd_dx_dw<1", ("bwd_full_dw1",)),
    (r"void k_bwd_dx<\d+, \d+, 0>", ("bwd_dx",)),
    (r"void k_bwd_dx<\d+, \d+, [1-9]>", ("bwd_dx_dw1",)),
    (r"void k_dw(_tr<|_tr_multi<|<0)", ("dw_hidden", "dw_gru", "dw_input")),
    (r"void k_dw_tr_multi_v<", ("dw_trunk",)),
    (r"(void )?k_dw_gru6", ("dw_gru",)),
    (r"void k_fwd_trunk<", ("fwd_trunk",)),
    (r"void k_bwd_trunk<", ("bwd_trunk",)),
    (r"void k_dw<1", ("dw_head",)),
    (r"void k_fwd_wide<", ("fwd_wide", "tangent_wide", "tangent_hidden")),  # (the one-launch hidden tangent is a k_fwd_wide)
    (r"(void )?k_x0n_", ("x0n_wide",)),
    (r"void k_actor_head<.*(true|false), true, (true|false)>", ("actor_head_loss",)),
    (r"void k_actor_head<.*(true|false), false, (true|false)>", ("actor_head_logp",)),
    (r"void k_critic_head<\d+, true", ("critic_head_loss",)),
    (r"(void )?k_gru_fwd", ("gru_fwd",)),
    (r"(void )?k_gru_(bwd|dx)", ("gru_bwd",)),
    (r"(void )?k_gae", ("gae_returns",)),
    (r"void k_upd_fwd<", ("update_fwd", "update_logp", "update_fwd_critic", "update_values")),
    (r"void k_upd_d", ("update_bwd",)),
]


def load_counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def report(root, out_base):
    sha = open(os.path.join(root, "git_sha.txt")).read().strip() if os.path.exists(os.path.join(root, "git_sha.txt")) else None
    out = dict(git_sha=sha, source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
               "`python bench.py --config <c> --steps 1 --warmup 0 --instr-steps 0 --no-kernel-timing --cpu-cols 0`; FETCH_SIZE x2 "
               "(gfx950 wide-read correction), WRITE_SIZE as reported; KB = 1024 B; algorithmic bytes per launch from a plain run of the "
               "same command with kernel timing (harl_amd/traffic.py)", workloads={})
    md = [f"# HBM traffic per launch (PMC) of bench.py's own kernels, commit {sha}\n",
          "Two separate `rocprofv3 --pmc <counter> --kernel-trace` passes per workload over one `bench.py` step (recipe: `tools/pmc_traffic.sh`).",
          "FETCH_SIZE doubled (gfx950 counts the 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md / HBM); WRITE_SIZE as",
          "reported.  `algorithmic` = bytes every launch must move once (harl_amd/traffic.py, evaluated on each launch's own arguments),",
          "averaged over the launches of the tags that share the kernel.\n"]
    for cfg in sorted(os.listdir(root)):
        plain = os.path.join(root, cfg, "plain.json")
        if not os.path.exists(plain):
            continue
        line = [ln for ln in open(plain) if ln.startswith("{")][-1]
        kern = json.loads(line)["kernels"]
        fe = load_counters(os.path.join(root, cfg, "fetch"))
        wr = load_counters(os.path.join(root, cfg, "write"))
        res = {}
        md.append(f"\n## {cfg}: {json.loads(line)['config']['workload']}\n")
        md.append("| kernel (tags) | kernel launches / calls | fetch MB/call | write MB/call | algorithmic MB/call | traffic / algorithmic |\n|---|---|---|---|---|---|")
        for pat, tags in GROUPS:
            names = [k for k in fe if re.match(pat, k)]
            if not names:
                continue
            n = sum(len(fe[k]["FETCH_SIZE"]) for k in names)
            fetch = 2.0 * sum(sum(fe[k]["FETCH_SIZE"]) for k in names)
            write = sum(sum(wr[k]["WRITE_SIZE"]) for k in names if k in wr)
            tg = [t for t in tags if t in kern and kern[t].get("alg_bytes")]
            if not tg or n == 0:
                continue
            # both runs execute exactly ONE update step, so totals are comparable even where one call of a tag issues several
            # kernels (the first-layer weight gradient of a 416-wide input is k_dw_tr launches)
            calls = sum(kern[t]["n"] for t in tg)
            alg_total = sum(kern[t]["alg_bytes"] for t in tg)
            ratio = (fetch + write) / alg_total
            for t in tg:
                res[t] = dict(group=list(tg), kernels=sorted(set(k.split("(")[0] for k in names)), kernel_launches=n, calls=calls,
                              fetch_bytes_per_call=fetch / calls, write_bytes_per_call=write / calls,
                              algorithmic_bytes_per_call=alg_total / calls, ratio=ratio)
            md.append(f"| {names[0].split('(')[0][:60]}{' +%d more' % (len(names) - 1) if len(names) > 1 else ''} ({', '.join(tg)}) | {n} / {calls} | "
                      f"{fetch / calls / 1e6:.1f} | {write / calls / 1e6:.1f} | {alg_total / calls / 1e6:.1f} | {ratio:.3f} |")
        out["workloads"][cfg] = res
    json.dump(out, open(out_base + ".json", "w"), indent=1)
    open(out_base + ".md", "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    report(sys.argv[2], sys.argv[3])
