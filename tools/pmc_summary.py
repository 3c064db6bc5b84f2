#!/usr/bin/env python3
"""Summarise `rocprofv3 --pmc ... --kernel-trace --output-format csv` counter_collection CSVs per kernel (markdown).

    python tools/pmc_summary.py gpurun_out/pmc/*_counter_collection.csv [name-filter ...]
Derived columns (when the counters are present): clock = GRBM_GUI_ACTIVE/8 XCDs / duration; MFMA pipe busy =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); wave-time split: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (wave has
an instruction ready but cannot issue it: pipe busy or arbitration lost), SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (issuing), the
rest (waiting on s_waitcnt / barriers / nothing to issue); VALU+MFMA instructions per launch.
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    paths = [p for a in sys.argv[1:] if a.endswith(".csv") for p in glob.glob(a)]
    filters = [a for a in sys.argv[1:] if not a.endswith(".csv")]
    per = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
    dur = defaultdict(dict)                         # kernel -> dispatch -> ns
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            if filters and not any(f in k for f in filters):
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    names = sorted({c for k in per for c in per[k]})
    print("counters:", " ".join(names), "\n")
    print("| kernel | n | avg us | clock GHz | MFMA pipe busy | waves: blocked at issue | waves: waiting (waitcnt etc.) | waves: issuing | VALU+MFMA insts per launch |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k in sorted(per, key=lambda k: -sum(dur[k].values())):
        c = {n: sum(v) / len(v) for n, v in per[k].items()}
        n = len(dur[k])
        us = sum(dur[k].values()) / n / 1e3
        g = c.get("GRBM_GUI_ACTIVE")
        f = lambda x: "-" if x is None else f"{100*x:.1f} %"  # noqa: E731
        clock = g / 8 / (us * 1e3) if g else None
        wc = c.get("SQ_WAVE_CYCLES")
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        busy = mf / (g / 8 * 1024) if (mf is not None and g) else None
        wait = c.get("SQ_WAIT_INST_ANY") / wc if (wc and "SQ_WAIT_INST_ANY" in c) else None
        act = c.get("SQ_ACTIVE_INST_ANY") / wc if (wc and "SQ_ACTIVE_INST_ANY" in c) else None
        stall = None if (wait is None or act is None) else 1 - wait - act
        occ = wc / c["SQ_BUSY_CYCLES"] / 4 if (wc and c.get("SQ_BUSY_CYCLES")) else None
        valu = c.get("SQ_INSTS_VALU") / wc if (wc and "SQ_INSTS_VALU" in c) else None
        iv = c.get("SQ_INSTS_VALU")
        print(f"| `{k[:48]}` | {n} | {us:.1f} | {'-' if clock is None else f'{clock:.2f}'} | {f(busy)} | {f(wait)} | "
              f"{f(stall)} | {f(act)} | {'-' if iv is None else f'{iv:.4g}'} |")


if __name__ == "__main__":
    main()
