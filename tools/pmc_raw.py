#!/usr/bin/env python3
"""Raw per-kernel averages of every counter in rocprofv3 counter_collection CSVs (one or more PMC passes merged).

    python tools/pmc_raw.py /tmp/pmcA/*/*_counter_collection.csv /tmp/pmcB/*/*_counter_collection.csv [name-filter ...]
Counters are summed over the chip; per-wave figures divide by SQ_WAVES when present."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    paths = [p for a in sys.argv[1:] if a.endswith(".csv") for p in glob.glob(a)]
    filters = [a for a in sys.argv[1:] if not a.endswith(".csv")]
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for p in paths:
        seen = set()
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            if filters and not any(f in k for f in filters):
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (p, r["Dispatch_Id"]) not in seen:
                seen.add((p, r["Dispatch_Id"]))
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(per, key=lambda k: -sum(dur[k])):
        us = sum(dur[k]) / len(dur[k]) / 1e3
        print(f"## {k[:110]}\n   n={len(dur[k])} avg_us={us:.1f}")
        c = {n: sum(v) / len(v) for n, v in per[k].items()}
        wc = c.get("SQ_WAVE_CYCLES")
        for n in sorted(c):
            extra = f"  ({100 * c[n] / wc:.1f} % of SQ_WAVE_CYCLES)" if wc and n != "SQ_WAVE_CYCLES" and n.startswith("SQ_") else ""
            print(f"   {n:32s} {c[n]:.4g}{extra}")


if __name__ == "__main__":
    main()
