#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd .db or kernel-trace .csv) into a per-kernel table (markdown).

    python tools/prof_summary.py gpurun_out/prof/r1_results.db > profiles/r01_bench_kernel_stats.md
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    for name, start, end, vgpr, lds, gx, wx in cur.execute(
            "select name, start, end, vgpr_count, lds_size, grid_x, workgroup_x from kernels"):
        yield name, end - start, vgpr, lds, gx, wx


def rows_from_csv(path):
    for r in csv.DictReader(open(path)):
        yield (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("VGPR_Count", ""),
               r.get("LDS_Block_Size", ""), r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(lambda: [0, 0, 10**18, 0, "", ""])
    for name, dur, vgpr, lds, gx, wx in rows:
        a = agg[name]
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
        a[4], a[5] = vgpr, lds
    tot = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of `{path}`\n")
    print(f"total kernel time {tot/1e6:.3f} ms over {sum(a[0] for a in agg.values())} launches\n")
    print("| kernel | calls | total ms | % | avg us | min us | max us | VGPR | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name.replace("(anonymous namespace)::", "")
        short = short[:100]
        print(f"| `{short}` | {a[0]} | {a[1]/1e6:.3f} | {100*a[1]/tot:.1f} | {a[1]/a[0]/1e3:.1f} | {a[2]/1e3:.1f} | "
              f"{a[3]/1e3:.1f} | {a[4]} | {a[5]} |")


if __name__ == "__main__":
    main()
