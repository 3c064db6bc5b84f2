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


def gaps(path, last_ms=None):
    """GPU idle analysis of the kernel timeline: busy time vs span, idle time attributed to the kernel that FOLLOWS
    each gap (= the launch the host was late for)."""
    if path.endswith(".db"):
        db = sqlite3.connect(path)
        ks = sorted(db.cursor().execute("select start, end, name from kernels").fetchall())
    else:
        ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
    if last_ms is not None:  # only the tail of the run (the timed region)
        t1 = ks[-1][1]
        ks = [k for k in ks if k[0] >= t1 - last_ms * 1e6]
    span = ks[-1][1] - ks[0][0]
    busy, cur_end = 0, ks[0][0]
    idle_by = defaultdict(lambda: [0, 0])
    for st, en, name in ks:
        if st > cur_end:
            g = idle_by[name.replace("(anonymous namespace)::", "")[:60]]
            g[0] += 1
            g[1] += st - cur_end
        busy += max(0, en - max(st, cur_end))
        cur_end = max(cur_end, en)
    print(f"\n## timeline: span {span/1e6:.2f} ms, GPU busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), "
          f"idle {(span-busy)/1e6:.2f} ms over {len(ks)} launches\n")
    print("| idle before kernel | gaps | total idle ms | avg gap us |")
    print("|---|---|---|---|")
    for name, (n, t) in sorted(idle_by.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"| `{name}` | {n} | {t/1e6:.3f} | {t/n/1e3:.1f} |")


if __name__ == "__main__":
    main()
    if len(sys.argv) > 2 and sys.argv[2] == "--gaps":
        gaps(sys.argv[1], float(sys.argv[3]) if len(sys.argv) > 3 else None)
