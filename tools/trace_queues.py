#!/usr/bin/env python3
"""One update of a raw `rocprofv3 --kernel-trace` CSV (between the last two k_gae launches), queue by queue: busy time, the
gaps of the main queue (what the critical path waits for) and its kernels by total time.
    python tools/trace_queues.py gpurun_out/r06h/smac_raw_kernel_trace.csv [min_gap_us]"""
import collections
import csv
import sys

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:34], r.get("Queue_Id")) for r in rows)
gae = [i for i, e in enumerate(ev) if "k_gae" in e[2]]
a, b = gae[-2], gae[-1]
upd = ev[a:b]
t0 = upd[0][0]
print(f"update: {(upd[-1][1] - t0) / 1e6:.2f} ms, {len(upd)} launches")
cur, idle = upd[0][1], 0.0
for s, e, n, q in upd[1:]:
    idle += max(0, s - cur)
    cur = max(cur, e)
print(f"no kernel on any queue: {idle / 1e6:.2f} ms")
qs = collections.Counter(e[3] for e in upd)
main = qs.most_common(1)[0][0]
for q, n in sorted(qs.items()):
    es = [e for e in upd if e[3] == q]
    print(f"queue {q}{' (main)' if q == main else ''}: {n} launches, busy {sum(e[1] - e[0] for e in es) / 1e6:.2f} ms, "
          f"{(es[0][0] - t0) / 1e6:.2f} .. {(es[-1][1] - t0) / 1e6:.2f} ms")
es = [e for e in upd if e[3] == main]
g = [((c[0] - p[1]) / 1e3, (p[1] - t0) / 1e6, p[2], c[2]) for p, c in zip(es[:-1], es[1:]) if (c[0] - p[1]) / 1e3 > min_gap]
print(f"main-queue gaps > {min_gap:.0f} us: {len(g)}, {sum(x[0] for x in g) / 1e3:.2f} ms; all smaller ones: "
      f"{sum(max(0, c[0] - p[1]) for p, c in zip(es[:-1], es[1:]) if (c[0] - p[1]) / 1e3 <= min_gap) / 1e6:.2f} ms")
for x in sorted(g, reverse=True)[:14]:
    print("  %5.0f us at %6.2f ms after %s before %s" % x)
c = collections.defaultdict(lambda: [0, 0.0])
for e in es:
    c[e[2]][0] += 1
    c[e[2]][1] += (e[1] - e[0]) / 1e3
for k, v in sorted(c.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-36s n %3d total %6.2f ms avg %6.1f us" % (k, v[0], v[1] / 1e3, v[1] / v[0]))
