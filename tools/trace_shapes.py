#!/usr/bin/env python3
"""Kernels of one replayed step grouped by (name, grid, LDS bytes): launches per step, average and total time.
usage: python tools/trace_shapes.py kernel_trace.csv i j [min_total_ms]   (window: optimizer steps #i..#j)"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", "?"))))
rows.sort()
ad = [r for r in rows if "adam_multi" in r[2]]
i, j = int(sys.argv[2]), int(sys.argv[3])
floor = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
a, b = ad[i][1], ad[j][1]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
c = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, g in sel:
    k = (n[:64], g)
    c[k][0] += 1
    c[k][1] += (e - s) / 1e3
n = j - i
print("step %.2f ms, %d kernels per step, sum of durations %.2f ms" % ((b - a) / 1e6 / n, len(sel) // n, sum(v[1] for v in c.values()) / 1e3 / n))
for (name, g), (cnt, us) in sorted(c.items(), key=lambda kv: -kv[1][1]):
    if us / 1e3 / n < floor:
        continue
    wg = int(g[3]) if g[3].isdigit() else 1
    print("%7.3f ms  x%-5.1f avg %7.1f us  grid %6d x%s x%s wg %4s lds %6s  %s" % (us / 1e3 / n, cnt / n, us / cnt, int(g[0]) // max(wg, 1), g[1], g[2], g[3], g[4], name))
