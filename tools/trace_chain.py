#!/usr/bin/env python3
"""The busiest queue of one replayed step (between optimizer steps #i and #j of a rocprofv3 kernel_trace.csv): its kernels by
total time, and the idle time between consecutive kernels of that queue (waits on other queues / launch gaps).
usage: python tools/trace_chain.py kernel_trace.csv i j"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    q = r.get("Queue_Id", "?") + "/" + r.get("Stream_Id", "?")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], q))
rows.sort()
ad = [r for r in rows if "adam_multi" in r[2]]
i, j = int(sys.argv[2]), int(sys.argv[3])
a, b = ad[i][1], ad[j][1]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
span = (b - a) / 1e6 / (j - i)
busy = collections.defaultdict(float)
for s, e, n, q in sel:
    busy[q] += (e - s) / 1e6
print("step %.2f ms; queues: %s" % (span, ", ".join("%s %.1f ms" % (q, t / (j - i)) for q, t in sorted(busy.items(), key=lambda kv: -kv[1]))))
for main, _ in sorted(busy.items(), key=lambda kv: -kv[1])[:2]:
    ch = [r for r in sel if r[3] == main]
    gaps, tot_gap = [], 0.0
    for k in range(1, len(ch)):
        g = ch[k][0] - ch[k - 1][1]
        if g > 0:
            tot_gap += g
            gaps.append((g, ch[k - 1][2][:60], ch[k][2][:60]))
    print("queue %s: %d kernels, busy %.2f ms, idle between kernels %.2f ms per step" % (main, len(ch) // (j - i), busy[main] / (j - i), tot_gap / 1e6 / (j - i)))
    hist = collections.Counter()
    for g, _, _ in gaps:
        hist[min(int(g / 1e3) // 2 * 2, 40)] += 1
    print("  gap histogram (us bucket: count): " + "  ".join("%d: %d" % kv for kv in sorted(hist.items())))
    by_next = collections.defaultdict(float)
    for g, p, n in gaps:
        by_next[n] += g / 1e6
    print("  idle time by the kernel that follows the gap (ms per step):")
    for n, t in sorted(by_next.items(), key=lambda kv: -kv[1])[:14]:
        print("    %6.2f  %s" % (t / (j - i), n))
    c = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, q in ch:
        c[n[:90]][0] += 1
        c[n[:90]][1] += (e - s) / 1e6
    print("  kernels of this queue (ms per step):")
    for n, (cnt, t) in sorted(c.items(), key=lambda kv: -kv[1][1])[:24]:
        print("    %6.2f  x%-4d %s" % (t / (j - i), cnt // (j - i), n))
