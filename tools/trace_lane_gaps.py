#!/usr/bin/env python3
"""Per queue of a rocprofv3 kernel_trace.csv, between the ends of optimizer steps i and j: kernels, busy time, and the time
lost between consecutive kernels of the queue in gaps below `cut` us (back-to-back dependent launches; longer gaps are waits
for another queue).  usage: python tools/trace_lane_gaps.py kernel_trace.csv i j [cut_us]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")))
rows.sort()
ad = [r for r in rows if "adam_multi" in r[2]]
i, j = int(sys.argv[2]), int(sys.argv[3])
cut = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 30e3
a, b = ad[i][1], ad[j][1]
steps = j - i
perq = collections.defaultdict(list)
for s, e, n, q in rows:
    if s >= a and e <= b:
        perq[q].append((s, e, n))
print("window %.2f ms = %d steps of %.2f ms" % ((b - a) / 1e6, steps, (b - a) / 1e6 / steps))
for q, ks in sorted(perq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, n in ks)
    gaps = [ks[k + 1][0] - ks[k][1] for k in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < cut]
    hist = collections.Counter(min(int(g // 1000), 20) for g in small)
    print("queue %s: %5d kernels/step, busy %6.2f ms/step, %4d gaps/step below %.0f us = %5.2f ms/step (median %.1f us); overlapped starts %d/step" % (
        q, len(ks) // steps, busy / 1e6 / steps, len(small) // steps, cut / 1e3, sum(small) / 1e6 / steps,
        sorted(small)[len(small) // 2] / 1e3 if small else 0.0, sum(1 for g in gaps if g < 0) // steps))
    print("    gap histogram (us: count/step): " + " ".join("%d:%d" % (k, v // steps) for k, v in sorted(hist.items())))
