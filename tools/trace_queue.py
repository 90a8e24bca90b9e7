#!/usr/bin/env python3
"""The kernels of ONE queue between two time marks (ms from the end of optimizer step i) of a rocprofv3 kernel_trace.csv, with the
gap to the queue's previous kernel.  usage: python tools/trace_queue.py kernel_trace.csv i queue t0_ms t1_ms"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:64], r.get("Queue_Id", "?")))
rows.sort()
ad = [r for r in rows if "adam_multi" in r[2]]
a = ad[int(sys.argv[2])][1]
q = sys.argv[3]
t0, t1 = float(sys.argv[4]) * 1e6 + a, float(sys.argv[5]) * 1e6 + a
prev = None
tot_gap = 0.0
for s, e, n, qq in rows:
    if qq != q:
        continue
    if s >= t0 and s <= t1:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        if prev is not None and 0 < gap < 50:
            tot_gap += gap
        print("%8.3f ms dur %7.1f us gap %6.1f  %s" % ((s - a) / 1e6, (e - s) / 1e3, gap, n))
    prev = e
print("gaps below 50 us in the window: %.1f us" % tot_gap)
