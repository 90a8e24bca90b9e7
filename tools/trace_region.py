#!/usr/bin/env python3
"""Kernels of the main queue between two time marks (ms from the end of optimizer step i) of a rocprofv3 kernel_trace.csv.
usage: python tools/trace_region.py kernel_trace.csv i t0_ms t1_ms"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id", "?")))
rows.sort()
ad = [r for r in rows if "adam_multi" in r[2]]
a = ad[int(sys.argv[2])][1]
t0, t1 = float(sys.argv[3]) * 1e6 + a, float(sys.argv[4]) * 1e6 + a
prev = None
for s, e, n, q in rows:
    if s < t0 or s > t1:
        continue
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%8.3f ms q%s dur %7.1f us gap %7.1f  %s" % ((s - a) / 1e6, q, (e - s) / 1e3, gap, n))
    prev = e
