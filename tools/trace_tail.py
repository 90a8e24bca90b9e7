#!/usr/bin/env python3
"""The last milliseconds of a traced step: kernels (queue, start relative to the optimizer launch, duration) that run in the
window before optimizer step #i of a rocprofv3 kernel_trace.csv - what the end of backward waits for.
usage: python tools/trace_tail.py kernel_trace.csv i [window_ms]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ad = [r for r in rows if "adam_multi" in r[2]]
i = int(sys.argv[2])
win = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
t1 = ad[i][0]
t0 = t1 - int(win * 1e6)
sel = [r for r in rows if r[1] > t0 and r[0] <= t1]
print("window %.1f ms before optimizer step %d (step period %.2f ms)" % (win, i, (ad[i][0] - ad[i - 1][0]) / 1e6))
for s, e, n, q in sel:
    print("  q%s  start %8.1f us  dur %7.1f us  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, n[:90]))
