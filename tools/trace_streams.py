#!/usr/bin/env python3
"""Per-queue busy time and kernel concurrency inside a window of a rocprofv3 kernel_trace.csv.
usage: python tools/trace_streams.py kernel_trace.csv [lo_frac hi_frac | adam i j]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    q = r.get("Queue_Id", "?") + "/" + r.get("Stream_Id", "?")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], q))
rows.sort()
if len(sys.argv) > 2 and sys.argv[2] == "adam":   # window = from the end of optimizer step #i to the end of step #j
    ad = [r for r in rows if "adam_multi" in r[2]]
    i, j = int(sys.argv[3]), int(sys.argv[4])
    a, b = ad[i][1], ad[j][1]
    print("steps %d..%d: %.2f ms per step" % (i, j, (b - a) / 1e6 / (j - i)))
else:
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
    T0, T1 = rows[0][0], rows[-1][1]
    a, b = T0 + (T1 - T0) * lo, T0 + (T1 - T0) * hi
sel = [r for r in rows if r[0] >= a and r[1] <= b]
span = (b - a) / 1e6
print("window %.1f ms, %d kernels" % (span, len(sel)))
perq = collections.defaultdict(float); nq = collections.Counter()
for s, e, n, q in sel:
    perq[q] += (e - s) / 1e6; nq[q] += 1
for q, t in sorted(perq.items(), key=lambda kv: -kv[1]):
    print("  queue %-12s busy %7.1f ms (%5.1f%%)  %6d kernels" % (q, t, 100 * t / span, nq[q]))
ev = []
for s, e, n, q in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist = collections.defaultdict(float); cur = 0; last = a
for t, d in ev:
    hist[cur] += t - last; last = t; cur += d
hist[cur] += b - last
print("concurrency: " + "  ".join("%d kernels %.1f%%" % (k, 100 * v / (b - a)) for k, v in sorted(hist.items())))
# top kernels by time in window per queue
for q in perq:
    c = collections.defaultdict(float)
    for s, e, n, qq in sel:
        if qq == q: c[n[:70]] += (e - s) / 1e6
    print("  queue", q, "top:", [(k, round(v, 1)) for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:6]])
