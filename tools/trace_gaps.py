#!/usr/bin/env python3
"""Idle gaps (no kernel running on any queue) inside steps i..j of a rocprofv3 kernel_trace.csv, and the main queue's own
gaps.  usage: python tools/trace_gaps.py kernel_trace.csv i j"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Queue_Id", "?")))
rows.sort()
ad = [r for r in rows if "adam_multi" in r[2]]
i, j = int(sys.argv[2]), int(sys.argv[3])
a, b = ad[i][1], ad[j][1]
sel = [r for r in rows if r[0] >= a and r[1] <= b]
print("window %.1f ms (%d steps)" % ((b - a) / 1e6, j - i))
# global idle gaps
cur_end = a; gaps = []; last = None
for s, e, n, q in sel:
    if s > cur_end:
        gaps.append((s - cur_end, last, (n, q), (cur_end - a) / 1e6))
    if e > cur_end:
        cur_end = e; last = (n, q)
tot = sum(g[0] for g in gaps)
print("global idle: %.2f ms total in %d gaps; > 20 us: %.2f ms" % (tot / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
for g in sorted(gaps, key=lambda g: -g[0])[:12]:
    print("  %7.1f us at %6.1f ms  after %s  before %s" % (g[0] / 1e3, g[3], g[1], g[2]))
# gap histogram of the busiest queue
perq = collections.Counter()
for s, e, n, q in sel: perq[q] += e - s
mq = perq.most_common(1)[0][0]
mrows = [r for r in sel if r[3] == mq]
gs = [mrows[k + 1][0] - mrows[k][1] for k in range(len(mrows) - 1)]
print("main queue %s: busy %.1f ms, %d kernels, gaps total %.1f ms; <5us %d, 5-20us %d, 20-100us %d, >100us %d (%.1f ms)" % (
    mq, perq[mq] / 1e6, len(mrows), sum(gs) / 1e6, sum(g < 5000 for g in gs), sum(5000 <= g < 20000 for g in gs),
    sum(20000 <= g < 100000 for g in gs), sum(g >= 100000 for g in gs), sum(g for g in gs if g >= 100000) / 1e6))
big = sorted(((mrows[k + 1][0] - mrows[k][1], mrows[k][2], mrows[k + 1][2], (mrows[k][1] - a) / 1e6) for k in range(len(mrows) - 1)), reverse=True)[:14]
for g in big:
    print("  %7.1f us at %6.1f ms  after %s  before %s" % (g[0] / 1e3, g[3], g[1], g[2]))
