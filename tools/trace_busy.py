#!/usr/bin/env python3
"""GPU busy fraction from a rocprofv3 kernel_trace.csv: union of kernel intervals vs span (last N%% of the trace)."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0 = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1 - frac)
sel = [r for r in rows if r[0] >= t0]
span = sel[-1][1] - sel[0][0]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
for s, e, _ in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in sel)
print("window %.1f ms: union-busy %.1f ms (%.1f%%), sum of kernel durations %.1f ms, %d kernels" % (span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, len(sel)))
gaps = []
prev_e = sel[0][1]
for s, e, n in sel[1:]:
    if s > prev_e: gaps.append((s - prev_e, n))
    prev_e = max(prev_e, e)
gaps.sort(reverse=True)
print("largest idle gaps (us) before kernel:")
for g, n in gaps[:12]: print("  %8.1f  %s" % (g / 1e3, n[:100]))
