#!/usr/bin/env python3
"""Per-kernel SQ counter summary from a rocprofv3 --pmc pass (any SQ_* set).  usage: python tools/pmc_sq.py counter_collection.csv [filter]"""
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    if flt and flt not in r["Kernel_Name"]:
        continue
    d[k][r["Counter_Name"]] += float(r["Counter_Value"]); names.add(r["Counter_Name"])
    if r["Counter_Name"] == sorted(names)[0]:
        n[k] += 1
names = sorted(names)
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    base = v.get("SQ_WAVE_CYCLES", 1.0)
    print(k)
    for c in names:
        print("    %-28s %14.0f  %6.1f%% of SQ_WAVE_CYCLES" % (c, v[c], 100 * v[c] / base))
