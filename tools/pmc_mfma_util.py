#!/usr/bin/env python3
"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.
SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs (checked: equals 32 cycles x the number of
v_mfma_f32_32x32x16_bf16 instructions a launch executes); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
util = mfma_busy / (gui_active/8 * 1024).  usage: python tools/pmc_mfma_util.py counter_collection.csv"""
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); t = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:78], r["Grid_Size"])
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[k] += 1
        t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("%-80s %9s %5s %9s %9s %7s" % ("kernel", "grid", "calls", "ms/call", "clk GHz", "MfmaUtil"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:16]:
    c = max(n[k], 1)
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / c
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / c / 8.0
    ms = t[k] / c / 1e6
    print("%-80s %9s %5d %9.3f %9.2f %6.1f%%" % (k[0], k[1], c, ms, cyc / (ms * 1e6) if ms else 0, 100 * mf / (cyc * 1024) if cyc else 0))
