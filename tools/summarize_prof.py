#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into a short text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = name.replace("void ct::", "").replace("ct::", "")
    return name[:70]


# --- kernel stats -------------------------------------------------------------------------------
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.DictReader(fh)):
            if i >= 12:
                break
            print("  %-70s calls=%6s avg_ns=%12s total_ns=%14s pct=%s" % (
                short(row.get("Name", "")), row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"),
                row.get("Percentage")))

# --- kernel trace: VGPR etc. ---------------------------------------------------------------------
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    seen = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if k not in seen:
                seen[k] = row
    print("== kernel resources")
    for k, row in seen.items():
        print("  %-70s vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s grid=%s wg=%s" % (
            short(k), row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"),
            row.get("LDS_Block_Size"), row.get("Scratch_Size"), row.get("Grid_Size_X", row.get("Grid_Size")),
            row.get("Workgroup_Size_X", row.get("Workgroup_Size"))))

# --- PMC passes ---------------------------------------------------------------------------------
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row.get("Kernel_Name", ""))
            c = row.get("Counter_Name", "")
            try:
                v = float(row.get("Counter_Value", "nan"))
            except ValueError:
                continue
            a = agg[k][c]
            a[0] += v
            a[1] += 1
print("== PMC (mean per dispatch)")
for k in agg:
    print("  " + k)
    for c in sorted(agg[k]):
        s, n = agg[k][c]
        print("      %-28s %18.1f  (n=%d)" % (c, s / max(n, 1), n))
