#!/usr/bin/env python3
"""Per-kernel summary of ONE rocprofv3 --pmc run (counter_collection.csv + kernel_trace.csv of one process):
mean counter values per dispatch, mean duration, and MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x SIMDs)
(MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs; the busy counter over the 1024 SIMDs).

One run only: rocprofv3 names its files <pid>_counter_collection.csv; when a directory tree holds several runs (gpurun merges
every call's files into the same local gpurun_out/), the NEWEST counter file and the kernel trace of the same pid are used and
the others are listed under "_ignored_runs" -- round 3 averaged five builds this way (VERDICT W3).
usage: tools/summarize_pmc.py <dir> [substring ...]   -> JSON on stdout (with the source stamp of tools/stamp.py)"""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stamp import source_stamp
d = sys.argv[1]
want = sys.argv[2:]
cfiles = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
if not cfiles:
    sys.exit("no *counter_collection.csv under " + d)
cfile = cfiles[-1]
tfile = cfile.replace("counter_collection.csv", "kernel_trace.csv")
cnt = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for row in csv.DictReader(open(cfile)):
    a = cnt[row["Kernel_Name"]][row["Counter_Name"]]
    a[0] += float(row["Counter_Value"]); a[1] += 1
dur = defaultdict(lambda: [0.0, 0])
if os.path.exists(tfile):
    for row in csv.DictReader(open(tfile)):
        a = dur[row["Kernel_Name"]]
        a[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); a[1] += 1
out = {"source_stamp": source_stamp(), "_run": os.path.relpath(cfile, d), "_ignored_runs": [os.path.relpath(f, d) for f in cfiles[:-1]]}
tot_t = sum(v[0] for v in dur.values()) / 1e3
busy_t = 0.0
for k in cnt:
    c = {n: v[0] / v[1] for n, v in cnt[k].items()}
    if c.get("GRBM_GUI_ACTIVE") and k in dur:
        busy_t += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) * dur[k][0] / 1e3
out["_all_kernels"] = {"gpu_time_us": tot_t, "mfma_busy_frac_time_weighted": busy_t / tot_t if tot_t else None,
                       "note": "every kernel of the ONE run named in _run, torch's own included; MFMA-free kernels count as 0"}
for k in cnt:
    if want and not any(w in k for w in want):
        continue
    e = {c: v[0] / v[1] for c, v in cnt[k].items()}
    e["dispatches"] = max(v[1] for v in cnt[k].values())
    if k in dur:
        e["avg_duration_us"] = dur[k][0] / dur[k][1] / 1e3
        e["share_of_gpu_time"] = dur[k][0] / 1e3 / tot_t if tot_t else None
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        if "avg_duration_us" in e:
            e["clock_GHz_from_GRBM"] = e["GRBM_GUI_ACTIVE"] / 8.0 / (e["avg_duration_us"] * 1e3)
    out[k[:110]] = e
print(json.dumps(out, indent=1, sort_keys=True))
