#!/usr/bin/env python3
"""gpurun_out/prof_<round> (written on the GPU box by tools/gpu_profile_round.sh) -> the tracked summaries under profiles/, each
stamped with the source stamp of the build that was measured (tools/stamp.py; refused when it is not this tree's):
kernel-stats CSVs, bench JSON lines, HBM traffic + instruction counts of the Reinhard kernels (<round>_traffic.json), MFMA-busy
summaries of the CNN forwards."""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND)
DST = os.path.join(ROOT, "profiles")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from stamp import source_stamp
STAMP = open(os.path.join(SRC, "source_stamp.txt")).read().strip()
if STAMP != source_stamp():
    sys.exit("gpurun_out/prof_<round> was measured on sources with stamp %s, this tree has %s: re-run tools/gpu_profile_round.sh" % (STAMP, source_stamp()))
try:
    HEAD = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    DIRTY = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "color-transfer_amd/csrc", "include"], capture_output=True, text=True).stdout.strip())
except OSError:
    HEAD, DIRTY = None, None
STAMPS = {"source_stamp": STAMP, "git_head_at_collection": HEAD, "kernel_sources_uncommitted_at_collection": DIRTY}


def newest(sub, pattern):
    """gpurun merges a call's files INTO the local gpurun_out/: an earlier call's outputs may still lie beside the new ones"""
    f = glob.glob(os.path.join(SRC, sub, "**", pattern), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def stats_csv(sub, name):
    f = newest(sub, "*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, name))
        print("wrote", name)


def counters(sub):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    f = newest(sub, "*counter_collection.csv")
    if f:
        for row in csv.DictReader(open(f)):
            a = agg[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in agg.items()}


stats_csv("trace", ROUND + "_reinhard_bench_kernel_stats.csv")
stats_csv("trace_dc1080", ROUND + "_dcmcs3di_1080p_kernel_stats.csv")
stats_csv("trace_gm960", ROUND + "_gmflow_960x540_kernel_stats.csv")
stats_csv("trace_dmsct960", ROUND + "_dmsct_960x540_kernel_stats.csv")
stats_csv("trace_idt", ROUND + "_idt_kernel_stats.csv")
for a, b in (("bench_under_rocprofv3.json", ROUND + "_bench_under_rocprofv3.json"), ("bench_default.json", ROUND + "_bench_default.json"),
             ("bench_20_5.json", ROUND + "_bench_20_5.json")):
    p = os.path.join(SRC, a)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(DST, b)); print("wrote", b)

# ---- Reinhard: HBM bytes (MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 64 B per 128-B request) + instruction counts
bench = json.load(open(os.path.join(SRC, "bench_under_rocprofv3.json")))
pairs = bench["config"]["pairs_per_step_per_gpu"]
H, W = bench["config"]["height"], bench["config"]["width"]
per = defaultdict(dict)
for sub in sorted(os.listdir(SRC)):
    if sub.startswith("pmc_") and os.path.isdir(os.path.join(SRC, sub)):
        for k, d in counters(sub).items():
            if "lab_moments_lut_kernel" in k or "reinhard_apply_lut_kernel" in k or "lab_moments_kernel" in k or "reinhard_apply_kernel" in k:
                short = k.split("ct::")[-1].split("(")[0].replace("void ", "")
                per[short].update(d)
alg = 2 * 3 * 4 * H * W * pairs            # two float32 frames per pair, read or written once by each sweep
out = {**STAMPS, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* in separate passes over `bench.py --steps 20 --warmup 5` "
                 "(tools/gpu_profile_round.sh, tools/collect_profiles_round.py)",
       "pairs_per_step": pairs, "lab_mode": bench["config"].get("lab_arithmetic"),
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 (MI355X_MICROARCH.md, HBM)",
       "hbm_bytes_per_launch": {}, "algorithmic_bytes_per_launch": alg, "per_kernel": {}}
for k, d in per.items():
    e = dict(d)
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = e.pop("FETCH_SIZE"), e.pop("WRITE_SIZE")
        e["read_bytes"], e["write_bytes"] = int(2 * e["FETCH_SIZE_KB"] * 1024), int(e["WRITE_SIZE_KB"] * 1024)
        out["hbm_bytes_per_launch"][k] = e["read_bytes"] + e["write_bytes"]
        fused_psnr = "apply" in k and bench["config"].get("metrics") == ["psnr"] and out["lab_mode"] == "table"
        e["algorithmic_bytes"] = alg * 3 // 2 if fused_psnr else alg      # + the ground-truth plane of the fused per-frame PSNR
        e["traffic_over_algorithmic"] = (e["read_bytes"] + e["write_bytes"]) / e["algorithmic_bytes"]
    px = H * W * pairs * (2 if "moments" in k else 1)
    if "SQ_INSTS_VALU" in d:
        e["valu_instructions_per_pixel"] = d["SQ_INSTS_VALU"] * 64 / px
        e["all_counted_instructions_per_pixel"] = sum(d.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) * 64 / px
    out["per_kernel"][k] = e
json.dump(out, open(os.path.join(DST, ROUND + "_traffic.json"), "w"), indent=1)
print("wrote", ROUND + "_traffic.json", out["hbm_bytes_per_launch"])

# ---- IDT: HBM bytes per launch of its kernels (same gfx950 correction)
idt = defaultdict(dict)
for sub in ("idt_FETCH_SIZE", "idt_WRITE_SIZE"):
    if os.path.isdir(os.path.join(SRC, sub)):
        for k, d in counters(sub).items():
            if "idt_" in k:
                idt[k.split("ct::")[-1].split("(")[0].replace("void ", "")].update(d)
if idt:
    o = {**STAMPS, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/bench_idt.py (one pair per call and 16 per call mixed: per-launch means)",
         "correction": out["correction"], "per_kernel": {}}
    for k, d in idt.items():
        e = dict(d)
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            e["read_bytes_mean_per_launch"], e["write_bytes_mean_per_launch"] = int(2 * d["FETCH_SIZE"] * 1024), int(d["WRITE_SIZE"] * 1024)
        o["per_kernel"][k] = e
    json.dump(o, open(os.path.join(DST, ROUND + "_idt_traffic.json"), "w"), indent=1)
    print("wrote", ROUND + "_idt_traffic.json")

# ---- MFMA-busy of the CNN forwards: summarised on the GPU box from ONE run each (tools/summarize_pmc.py there)
for a, b in (("dcmcs3di_1080p_mfma_pmc.json", ROUND + "_dcmcs3di_1080p_mfma_pmc.json"), ("gmflow_960x540_mfma_pmc.json", ROUND + "_gmflow_960x540_mfma_pmc.json")):
    p = os.path.join(SRC, a)
    if os.path.exists(p) and os.path.getsize(p):
        j = json.load(open(p))
        assert j.get("source_stamp") == STAMP and not j.get("_ignored_runs"), (a, j.get("source_stamp"), j.get("_ignored_runs"))
        j.update(STAMPS)
        json.dump(j, open(os.path.join(DST, b), "w"), indent=1, sort_keys=True)
        print("wrote", b, j["_all_kernels"])

# ---- the persistent launch: kernel stats + HBM bytes per launch (float32 and uint8 instantiations)
stats_csv("trace_persist", ROUND + "_reinhard_persist_kernel_stats.csv")
pp = defaultdict(dict)
for sub in ("persist_FETCH_SIZE", "persist_WRITE_SIZE"):
    if os.path.isdir(os.path.join(SRC, sub)):
        for k, d in counters(sub).items():
            if "reinhard_persist_kernel" in k:
                pp[k.split("ct::rp::")[-1].split("(")[0]].update(d)
if pp:
    o = {**STAMPS, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/bench_reinhard_persist.py (16 pairs of 1080p per launch, PSNR fused)",
         "correction": out["correction"], "per_kernel": {}}
    for k, d in pp.items():
        e = dict(d)
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            e["read_bytes_per_launch"], e["write_bytes_per_launch"] = int(2 * d["FETCH_SIZE"] * 1024), int(d["WRITE_SIZE"] * 1024)
            in_b = 4 if "float" in k else 1
            e["algorithmic_bytes_per_launch"] = 16 * H * W * 3 * (3 * in_b + 4)        # target, reference, ground truth in; float32 result out
            e["traffic_over_algorithmic"] = (e["read_bytes_per_launch"] + e["write_bytes_per_launch"]) / e["algorithmic_bytes_per_launch"]
        o["per_kernel"][k] = e
    json.dump(o, open(os.path.join(DST, ROUND + "_reinhard_persist_traffic.json"), "w"), indent=1)
    print("wrote", ROUND + "_reinhard_persist_traffic.json")

def kernel_short(name):
    """'void ct::w4::conv_wino4_kernel<1, true>(ct::ConvArgs, int, ...)' -> 'w4::conv_wino4_kernel<1, true>' (the argument list also holds 'ct::')"""
    head = name.split("(")[0].replace("void ", "").strip()
    return head[4:] if head.startswith("ct::") else head


# ---- CNN forwards: HBM bytes per launch of every kernel (round 6; same gfx950 correction), DCMCS3DI 1080p and GMFlow 960x540
for tag, name in (("dc1080", "dcmcs3di_1080p"), ("gm960", "gmflow_960x540")):
    per = defaultdict(dict)
    n = {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        sub = tag + "_" + cname
        f = newest(sub, "*counter_collection.csv") if os.path.isdir(os.path.join(SRC, sub)) else None
        if f:
            agg = defaultdict(lambda: [0.0, 0])
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == cname:
                    a = agg[row["Kernel_Name"]]
                    a[0] += float(row["Counter_Value"]); a[1] += 1
            for k, (tot, cnt) in agg.items():
                per[k][cname] = tot / cnt
                n[k] = cnt
    if per:
        o = {**STAMPS, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/bench_%s (two timed forwards + warm-up; per-launch means)"
                                 % ("dcmcs3di.py 1080 1920 2" if tag == "dc1080" else "gmflow.py 540 960 2"),
             "correction": out["correction"], "per_kernel": {}}
        tot_r = tot_w = 0.0
        for k, d in per.items():
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d and "ct::" in k:
                short = kernel_short(k)
                r, w = 2 * d["FETCH_SIZE"] * 1024, d["WRITE_SIZE"] * 1024
                o["per_kernel"][short] = {"launches_in_run": n[k], "read_bytes_mean_per_launch": int(r), "write_bytes_mean_per_launch": int(w)}
                tot_r += r * n[k]; tot_w += w * n[k]
        # average launch duration of the same kernels from the kernel-trace statistics of the (separate, counter-free) stats run
        sf = newest("trace_" + tag, "*kernel_stats.csv")
        if sf:
            for row in csv.DictReader(open(sf)):
                short = kernel_short(row["Name"])
                if "ct::" in row["Name"] and short in o["per_kernel"]:
                    o["per_kernel"][short]["avg_duration_us"] = float(row["AverageNs"]) / 1e3
                    o["per_kernel"][short]["share_of_gpu_time_pct"] = float(row["Percentage"])
        o["all_ct_kernels_read_bytes_in_run"], o["all_ct_kernels_write_bytes_in_run"] = int(tot_r), int(tot_w)
        json.dump(o, open(os.path.join(DST, ROUND + "_" + name + "_traffic.json"), "w"), indent=1, sort_keys=True)
        print("wrote", ROUND + "_" + name + "_traffic.json")

# ---- the ResB convolution alone: both Winograd kernels (times, error, stall counters per step)
p = os.path.join(SRC, "conv_wino_bench.txt")
if os.path.exists(p):
    lines = ["source stamp " + STAMP, "== tools/bench_conv_ws.py 2 64 64 1080 1920 1.0, both Winograd kernels on one box"]
    lines += [l.rstrip() for l in open(p) if "amdgpu.ids" not in l]
    lines.append("== tools/pmc_conv_wino.sh: SQ / TCC counters of the same launches (per-launch means; SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over the waves)")
    for f in sorted(glob.glob(os.path.join(SRC, "form?_?.json"))):
        j = json.load(open(f))
        for k, v in j.items():
            if isinstance(v, dict) and "dispatches" in v:
                lines.append("%s  %s  %s" % (os.path.basename(f), k[:64], json.dumps({a: round(b, 1) for a, b in sorted(v.items()) if a != "dispatches"})))
    open(os.path.join(DST, ROUND + "_conv_wino.txt"), "w").write("\n".join(lines) + "\n")
    print("wrote", ROUND + "_conv_wino.txt")
