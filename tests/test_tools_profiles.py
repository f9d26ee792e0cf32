"""The evidence pipeline behind the profile-derived numbers of bench.py (round 3 averaged five rocprofv3 runs of five
builds): tools/summarize_pmc.py reads ONE run, tools/stamp.py ties a summary to the kernel sources it was measured on, and a summary
whose stamp is not this tree's is not quoted (bench.py prints null)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _write_run(d, pid, busy, gui, dur_ns, kernel="void ct::conv_ws_kernel<1, true>(ct::ConvArgs, int, int, int, int)"):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "%d_counter_collection.csv" % pid), "w") as f:
        f.write("Kernel_Name,Counter_Name,Counter_Value\n")
        for _ in range(3):
            f.write('"%s",SQ_VALU_MFMA_BUSY_CYCLES,%f\n"%s",GRBM_GUI_ACTIVE,%f\n' % (kernel, busy, kernel, gui))
    with open(os.path.join(d, "%d_kernel_trace.csv" % pid), "w") as f:
        f.write("Kernel_Name,Start_Timestamp,End_Timestamp\n")
        for i in range(3):
            f.write('"%s",%d,%d\n' % (kernel, 1000 * i, 1000 * i + dur_ns))


def test_summarize_pmc_reads_one_run_only(tmp_path):
    old, new = tmp_path / "prof" / "runA", tmp_path / "prof" / "runB"
    _write_run(str(old), 11, busy=0.9 * 1e6 / 8 * 1024, gui=1e6, dur_ns=500)          # a stale build: 0.9 busy
    time.sleep(0.05)
    _write_run(str(new), 22, busy=0.25 * 1e6 / 8 * 1024, gui=1e6, dur_ns=400)         # the newest run: 0.25 busy
    os.utime(str(new / "22_counter_collection.csv"), None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), str(tmp_path / "prof")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout)
    assert j["_run"].endswith("22_counter_collection.csv") and len(j["_ignored_runs"]) == 1 and "11_" in j["_ignored_runs"][0]
    assert abs(j["_all_kernels"]["mfma_busy_frac_time_weighted"] - 0.25) < 1e-9         # not the 0.575 an average of both would give
    k = [v for n, v in j.items() if "conv_ws_kernel" in n][0]
    assert abs(k["mfma_busy_frac"] - 0.25) < 1e-9 and k["dispatches"] == 3 and abs(k["avg_duration_us"] - 0.4) < 1e-9
    import stamp
    assert j["source_stamp"] == stamp.source_stamp()


def test_stamp_gates_profile_reads(tmp_path):
    import stamp
    cur = stamp.source_stamp()
    assert len(cur) == 16 and cur == stamp.source_stamp()
    files = stamp.source_files()
    assert any(f.endswith("linear.hip") for f in files) and any(f.endswith("ct_hip.h") for f in files) and any(f.endswith("Makefile") for f in files)
    good, stale, bare = tmp_path / "good.json", tmp_path / "stale.json", tmp_path / "bare.json"
    good.write_text(json.dumps({"source_stamp": cur, "x": 1}))
    stale.write_text(json.dumps({"source_stamp": "0123456789abcdef", "x": 2}))
    bare.write_text(json.dumps({"x": 3}))
    assert stamp.read_stamped(str(good))["x"] == 1
    assert stamp.read_stamped(str(stale)) is None and stamp.read_stamped(str(bare)) is None
    assert stamp.read_stamped(str(tmp_path / "missing.json")) is None


def test_committed_profiles_carry_a_stamp():
    """every JSON summary of rounds 4 and 5 under profiles/ says which sources it was measured on (it may be older than HEAD: then
    bench.py prints null for the numbers it would have quoted, which is the point)"""
    import glob
    js = [f for r in ("r04", "r05") for f in glob.glob(os.path.join(ROOT, "profiles", r + "_*.json")) if "bench" not in os.path.basename(f)]
    assert js, "no profile summaries committed"
    for f in js:
        j = json.load(open(f))
        assert isinstance(j.get("source_stamp"), str) and len(j["source_stamp"]) == 16, f
    for f in glob.glob(os.path.join(ROOT, "profiles", "r0[45]_*mfma_pmc.json")):
        j = json.load(open(f))
        assert j["_ignored_runs"] == [] and j["_run"].endswith("counter_collection.csv"), f


def test_no_build_artefacts_are_tracked():
    """ISA dumps and object files under csrc/build_* (make asm / tuning builds) never enter the history (+437 k lines
    twice); neither do shared objects or anything under gpurun_out/"""
    import subprocess
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        import pytest
        pytest.skip("not a git checkout (the GPU box runs a snapshot)")
    files = subprocess.run(["git", "-C", ROOT, "ls-files"], capture_output=True, text=True, check=True).stdout.splitlines()
    bad = [f for f in files if "/csrc/build" in f or f.startswith("gpurun_out/") or f.endswith((".so", ".o", ".s", ".hsaco", ".bc", ".hipi"))]
    assert not bad, bad[:10]
