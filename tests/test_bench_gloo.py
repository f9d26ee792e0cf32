"""A CPU dry run of bench.py's multi-rank path, launched exactly as the driver launches it for N > 1 (torch.distributed.run, one
process per rank, rendezvous on 127.0.0.1) but with CT_BENCH_DEVICE=cpu: gloo in place of RCCL and a stub in place of the HIP
call.  It exercises init_process_group -> the ranks_seen all-reduce -> the warm-up gather -> the timed steps -> the gather of the
[steps, pairs, metrics] table -> the max-over-ranks all-reduce -> rank 0's ONE JSON line, so that the first 8-GPU lease cannot
die on a typo in that plumbing.  No hardware scaling is measured or claimed by this test."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(world, steps, pairs):
    env = dict(os.environ, CT_BENCH_DEVICE="cpu", OMP_NUM_THREADS="1")
    args = ["bench.py", "--gpus", str(world), "--steps", str(steps), "--warmup", "1", "--pairs", str(pairs), "--init-seconds", "0", "--no-extra",
            "--no-cpu-baseline"]
    if world == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                             # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


def test_bench_world2_dry_run():
    steps, pairs = 3, 1
    j = _bench(2, steps, pairs)
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["ranks_seen"] == 2
    assert j["steps"] == steps and j["warmup"] == 1 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert j["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert j["config"]["pairs_per_step_per_gpu"] == pairs and j["value"] > 0 and j["ms_per_step"] > 0
    # the gathered table: [world, steps, pairs, (mse, PSNR)]; the stub writes frame_id + 0.001 * step into both columns,
    # frame ids of rank r are r + world * i (frame f -> rank f % world)
    assert j["gathered_rows"] == 2 * steps * pairs
    want = sum(2 * ((r + 2 * i) + 0.001 * k) for r in range(2) for k in range(steps) for i in range(pairs))
    assert abs(j["gathered_checksum"] - want) < 1e-9
    assert j["roofline"] is None and j["cpu_baseline"] is None    # nothing is measured in a dry run


def test_bench_world1_dry_run_and_flag_mismatch():
    j = _bench(1, 2, 1)
    assert j["dry_run"] is True and j["n_gpus"] == 1 and j["ranks_seen"] == 1
    env = dict(os.environ, CT_BENCH_DEVICE="cpu")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)      # --gpus must match the launcher's world size
