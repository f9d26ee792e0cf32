#!/usr/bin/env python3
"""bench.py -- stereopairs/s of the colour-transfer hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 the
driver launches one rank per GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): methods.linear.color_transfer_between_images (Reinhard) on
synthetic 1920x1080 float32 RGB pairs, inputs resident in HBM, `--pairs` pairs per step
(one step = one ct_reinhard_f32 call -- statistics sweep over 2*pairs images, apply sweep with
the statistics finished in its prologue -- plus the per-frame metric of Runner.test_step,
ct_frame_psnr_f32 of the corrected frames against resident ground-truth frames; `--metrics
psnr,ssim,fsim,icid` adds the others).  Frames shard across ranks (frame f -> rank f % world);
the [frames, n_metrics] table is gathered with ONE RCCL all_gather at the end of the timed
region (configs[4]).  `value` = all ranks' pairs / max-over-ranks wall time.

Extra keys: `roofline` (dominant kernel, HIP events per launch), `cpu_baseline` (the numpy oracle
of the same function, rank 0, N=1 only), `extra` (MK / Xiao / IDT rates, informational).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "color-transfer_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
from stamp import read_stamped, source_stamp  # noqa: E402  (profiles/*.json are quoted only when measured on this tree's sources)

H, W = 1080, 1920
N_PIX = H * W
PLANE_F32 = N_PIX * 3 * 4                    # 24 883 200 B
ALGO_BYTES_PER_PAIR = 3 * PLANE_F32          # read target, read reference, write output
PROFILE_ROUND = "r06"                         # profiles/<round>_*: quoted only when their source stamp is this tree's
HBM_PEAK = 8.0e12                            # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
BASELINE_METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
# arithmetic the headline path computes in, per Lab mode (ct_set_lab_mode)
DTYPE = {"table": "f32 (float32 I/O; float32 difference forms in both sweeps, float64 only in the per-frame statistics records)",
         "exact": "f64 (float32 I/O, float64 arithmetic)"}


def synth_frames(frame_ids, device):
    """SURVEY 8d synthetic inputs: rng = default_rng(1234 + frame); uniform float32 HWC (target, reference, ground truth)."""
    t = np.empty((len(frame_ids), H, W, 3), dtype=np.float32)
    r, g = np.empty_like(t), np.empty_like(t)
    for i, f in enumerate(frame_ids):
        rng = np.random.default_rng(1234 + f)
        t[i] = rng.random((H, W, 3), dtype=np.float32)
        r[i] = rng.random((H, W, 3), dtype=np.float32)
        g[i] = rng.random((H, W, 3), dtype=np.float32)
    return torch.from_numpy(t).to(device), torch.from_numpy(r).to(device), torch.from_numpy(g).to(device)


def cpu_baseline(n_pairs=4):
    """The numpy oracle (float64 port of methods.linear.color_transfer_between_images) on host cores."""
    from oracle import linear as olin
    rng = np.random.default_rng(1234)
    t = rng.random((H, W, 3), dtype=np.float32)
    r = rng.random((H, W, 3), dtype=np.float32)
    olin.color_transfer_between_images(t[:270], r[:270])          # warm-up (quarter frame)
    times = []
    for _ in range(n_pairs):
        t0 = time.perf_counter()
        olin.color_transfer_between_images(t, r)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "stereopairs/s", "cores": 1, "kind": "port",
            "sample": "%d x 1080p pair, oracle.linear.color_transfer_between_images (numpy float64, single thread), "
                      "median %.3f s/pair; host has %d cores" % (n_pairs, med, os.cpu_count() or 0)}


def video_stream(n_frames, rank, world):
    """BASELINE.json configs[4] through the PRODUCT'S entry point: `utils.cli test --config configs/others.yaml --model.metrics psnr
    --data.synthetic video_u8 --data.n_frames N` (the reference's way in, utils/cli.py:1-3; methods/__init__.py:18-40).  This
    rank's share of an n_frames 1080p stereo video (frame f -> rank f % world) arrives as uint8 frames in pinned host memory (the
    host-buffer hand-over of the drop-in boundary: PCIe is inside this measurement, unlike `value`), eight stereo triples per
    upload (149 MB) on a copy stream two uploads ahead; Runner.test_group corrects and scores eight pairs per call
    (ct_reinhard_psnr_u8 reads the bytes; clamp + PSNR of Runner.test_step fused); ONE gather of the [frames, metrics] table
    at the end.  Timed by utils.cli.main itself (barrier + synchronize on both sides, max over ranks)."""
    from utils import cli
    timing = {}
    argv = ["test", "--config", os.path.join(PKG, "configs", "others.yaml"), "--model.metrics", "psnr", "--data.data_dir", "null",
            "--data.synthetic", "video_u8", "--data.n_frames", str(n_frames), "--data.height", str(H), "--data.width", str(W)]
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):        # the CLI prints its means; this program's stdout carries ONE JSON line
        table = cli.main(argv, timing=timing)
    dt, n_local = timing["seconds"], timing["frames_local"]
    return {"frames": n_frames, "host_dtype": "u8", "frames_per_s": n_frames / dt, "ms_per_frame_per_gpu": dt / max(n_local, 1) * 1e3,
            "h2d_GB_per_s_per_gpu": timing["h2d_bytes"] / dt / 1e9, "h2d_bytes_per_frame": 3 * H * W * 3,
            "frames_per_copy_and_call": timing["frames_per_call"], "pcie_gen5_x16_GB_per_s": 63.0, "mean_psnr": float(table[:, 0].mean()),
            "entry": "utils.cli.main(%s) -> Runner.test_group -> ct_reinhard_psnr_u8 (persistent launch, reads the bytes)" % " ".join(argv[3:]),
            "note": "measured inside utils.cli.main (the product's CLI), not by a loop of bench.py; grouped path: %s" % timing["grouped"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--pairs", type=int, default=16, help="stereopairs per step per GPU (one launch pair sweeps them all: the ~8 us a launch costs before it streams is paid once per step)")
    ap.add_argument("--metrics", default="psnr", help="per-frame metrics inside the timed region: psnr[,ssim,fsim,icid] or none")
    ap.add_argument("--init-seconds", type=float, default=0.25, help="clock / code-object initialisation before the warm-up steps")
    ap.add_argument("--video", type=int, default=0, help="also stream this many 1080p frames from pinned host memory (configs[4]; 0 = the default "
                    "1000 on single-GPU runs with the extras, off otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    from utils.sharding import pin_rank_to_cpus
    pin_rank_to_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))      # one CPU slice per rank, before any GPU call
    # CT_BENCH_DEVICE=cpu: a DRY RUN of this file's multi-rank plumbing on CPU tensors with the gloo backend (tests/test_bench_gloo.py):
    # process group -> ranks_seen all-reduce -> warm-up gather -> timed steps -> gather -> max-over-ranks -> the JSON line, with a
    # stub in place of the HIP call.  It measures nothing and says so ("dry_run": true); the product path has no CPU fallback.
    on_cpu = os.environ.get("CT_BENCH_DEVICE", "cuda") == "cpu"
    if on_cpu:
        device = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        assert dist.get_world_size() == args.gpus, "--gpus %d but the communicator has %d ranks" % (args.gpus, dist.get_world_size())
        one = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(one)                            # counted by the communicator itself, not read from the environment
        ranks_seen = int(one.item())
        assert ranks_seen == args.gpus

    import ct_hip
    import methods.linear as lin
    ct_hip.lib()                                   # fail loudly if the HIP library is missing

    B, K, Wm = args.pairs, args.steps, args.warmup
    names = [m for m in args.metrics.split(",") if m and m != "none"]
    assert all(m in ("psnr", "ssim", "fsim", "icid") for m in names), names
    # frames owned by this rank: f % world == rank; a ring of `B` resident pairs is re-used every step
    frame_ids = [rank + world * i for i in range(B)]
    tgt, ref, gt = synth_frames(frame_ids, device)
    out = torch.empty_like(tgt)
    gt_nchw = gt.permute(0, 3, 1, 2).contiguous() if any(m in names for m in ("ssim", "fsim", "icid")) else None
    n_m = max(len(names), 1)
    # this rank's [frames, n_metrics] table.  PSNR-only (default): ct_reinhard_psnr_f32 writes its (mse, PSNR) records straight
    # into the table -- no torch op at all inside the timed region besides the gather
    psnr_rec = torch.zeros((K, B, 2), dtype=torch.float64, device=device) if names == ["psnr"] else None
    metrics = psnr_rec if psnr_rec is not None else torch.zeros((K, B, n_m), dtype=torch.float64, device=device)
    gathered = torch.empty((world,) + tuple(metrics.shape), dtype=torch.float64, device=device) if world > 1 else None

    def step(i):
        if on_cpu:                                  # dry run: a stand-in that fills this step's rows of the table
            metrics[i] = torch.tensor(frame_ids, dtype=torch.float64).view(B, 1) + 0.001 * i
            return
        if psnr_rec is not None:                    # transfer + per-frame PSNR in one fused call (ct_reinhard_psnr_f32)
            ct_hip.reinhard_psnr(tgt, ref, gt, out=out, psnr_out=psnr_rec[i])
            return
        ct_hip.reinhard(tgt, ref, out=out)
        for j, m in enumerate(names):
            if m == "psnr":
                metrics[i, :, j] = ct_hip.frame_psnr(out, gt)[:, 1]            # layout-agnostic: HWC against HWC
            else:
                o = out.permute(0, 3, 1, 2).contiguous()
                metrics[i, :, j] = {"ssim": ct_hip.frame_ssim, "fsim": ct_hip.frame_fsim, "icid": ct_hip.frame_icid}[m](o, gt_nchw)

    def barrier():
        if world > 1:
            dist.barrier()
        if not on_cpu:
            torch.cuda.synchronize()

    # initialisation, not warm-up: every op of the timed region runs once (code objects load on first launch -- round 2's
    # driver line paid 42 ms for the first launch of a torch copy kernel inside its 50 ms timed region) and the GPU is
    # brought to its sustained clock with INIT_S seconds of the same step; the W warm-up steps follow as the contract says
    step(0)
    if world > 1:
        dist.all_gather_into_tensor(gathered.flatten(0, 1), metrics)      # warms the communicator (concatenated form: every backend takes it)
    barrier()
    t_init = time.perf_counter()
    init_steps = 1
    while time.perf_counter() - t_init < args.init_seconds:
        for _ in range(8):
            step(0)
        init_steps += 8
        if not on_cpu:
            torch.cuda.synchronize()
    for i in range(Wm):
        step(i % K)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    if world > 1:
        dist.all_gather_into_tensor(gathered.flatten(0, 1), metrics)      # the per-frame metric gather (RCCL over xGMI)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    pairs_total = B * K * world
    value = pairs_total / dt

    # ---- per-kernel roofline: HIP events around each launch group on the launch stream ----------
    roof = None
    roof_cnn = None
    extra = {}
    if rank == 0 and not on_cpu:
        # exact per-kernel durations: the library records HIP events on the launch stream right before / after
        # moments_kernel<float,true> and reinhard_apply_kernel<float,false> of the same fused call that `value` times
        n_prof = min(K, 50)
        ts, ta = [], []
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for e in ev:
            e.record()
        torch.cuda.synchronize()
        ct_hip.profile_events(ev)
        fused_psnr = psnr_rec is not None           # the same call the timed region makes
        for i in range(n_prof):
            if fused_psnr:
                ct_hip.reinhard_psnr(tgt, ref, gt, out=out, psnr_out=psnr_rec[0])
            else:
                ct_hip.reinhard(tgt, ref, out=out)
            torch.cuda.synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) * 1e-3)
            ta.append(ev[2].elapsed_time(ev[3]) * 1e-3)
        # the same two sweeps without the fused metric (ct_reinhard_f32): what the apply sweep does on its own two planes
        ts0, ta0 = [], []
        if fused_psnr:
            for i in range(min(n_prof, 20)):
                ct_hip.reinhard(tgt, ref, out=out)
                torch.cuda.synchronize()
                ts0.append(ev[0].elapsed_time(ev[1]) * 1e-3)
                ta0.append(ev[2].elapsed_time(ev[3]) * 1e-3)
        ct_hip.profile_events(None)
        t_stats, t_apply = float(np.mean(ts)), float(np.mean(ta))
        table = ct_hip.lab_mode() == "table"
        k_stats = "lab_moments_lut_kernel" if table else "moments_kernel<float,true>"
        k_apply = "reinhard_apply_lut_kernel<false>" if table else "reinhard_apply_kernel<float,false>"
        # algorithmic bytes per launch: the statistics sweep reads the 2B images once; the apply sweep reads B targets and
        # writes B results, and -- when the per-frame PSNR rides on it (table mode) -- also reads the B ground-truth frames
        # (the metric's one compulsory plane; the result it compares is still in registers)
        # algorithmic bytes per launch on SURVEY 8(d)'s planes: the statistics sweep reads the 2B images once; the apply sweep
        # reads B targets and writes B results.  When the per-frame PSNR rides on the apply sweep (table mode) the launch also
        # reads the B ground-truth frames: real traffic, but the metric's plane, not the transfer's -- reported as a
        # secondary figure (`frac_with_metric_plane`), never as `frac`.
        metric_planes = 1 if (fused_psnr and table) else 0
        kern = {k_stats: {"bytes": 2 * B * PLANE_F32, "t": t_stats, "extra": 0},
                k_apply: {"bytes": 2 * B * PLANE_F32, "t": t_apply, "extra": metric_planes * B * PLANE_F32}}
        dom = max(kern, key=lambda k: kern[k]["t"])
        ach = kern[dom]["bytes"] / kern[dom]["t"]
        # HBM traffic per launch of the dominant kernel from the committed PMC profile (separate --pmc passes,
        # FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md "HBM"), valid for the same pairs-per-step only
        traffic = None
        tj = read_stamped(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_traffic.json"))     # None unless measured on THIS tree's kernel sources
        if tj and tj.get("pairs_per_step") == B and tj.get("lab_mode") == ct_hip.lab_mode():
            traffic = tj.get("hbm_bytes_per_launch", {}).get(dom)
        t_kernels = t_stats + t_apply
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": ach / HBM_PEAK, "traffic": traffic,
                "traffic_source": ("profiles/" + PROFILE_ROUND + "_traffic.json (rocprofv3 --pmc passes of this command on a build with source stamp %s; not live)" % source_stamp()) if traffic
                else "no PMC profile of this build under profiles/ (stamp %s)" % source_stamp(),
                "algorithmic_bytes_per_launch": kern[dom]["bytes"],
                "algorithmic_bytes_note": "2 float32 planes x %d pairs (SURVEY 8d): %s" % (
                    B, "target read + result write" if dom == k_apply else "target + reference read"),
                "avg_launch_s": kern[dom]["t"],
                "frac_with_metric_plane": (kern[dom]["bytes"] + kern[dom]["extra"]) / kern[dom]["t"] / HBM_PEAK,
                "lab_arithmetic": ct_hip.lab_mode(),
                "kernels": {k: {"GB/s": v["bytes"] / v["t"] / 1e9, "frac": v["bytes"] / v["t"] / HBM_PEAK, "avg_launch_us": v["t"] * 1e6,
                                "algorithmic_bytes_per_launch": v["bytes"],
                                "GB/s_with_metric_plane": (v["bytes"] + v["extra"]) / v["t"] / 1e9} for k, v in kern.items()},
                "path": {"algorithmic_bytes_per_pair": ALGO_BYTES_PER_PAIR,
                         "GB/s": ALGO_BYTES_PER_PAIR * value / world / 1e9,
                         "frac_of_peak": ALGO_BYTES_PER_PAIR * value / world / HBM_PEAK,
                         "sum_of_event_timed_kernels_ms": t_kernels * 1e3,
                         "note": "3 compulsory planes per pair over the whole step; value includes the per-frame metric (%s)" % (",".join(names) or "none")}}
        if ta0:
            roof["without_metric"] = {"note": "ct_reinhard_f32 (no fused PSNR), same tensors: the apply launch then moves exactly its two planes",
                                      "apply_avg_launch_us": float(np.mean(ta0)) * 1e6, "apply_frac": 2 * B * PLANE_F32 / float(np.mean(ta0)) / HBM_PEAK,
                                      "stats_avg_launch_us": float(np.mean(ts0)) * 1e6, "stats_frac": 2 * B * PLANE_F32 / float(np.mean(ts0)) / HBM_PEAK}
        # consistency of the timed region with the kernels it is made of (PSNR-only mode: two sweeps + a 5 us finish)
        if fused_psnr and dt / K > 1.3 * t_kernels + 30e-6:
            roof["warning"] = "ms_per_step %.3f exceeds 1.3 x the event-timed kernels (%.3f ms): host-side cost inside the timed region" % (
                dt / K * 1e3, t_kernels * 1e3)

        if not args.no_extra and world == 1:        # informational rates of the other paths: single-GPU runs only
            def rate(fn, n=10):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return n / (time.perf_counter() - t0)

            def forward_ms(fn, n=10, warm=3):
                """one whole forward per sample, HIP events on the launch stream: median and min of n after `warm` warm-ups"""
                for _ in range(warm):
                    fn()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
                for a, b in ev:
                    a.record()
                    fn()
                    b.record()
                torch.cuda.synchronize()
                ms = sorted(a.elapsed_time(b) for a, b in ev)
                return {"median_ms": float(np.median(ms)), "min_ms": ms[0], "n": n, "warmup": warm, "timer": "HIP events"}
            extra["reinhard_pairs_per_s_transfer_only"] = B * rate(lambda: ct_hip.reinhard(tgt, ref, out=out), n=200)
            # two batches in flight on two streams: the next batch's statistics sweep (issue bound) runs into the tail of this batch's
            # apply sweep (HBM bound).  The headline keeps ONE stream (its per-kernel rooflines stay clean); this is what a pipelined
            # caller gets from the same entry (tools/bench_reinhard_streams.py)
            if gt is not None:
                two = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
                bufs = [(torch.empty_like(out), torch.zeros((B, 2), dtype=torch.float64, device=device)) for _ in range(2)]
                def pipelined(n):
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    for i in range(n):
                        with torch.cuda.stream(two[i & 1]):
                            ct_hip.reinhard_psnr(tgt, ref, gt, out=bufs[i & 1][0], psnr_out=bufs[i & 1][1])
                    torch.cuda.synchronize(device)
                    return n / (time.perf_counter() - t0)
                pipelined(20)
                extra["reinhard_pairs_per_s_with_psnr_two_streams"] = B * pipelined(200)
                del bufs, two
            # how much of a step is launch ramp / tail: the same fused call over 8 and 32 pairs (the headline keeps 16)
            by_pairs = {B: value}
            for nb in (8, 32):
                ids = [rank + world * i for i in range(nb)]
                tb, rb, gb = synth_frames(ids, device)
                ob, pb = torch.empty_like(tb), torch.zeros((nb, 2), dtype=torch.float64, device=device)
                by_pairs[nb] = nb * rate(lambda: ct_hip.reinhard_psnr(tb, rb, gb, out=ob, psnr_out=pb), n=100)
                del tb, rb, gb, ob, pb
            extra["reinhard_pairs_per_s_with_psnr_by_pairs_per_call"] = by_pairs
            # the one-launch form (csrc/reinhard_persist.hip): float32 frames by name, uint8 frames through the u8 front door
            pr = torch.zeros((B, 2), dtype=torch.float64, device=device)
            extra["reinhard_persist_f32_pairs_per_s_with_psnr"] = B * rate(lambda: ct_hip.reinhard_persist(tgt, ref, gt=gt, out=out, psnr_out=pr), n=100)
            t8, r8, g8 = ((x * 255).round().to(torch.uint8) for x in (tgt, ref, gt))
            extra["reinhard_u8_pairs_per_s_with_psnr"] = B * rate(lambda: ct_hip.reinhard_persist(t8, r8, gt=g8, out=out, psnr_out=pr), n=100)
            extra["reinhard_u8_pairs_per_s_transfer_only"] = B * rate(lambda: ct_hip.reinhard_persist(t8, r8, out=out), n=100)
            extra["reinhard_u8_note"] = ("ct_reinhard_psnr_u8: uint8 frames (what the reference's datasets deliver, utils/data.py:84) read as bytes; "
                                         "algorithmic bytes per pair 2 x 6.2 MB in + 24.9 MB out")
            del t8, r8, g8, pr
            o_nchw, g_nchw = out.permute(0, 3, 1, 2).contiguous(), gt.permute(0, 3, 1, 2).contiguous()
            extra["frames_per_s_psnr"] = B * rate(lambda: ct_hip.frame_psnr(out, gt), n=100)
            extra["frames_per_s_ssim"] = B * rate(lambda: ct_hip.frame_ssim(o_nchw, g_nchw), n=100)
            extra["frames_per_s_icid"] = B * rate(lambda: ct_hip.frame_icid(o_nchw, g_nchw), n=100)
            extra["frames_per_s_fsim"] = B * rate(lambda: ct_hip.frame_fsim(o_nchw, g_nchw), n=20)
            del o_nchw, g_nchw
            extra["mk_pairs_per_s_f64out_hostalgebra"] = rate(
                lambda: lin.monge_kantorovitch_color_transfer_cuda(tgt[0], ref[0], host_algebra=True))
            mk_out = torch.empty(tgt.shape, dtype=torch.float32, device=device)
            extra["mk_pairs_per_s_f32out_device_algebra"] = B * rate(
                lambda: lin.monge_kantorovitch_color_transfer_cuda(tgt, ref, out_dtype=torch.float32, out=mk_out), n=20)
            extra["mk_frac_hbm_peak"] = ALGO_BYTES_PER_PAIR * extra["mk_pairs_per_s_f32out_device_algebra"] / HBM_PEAK
            del mk_out
            extra["xiao_pairs_per_s_f64out_hostalgebra"] = rate(
                lambda: lin.color_transfer_in_correlated_color_space_cuda(tgt[0], ref[0]))
            import methods.iterative as it
            rots = it.draw_rotations(4, seed=0)
            idt_out = torch.empty(tgt.shape, dtype=torch.float64, device=device)
            idt = rate(lambda: it.iterative_distribution_transfer_cuda(tgt[0], ref[0], rotations=rots, out=idt_out[0]), n=10)
            extra["idt_pairs_per_s_f64_one_pair_per_call"] = idt
            idt = B * rate(lambda: it.iterative_distribution_transfer_cuda(tgt, ref, rotations=rots, out=idt_out), n=10)
            extra["idt_pairs_per_s_f64"] = idt                           # B pairs per call, like the headline
            extra["idt_frac_hbm_peak"] = 920678400 * idt / HBM_PEAK      # float64 working image
            # bytes the kernels actually move per pair: float32 frames are read as float32 (min/max x2, histogram of iteration
            # 0, reference in every histogram, apply 0 input) = 8 float32 planes of 24.9 MB + 10 float64 planes of 49.8 MB
            extra["idt_frac_hbm_peak_bytes_moved"] = (8 * 4 + 10 * 8) * 3 * H * W * idt / HBM_PEAK
            del idt_out
            # the same two paths as roofline objects: bound, achieved, peak, and the counter traffic of their
            # kernels from the committed profile of THIS build (null otherwise)
            itj = read_stamped(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_idt_traffic.json")) or {}
            idt_moved = (8 * 4 + 10 * 8) * 3 * H * W
            idt_traffic = None
            if itj.get("per_kernel"):
                pk = itj["per_kernel"]
                idt_traffic = {k: v.get("read_bytes_mean_per_launch", 0) + v.get("write_bytes_mean_per_launch", 0) for k, v in pk.items()}
            extra["rooflines_other_paths"] = {
                "idt": {"bound": "hbm", "achieved": idt_moved * idt / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": idt_moved * idt / HBM_PEAK,
                        "bytes_per_pair": idt_moved, "pairs_per_s": idt,
                        "bytes_note": "what the 14 launches of a call move per pair: 8 float32 + 10 float64 planes (SURVEY 8d's unfused figure is 920.7 MB: "
                                      "frac %.3f on that)" % (920678400 * idt / HBM_PEAK),
                        "traffic_mean_bytes_per_launch_by_kernel": idt_traffic,
                        "traffic_source": ("profiles/%s_idt_traffic.json (separate --pmc passes, this build's stamp; launches of 1 and of 16 pairs mixed)" % PROFILE_ROUND)
                        if idt_traffic else None},
                "mk": {"bound": "hbm", "achieved": ALGO_BYTES_PER_PAIR * extra["mk_pairs_per_s_f32out_device_algebra"] / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                       "frac": extra["mk_frac_hbm_peak"], "bytes_per_pair": ALGO_BYTES_PER_PAIR, "pairs_per_s": extra["mk_pairs_per_s_f32out_device_algebra"],
                       "bytes_note": "target + reference read by the moments sweep, target re-read and the result written by the affine sweep are 4 "
                                     "planes moved for SURVEY 8d's 3; priced on the 3", "traffic": None}}
            # configs[2]: DCMCS3DI forward, random init, 512x512.  The convolutions and the attention run float32 operands as two
            # fp16 pieces with three MFMAs per product (float32-grade accuracy, csrc/conv_ws.hip, conv_split.hip, attention16.hip);
            # rates are quoted in algorithmic (float32) FLOPs, and x3 (MFMA flops issued) against the dense 16-bit peak.
            ws16 = ct_hip.conv_ws16() and ct_hip.conv_mode() == "split"
            extra["cnn_conv_arithmetic"] = (
                "float32 operands, float32 accumulate; convolutions (Winograd F(2x2,3x3) conv_wino for the ResB convs -- conv_ws, the "
                "direct weight-stationary form, with CT_HIP_CONV_WINO=0 --, tile kernel conv_split elsewhere), attention (attention16) and the FFN linears (linear_ws16): 2 fp16 pieces with power-of-two "
                "scales, 3 fp16 MFMAs per product; the 128 -> 128 token projections: 3 bf16 pieces, 6 MFMAs per product"
                if ws16 else "f32 as 3 bf16 pieces, 6 bf16 MFMAs per product, f32 accumulate (%s mode)" % ct_hip.conv_mode())
            # MFMA flops issued per algorithmic (float32) flop in the dominant convs: 3 products of fp16 pieces (6 of bf16 pieces), and
            # with the Winograd form (default since round 5, csrc/conv_wino.hip) 16 instead of 36 multiplications per 2x2 tile: 3 / 2.25
            mfma_per_flop = (3.0 / 2.25 if ct_hip.conv_wino() else 3.0) if ws16 else 6.0
            from methods.dcmcs3di import DCMCS3DI
            torch.manual_seed(0)
            net = DCMCS3DI().to(device).eval()
            l512, r512 = torch.rand(1, 3, 512, 512, device=device), torch.rand(1, 3, 512, 512, device=device)
            t512 = forward_ms(lambda: net(l512, r512, inference=True))
            dc = 1e3 / t512["median_ms"]
            extra["dcmcs3di_512_forward"] = t512
            flop = 512 * 512 * (6591040 + 390 * 512)
            extra["dcmcs3di_512_pairs_per_s_f32"] = dc
            extra["dcmcs3di_512_tflops"] = flop * dc / 1e12
            extra["dcmcs3di_512_frac_16bit_mfma_peak_issued"] = mfma_per_flop * flop * dc / 2.5e15   # MFMA flops issued per algorithmic flop
            # the size BASELINE.json's metric names: 1920x1080 (H*W*(6591040 + 390*W) FLOP/pair)
            l1080, r1080 = torch.rand(1, 3, H, W, device=device), torch.rand(1, 3, H, W, device=device)
            t1080 = forward_ms(lambda: net(l1080, r1080, inference=True))
            dc2 = 1e3 / t1080["median_ms"]
            flop2 = H * W * (6591040 + 390 * W)
            extra["dcmcs3di_1080p_pairs_per_s_f32"] = dc2
            # the second roofline object: the CNN half of BASELINE.json's metric.  bound = the bf16 matrix pipe (the one the
            # kernels run on: 6 bf16 MFMAs per float32 product in the convolutions and the q.k scores); `frac` = MFMA-busy from
            # the committed counter profile of the same forward (SQ_VALU_MFMA_BUSY_CYCLES over all kernels of the run, time
            # weighted, profiles/r02_dcmcs3di_1080p_mfma_pmc.json); achieved = issued bf16-MFMA-equivalent work, live.
            busy = read_stamped(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_dcmcs3di_1080p_mfma_pmc.json")) or {}      # {} unless measured on this tree's sources
            traf = read_stamped(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_dcmcs3di_1080p_traffic.json")) or {}
            # the dominant kernel: the second convolution of a ResB (no activation, skip tensor; reference pasmnet/backbone.py:8-15) --
            # conv_wino4.hip since round 6, conv_wino.hip with CT_HIP_WINO_FORM=1
            kname = "w4::conv_wino4_kernel<0, true>" if os.environ.get("CT_HIP_WINO_FORM", "0") != "1" else "conv_wino_kernel<0, true>"
            kdom = [v for k, v in busy.items() if kname.split("::")[-1] in k]
            tk = (traf.get("per_kernel") or {}).get(kname) or {}
            act = 2 * 64 * H * W * 4                      # one 2-view 64-channel float32 activation: 1.062 GB
            # input + skip read, output written = the convolution's own three planes; the profile's per-launch means run over the
            # forward's 25 such launches: 19 on both views, 6 (the transfer blocks) on one
            conv_bytes = 3 * act * (19 * 2 + 6 * 1) // (25 * 2)
            # ResB convolutions of a forward: 19 blocks on both views (18 extraction + the matcher's head), 6 on one view (transfer); per
            # block one convolution moves 2 planes (in, out) and one 3 (in, skip, out)
            resb_bytes = (19 * 2 + 6 * 1) * 5 * (act // 2)
            hbm = None
            if tk.get("avg_duration_us"):
                hbm = {"dominant_kernel": kname, "avg_launch_us": tk["avg_duration_us"], "share_of_gpu_time_pct": tk.get("share_of_gpu_time_pct"),
                       "algorithmic_bytes_per_launch": conv_bytes,
                       "algorithmic_bytes_note": "mean over the forward's 25 launches of this kernel (19 x 3.185 GB on two views, 6 x 1.593 GB on one)",
                       "GB/s": conv_bytes / (tk["avg_duration_us"] * 1e-6) / 1e9,
                       "frac_of_hbm_peak": conv_bytes / (tk["avg_duration_us"] * 1e-6) / HBM_PEAK,
                       "traffic": tk["read_bytes_mean_per_launch"] + tk["write_bytes_mean_per_launch"],
                       "traffic_over_algorithmic": (tk["read_bytes_mean_per_launch"] + tk["write_bytes_mean_per_launch"]) / conv_bytes,
                       "source": "profiles/%s_dcmcs3di_1080p_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gfx950 correction applied) + the "
                                 "kernel-trace statistics of the same forward, build with source stamp %s (not live)" % (PROFILE_ROUND, source_stamp())}
            roof_cnn = {"bound": "mfma", "workload": "dcmcs3di forward, random init, 1 pair of 1920x1080, float32 I/O",
                        "dtype": "16-bit MFMA pipe (float32 operands as 2 fp16 pieces / 3 MFMAs per product in the convolutions and the "
                                 "attention), f32 accumulate",
                        "achieved": mfma_per_flop * flop2 * dc2 / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                        "frac_flops": mfma_per_flop * flop2 * dc2 / 2.5e15,
                        "frac": busy.get("_all_kernels", {}).get("mfma_busy_frac_time_weighted"),
                        "frac_source": ("profiles/" + PROFILE_ROUND + "_dcmcs3di_1080p_mfma_pmc.json: ONE rocprofv3 --pmc run of the same forward on a build with source "
                                        "stamp %s (not live)" % source_stamp()) if busy else
                                       "no PMC profile of this build under profiles/ (source stamp %s): frac is null, never a stale number" % source_stamp(),
                        "forward": t1080,
                        "frac_definition": "MFMA-busy: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), time-weighted over every "
                                           "kernel of the forward",
                        "dominant_kernel": kname,
                        "dominant_kernel_mfma_busy": kdom[0].get("mfma_busy_frac") if kdom else None,
                        # which roof, and how far: the forward is NOT near the matrix roof and cannot be as an unfused
                        # float32 design -- its ceiling is HBM
                        "which_roof": "neither yet: with float32 activations in HBM every ResB convolution streams 2-3 activation planes, so the "
                                      "unfused design is capped by HBM long before the matrix pipe (ceiling_pairs_per_s_unfused_f32); the dominant "
                                      "kernel runs at hbm_frac_dominant_kernel of the HBM peak on its own planes and is itself bound by the "
                                      "issue of one wave per SIMD (vector + LDS + memory instructions ~4000 of ~7000 cycles per 2-row step, "
                                      "the rest stalls; profiles/" + PROFILE_ROUND + "_conv_wino.txt, DESIGN.md 4.4)",
                        "hbm_dominant_kernel": hbm,
                        "hbm_frac_dominant_kernel": hbm["frac_of_hbm_peak"] if hbm else None,
                        "traffic": hbm["traffic"] if hbm else None,
                        "ceiling_pairs_per_s_unfused_f32": {"resb_activation_bytes_per_pair": resb_bytes,
                                                            "at_hbm_peak_8.0_TB/s": HBM_PEAK / resb_bytes,
                                                            "at_the_5.5_TB/s_a_copy_streams_on_this_chip": 5.5e12 / resb_bytes,
                                                            "note": "the 50 ResB convolutions alone (19 blocks on two views, 6 on one; 2 + 3 planes "
                                                                    "per block), before attention / first / last layers: no unfused float32 "
                                                                    "forward exceeds this; north_star's 0.40 MFMA needs the hidden rows of a "
                                                                    "block to stay on the CU (DESIGN.md 7)"},
                        "frac_note": "MFMA-busy is not comparable with rounds 2-4: as Winograd F(2x2,3x3) the ResB convolutions (70 % of the "
                                     "forward) issue 2.25x fewer matrix instructions per output than the direct kernel did (busy 0.40 at "
                                     "1.0 ms per 2-view conv then).  pairs_per_s is the figure to compare",
                        "pairs_per_s": dc2, "algorithmic_f32_tflops": flop2 * dc2 / 1e12,
                        "note": "achieved = MFMA flops issued, priced at the dominant convolutions' rate: 3 products of fp16 pieces per float32 "
                                "product and 2.25x fewer multiplications as Winograd F(2x2,3x3) = 1.33 per algorithmic flop (the attention and "
                                "the 1x1 / first / last convolutions issue 3); on real data the 16-bit matrix pipe of this chip is power "
                                "limited near 1.1-1.3 PFLOP/s (DESIGN.md 4.4)"}
            del net, l1080, r1080
            # configs[3]: GMFlow matcher as DMSCT calls it (bidirectional + occlusion), random init, 540x960 -> 512x896
            from unimatch import GMFlow
            from methods.dmsct import DMSCT
            gm = GMFlow().to(device)
            a960, b960 = torch.rand(1, 3, 540, 960, device=device) * 255, torch.rand(1, 3, 540, 960, device=device) * 255
            size = DMSCT.derive_matcher_inference_size(a960.shape)
            tgm = forward_ms(lambda: gm(a960, b960, inference_size=size, pred_bidir_flow=True, fwd_bwd_consistency_check=True))
            gr = 1e3 / tgm["median_ms"]
            extra["gmflow_960x540_pairs_per_s_f32"] = gr
            extra["gmflow_960x540_forward"] = tgm
            # the metric is pairs/s: the module takes B pairs per forward (unimatch/__init__.py:60-67 does too), and at batch 1 most
            # launches of this network are 448 one-unit workgroups on 256 CUs (one 87 %-filled round); larger batches fill their rounds
            by_batch = {1: gr}
            for nb in (2, 4):
                an, bn = torch.rand(nb, 3, 540, 960, device=device) * 255, torch.rand(nb, 3, 540, 960, device=device) * 255
                tb = forward_ms(lambda: gm(an, bn, inference_size=size, pred_bidir_flow=True, fwd_bwd_consistency_check=True), n=6, warm=2)
                by_batch[nb] = nb * 1e3 / tb["median_ms"]
                del an, bn
            extra["gmflow_960x540_pairs_per_s_f32_by_batch"] = by_batch
            extra["gmflow_960x540_pairs_per_s_f32_best_batch"] = max(by_batch.values())
            # the reference's algorithmic FLOPs per pair; 0.376e12 of them (the SepConvGRU's loop-invariant input blocks)
            # are convolved once per forward instead of once per refinement iteration, so this is a throughput equivalent
            extra["gmflow_960x540_tflops_f32_equivalent"] = 3.58357106688e12 * gr / 1e12
            # the refinement correlation (6 launches per pair) is data dependent since round 4: a 4x8-pixel tile shares the box of
            # its windows in LDS when the flow is smooth across the tile (what an optical flow is); the random-init model above emits
            # noise of +-40 px, so every tile of the forward timed here takes the per-pixel form.  Both cases, on their own:
            tk = torch.randn(2, 128 * 224, 128, device=device)
            yy, xx = torch.meshgrid(torch.arange(128, dtype=torch.float32, device=device), torch.arange(224, dtype=torch.float32, device=device), indexing="ij")
            smooth = torch.stack([0.05 * xx - 0.02 * yy, 0.03 * yy + 0.01 * xx], 0)[None].repeat(2, 1, 1, 1).contiguous()
            noise = 40.0 * torch.randn(2, 2, 128, 224, device=device)
            extra["gmflow_local_corr_flow_us"] = {
                "smooth_flow_shared_box": forward_ms(lambda: ct_hip.local_corr_flow(tk, tk, smooth, 4))["median_ms"] * 1e3,
                "noise_flow_40px_per_pixel_form": forward_ms(lambda: ct_hip.local_corr_flow(tk, tk, noise, 4))["median_ms"] * 1e3,
                "note": "128x224 tokens, batch 2, radius 4; the forward above (random weights) runs the second case 6 times per pair"}
            del tk, smooth, noise
            # (no whole-forward number for the smooth case: a random-weight matcher emits noise flows for ANY pair -- translated
            # copies by 3 .. 32 px still give a flow std of 130 px, round 5 -- and scaling refine.* down does not change the flow that
            # ENTERS the refinement)
            gmp = read_stamped(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_gmflow_960x540_mfma_pmc.json"))
            extra["gmflow_960x540_mfma_busy_time_weighted"] = gmp.get("_all_kernels", {}).get("mfma_busy_frac_time_weighted") if gmp else None
            if gmp:
                extra["gmflow_960x540_mfma_busy_note"] = ("MFMA-busy is not comparable across arithmetic forms: the fp16 two-piece kernels issue "
                                                          "half the MFMA cycles per product of the bf16 three-piece ones they replaced "
                                                          "(33 -> 24 ms per pair at 0.245 -> 0.22 busy)")
            # configs[3] whole: DMSCT.forward (matcher + EfficientNet-B2 encoder on both views + fusion + U-Net decoder + head), random init
            dm = DMSCT().to(device).eval()
            dm.matcher = gm
            t960, r960 = a960 / 255, b960 / 255
            tdm = forward_ms(lambda: dm(t960, r960))
            extra["dmsct_960x540_pairs_per_s_f32"] = 1e3 / tdm["median_ms"]
            extra["dmsct_960x540_forward"] = tdm
            del dm

    # configs[4] with the uploads inside the measurement: every rank takes part (frame f -> rank f % world)
    n_video = args.video if args.video > 0 else (1000 if (world == 1 and not args.no_extra) else 0)
    if n_video > 0 and not on_cpu:
        del tgt, ref, gt, out
        torch.cuda.empty_cache()
        vid = {"u8": video_stream(n_video, rank, world)}
        if rank == 0:
            extra["video_stream"] = vid
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not on_cpu:
        cpu = cpu_baseline()

    if rank == 0:
        line = {
            "metric": BASELINE_METRIC,
            "value": value, "unit": "stereopairs/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": K, "warmup": Wm,
            "init_seconds": args.init_seconds, "init_steps": init_steps,      # un-timed steps BEFORE the W warm-up steps (code objects, clocks)
            "source_stamp": source_stamp(),
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[ct_hip.lab_mode()], "data": "synthetic",
            **({"dry_run": True, "dry_run_note": "CT_BENCH_DEVICE=cpu: gloo plumbing check with a stub step; `value` is meaningless",
                "gathered_rows": int(gathered.shape[0] * gathered.shape[1] * gathered.shape[2]) if gathered is not None else int(metrics.shape[0] * metrics.shape[1]),
                "gathered_checksum": float((gathered if gathered is not None else metrics).sum())} if on_cpu else {}),
            # the other half of BASELINE.json's metric: dcmcs3di forward at 1920x1080 on the same GPU (details: roofline_cnn)
            "value_cnn": roof_cnn["pairs_per_s"] if roof_cnn else None, "unit_cnn": "stereopairs/s (dcmcs3di fwd, 1920x1080)",
            "config": {"workload": "configs[1]: methods.linear.color_transfer_between_images (Reinhard) on "
                                   "1920x1080 synthetic float32 RGB pairs, HBM-resident; float32 arithmetic in both sweeps "
                                   "(difference forms, Lab within 5e-5 of the float64 reference; gate 1e-4); "
                                   "+ per-frame %s against resident ground truth" % (",".join(names) or "no metric"),
                       "pairs_per_step_per_gpu": B, "io_dtype": "float32", "height": H, "width": W,
                       "lab_arithmetic": ct_hip.lab_mode(), "metrics": names,
                       "sharding": "frame f -> rank f % world; one all_gather of the [frames, n_metrics] table"},
            "roofline": roof, "roofline_cnn": roof_cnn, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                              # rank 0 is still measuring its roofline while the others arrive here
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
