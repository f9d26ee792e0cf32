"""`python -m utils.cli test --config <yaml> [--model.func_spec ...] [--data.n_frames N] [--ckpt_path P]`

A minimal look-alike of the reference's LightningCLI entry point (utils/cli.py:1-3, README.md:69-71) for the
`test` sub-command only (Lightning/jsonargparse are not part of this stack): YAML with `class_path/init_args`
for model and data, dotted `--section.key value` overrides, `trainer.*` keys accepted and ignored.  Frames are
sharded over ranks when launched with torch.distributed.run (frame f -> rank f % world) and the per-frame
metrics (PSNR, SSIM, FSIM, iCID: the reference's Test PSNR / Test SSIM / Test FSIM / Test iCID) are gathered with ONE collective
(utils/sharding.py); rank 0 prints their means.
"""
import importlib
import os
import sys

import torch
import yaml


def _set(cfg, dotted, value):
    """`--model.func_spec X`, `--model.init_args.func_spec X` (LightningCLI's canonical form) and `--model.class_path X`"""
    keys = dotted.split(".")
    node, in_init = cfg, False
    for k in keys[:-1]:
        if in_init and k == "init_args":
            continue                                  # already inside init_args: the explicit component is redundant
        node = node.setdefault(k, {})
        if k in ("model", "data") and keys[-1] != "class_path":
            node = node.setdefault("init_args", {})
            in_init = True
    node[keys[-1]] = yaml.safe_load(value)


def _instantiate(section):
    module, cls = section["class_path"].rsplit(".", 1)
    return getattr(importlib.import_module(module), cls)(**section.get("init_args", {}))


def main(argv=None, timing=None):
    """timing: None, or a dict that receives {"seconds", "frames", "frames_local", "h2d_bytes"} of the first loader's loop --
    barrier + synchronize on both sides, the gather inside, an untimed first pass over a few groups before it (code objects,
    clocks, the communicator) -- for bench.py's configs[4] leg, which measures THIS entry point rather than a loop of its own."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] != "test":
        raise SystemExit("only the `test` sub-command exists here (fit/validate/predict are Lightning training paths)")
    cfg, ckpt, i = {}, None, 1
    if (len(argv) - 1) % 2:
        raise SystemExit("arguments come in `--key value` pairs; got a dangling %r" % argv[-1])
    while i < len(argv):
        key, val = argv[i], argv[i + 1]
        if key == "--config":
            with open(val) as fh:
                cfg = yaml.safe_load(fh) or {}
        elif key == "--ckpt_path":
            ckpt = val
        elif key.startswith("--"):
            _set(cfg, key[2:], val)
        i += 2
    import torch.distributed as dist
    from utils import sharding as sh
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    sh.pin_rank_to_cpus(int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # before any GPU call
    # CT_CLI_DEVICE=cpu: the host logic (argv / YAML handling, sharding, the one gather, the printed means) on CPU tensors with
    # the gloo backend -- for tests of a multi-rank run without GPUs.  It is no compute fallback: methods.* still need the
    # HIP library and raise without it; only a model whose test_step works on CPU tensors (a test stub) runs this way.
    on_cpu = os.environ.get("CT_CLI_DEVICE", "cuda") == "cpu"
    device = torch.device("cpu") if on_cpu else torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    if not on_cpu:
        torch.cuda.set_device(device)
    own_group = world > 1 and not dist.is_initialized()          # a caller (bench.py) may have built the communicator already
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    if world > 1 and dist.get_world_size() != world:
        raise SystemExit("WORLD_SIZE=%d but the process group has %d ranks" % (world, dist.get_world_size()))
    model = _instantiate(cfg["model"]).to(device).eval()
    if ckpt:
        # Lightning checkpoints carry hyper-parameters / optimizer state next to "state_dict"; weights_only=True refuses the
        # ones that pickle arbitrary objects.  Falling back to the unsafe loader executes whatever the file pickles, so it
        # happens only for that one error class and never silently; CT_TRUST_CKPT=0 forbids it.
        import pickle
        try:
            state = torch.load(ckpt, map_location=device, weights_only=True)
        except pickle.UnpicklingError as e:
            if os.environ.get("CT_TRUST_CKPT", "1") == "0":
                raise
            print("warning: %s is not loadable with weights_only=True (%s); loading it as a TRUSTED pickle "
                  "(set CT_TRUST_CKPT=0 to refuse)" % (ckpt, str(e).splitlines()[0][:120]), file=sys.stderr)
            state = torch.load(ckpt, map_location=device, weights_only=False)
        model.load_state_dict(state["state_dict"] if "state_dict" in state else state, strict=True)
    data_cfg = dict(cfg.get("data", {}))
    data_cfg["class_path"] = "utils.data.DataModule"
    dm = _instantiate(data_cfg)
    from methods import METRICS, fsim, icid, psnr, ssim
    from utils.data import prefetch, prefetch_groups
    # the reference's test_dataloader() returns [artificial, real-world] (utils/data.py:168-179) and Lightning logs each
    # metric once per loader ("Test PSNR/dataloader_idx_1"); a single loader prints the bare names like Lightning does
    loaders = dm.test_dataloader()
    tables = []

    def fence():
        if world > 1:
            dist.barrier()
        if not on_cpu:
            torch.cuda.synchronize()

    for li, frames in enumerate(loaders):
        mine = sh.frames_of_rank(len(frames), rank, world)
        grouped = (not on_cpu and hasattr(frames, "host_chunk") and hasattr(model, "test_group") and model.takes_groups())
        if grouped:
            # uint8 frames in pinned groups (configs[4]): one upload and ONE fused call per group -- transfer + clamp + PSNR of
            # Runner.test_step (methods/__init__.py:29-32) -- no torch kernel in the loop, one gather at the end
            rec = torch.zeros((max(len(mine), 1), 2), dtype=torch.float64, device=device)

            def run(indices):
                n = 0
                for ids, dev in prefetch_groups(frames, indices, device):
                    model.test_group(dev, rec[n:n + len(ids)])
                    n += len(ids)

            if timing is not None and li == 0:
                if hasattr(frames, "prepare"):
                    frames.prepare(mine)                        # the synthetic frames themselves: made before the clock starts
                run(mine[:3 * frames.group])                    # initialisation, not part of the measurement
                fence()
                import time
                t0 = time.perf_counter()
            run(mine)
            local = torch.full((len(mine), len(METRICS)), float("nan"), dtype=torch.float64, device=device)
            local[:, 0] = rec[:len(mine), 1]
        else:
            if timing is not None and li == 0:
                fence()
                import time
                t0 = time.perf_counter()
            rows = []
            for f, sample in prefetch(frames, mine, device):       # pinned double-buffered uploads on a second stream
                batch = {k: v.unsqueeze(0) for k, v in sample.items()}
                if hasattr(model, "test_step"):
                    m = model.test_step(batch, f)
                    nan = torch.full((), float("nan"), device=device)
                    rows.append(torch.stack([m[k].reshape(()).to(device) if k in m else nan for k in METRICS]))
                else:                                   # CNN modules: forward(target, reference, inference=True)
                    corrected, _ = model(batch["target"], batch["reference"], inference=True)
                    corrected = corrected.clamp(0, 1)
                    rows.append(torch.stack([fn(corrected, batch["gt"]).reshape(()) for fn in (psnr, ssim, fsim, icid)]))
            local = torch.stack(rows).double() if rows else torch.zeros((0, len(METRICS)), dtype=torch.float64, device=device)
        table = sh.gather_frame_metrics(local, len(frames), rank, world)        # [n_frames, 4]: PSNR, SSIM, FSIM, iCID per frame
        if timing is not None and li == 0:
            fence()
            dt = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([dt], dtype=torch.float64, device=device)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            per_frame = 3 * frames.height * frames.width * 3 if grouped else 0
            timing.update({"seconds": dt, "frames": len(frames), "frames_local": len(mine), "h2d_bytes": per_frame * len(mine),
                           "grouped": grouped, "frames_per_call": frames.group if grouped else 1})
        tables.append(table)
        if rank == 0:
            suffix = "/dataloader_idx_%d" % li if len(loaders) > 1 else ""
            cols = [(i, name) for i, name in enumerate(METRICS) if not bool(torch.isnan(table[:, i]).all())]
            for j, (i, name) in enumerate(cols):
                print("%s%s: %.4f" % (name, suffix, float(table[:, i].mean())), end="   " if j + 1 < len(cols) else "")
            print("  (%d frames, %d GPU%s)" % (len(frames), world, "" if world == 1 else "s"))
    table = tables[0]
    if own_group:
        dist.destroy_process_group()
    return table


if __name__ == "__main__":
    main()
