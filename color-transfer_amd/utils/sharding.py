"""Frame sharding of a stereo video across the GPUs of one node.

Frames are independent units (the reference's Runner loops samples independently,
methods/__init__.py:20-25): frame f belongs to rank f % world.  Each rank keeps a
[n_local, n_metrics] tensor of per-frame metrics; ONE collective at the end
(all_gather_into_tensor; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)
reassembles the [n_frames, n_metrics] table in frame order on every rank.  Seeds are derived from
the FRAME index, never from the rank, so results are identical for any world size.
"""
import torch
import torch.distributed as dist


def pin_rank_to_cpus(local_rank, local_world):
    """One process per GPU: give rank r the r-th contiguous slice of the CPUs this process may run on (NUMA nodes are
    contiguous CPU ranges and GPUs 0..3 / 4..7 hang off socket 0 / 1 on the 8-GPU MI355X nodes), so that the launch threads
    of different ranks do not migrate across sockets or share cores.  Call BEFORE the first GPU call (the HIP runtime's
    helper threads inherit the mask).  Returns the CPU list (or None when the platform has no sched_setaffinity)."""
    import os
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // local_world
    if per < 1:
        return None
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    return mine


def frames_of_rank(n_frames, rank, world):
    """Frame indices owned by `rank`: rank, rank + world, ..."""
    return list(range(rank, n_frames, world))


def padded_local_count(n_frames, world):
    """Every rank contributes the same number of rows to the gather (pad with NaN rows)."""
    return (n_frames + world - 1) // world


def frame_seed(frame_index, base=1234):
    """Seed of the synthetic frame / of the IDT rotations of frame `frame_index` (world-size independent)."""
    return base + int(frame_index)


def gather_frame_metrics(local_metrics, n_frames, rank=None, world=None, status=None):
    """local_metrics: [n_local, n_metrics] for frames_of_rank(...) in that order.
    Returns [n_frames, n_metrics] in frame order on every rank (a single all_gather).
    status: this rank's device status word (default: read from the device when the table lives there; tests pass it in)."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if status is not None:
        status = int(status)
    elif not local_metrics.is_cuda:
        status = 0
    else:
        # the one place a run that never synchronises per frame looks at the device's sticky status (include/ct_hip.h:
        # ct_device_status): a persistent launch / stream-K convolution that gave up a bounded spin left NaN / wrong frames
        import ct_hip
        status = int(ct_hip.device_status(clear=True))
    n_pad = padded_local_count(n_frames, world)
    n_metrics = local_metrics.shape[1]
    # one extra row carries this rank's status through the SAME collective: a rank that raised before the gather would leave the
    # others blocked in it until the backend's timeout; this way every rank sees every status and all raise together
    buf = torch.full((n_pad + 1, n_metrics), float("nan"), dtype=local_metrics.dtype, device=local_metrics.device)
    buf[: local_metrics.shape[0]] = local_metrics
    buf[n_pad] = float(status)
    if world == 1:
        if status:
            raise RuntimeError("rank %d: the HIP kernels reported status 0x%x (a launch gave up a bounded spin: results invalid)" % (rank, status))
        return buf[:n_frames]
    out = torch.empty((world, n_pad + 1, n_metrics), dtype=local_metrics.dtype, device=local_metrics.device)
    dist.all_gather_into_tensor(out.view(world * (n_pad + 1), n_metrics), buf)
    bad = [(r, int(v)) for r, v in enumerate(out[:, n_pad, 0].tolist()) if v == v and int(v) != 0]
    if bad:
        raise RuntimeError("rank %d: the HIP kernels reported a status on rank(s) %s (a launch gave up a bounded spin: results invalid)"
                           % (rank, ", ".join("%d: 0x%x" % b for b in bad)))
    out = out[:, :n_pad]
    # rank r, slot i  ->  frame r + i * world
    table = out.permute(1, 0, 2).reshape(world * n_pad, n_metrics)
    return table[:n_frames]
