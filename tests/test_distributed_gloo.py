"""world_size = 2 on CPU (gloo): the frame sharding + single-collective metric gather used by bench.py
and utils.cli reassembles the per-frame table in frame order, identically to a 1-rank run."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "color-transfer_amd"))
    from utils import sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = sh.frames_of_rank(n_frames, rank, world)
    # a deterministic per-frame "metric" that depends on the frame seed only (never on the rank)
    local = torch.tensor([[f, sh.frame_seed(f) * 0.5, float(f) ** 2] for f in frames], dtype=torch.float64).reshape(-1, 3)
    table = sh.gather_frame_metrics(local, n_frames)
    torch.save(table, os.path.join(out_dir, "table_%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_rank(tmp_path):
    n_frames, world = 7, 2          # odd on purpose: ranks own 4 and 3 frames, padded rows are dropped
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "color-transfer_amd"))
    from utils import sharding as sh
    want = torch.tensor([[f, sh.frame_seed(f) * 0.5, float(f) ** 2] for f in range(n_frames)], dtype=torch.float64)
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), "table_%d.pt" % r))
        assert got.shape == (n_frames, 3)
        assert torch.equal(got, want)


def _worker_status(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "color-transfer_amd"))
    from utils import sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = torch.zeros((len(sh.frames_of_rank(5, rank, world)), 2), dtype=torch.float64)
    try:
        sh.gather_frame_metrics(local, 5, status=1 if rank == 1 else 0)
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, "status_%d.txt" % rank), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_a_status_on_one_rank_raises_on_every_rank(tmp_path):
    """the device status travels through the gather itself -- a rank that raised BEFORE the collective would leave the
    others blocked in it until the backend's timeout; here rank 1 reports a status and both ranks raise, after the same collective"""
    port = _free_port()
    mp.spawn(_worker_status, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg = open(os.path.join(str(tmp_path), "status_%d.txt" % r)).read()
        assert "status on rank(s) 1: 0x1" in msg, (r, msg)
