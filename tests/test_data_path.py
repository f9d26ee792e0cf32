"""The test-time data path (reference utils/data.py:12-22,87-125,168-179): the 31 grid distortions on the GPU against the torch
restatement of torchvision's uint8 arithmetic (oracle/distort.py -- parity unpinned: torchvision is absent offline), the file
datasets on PNGs written here, the pinned double-buffered prefetcher, and CPU affinity slicing."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import distort as od      # noqa: E402


def test_grid_matches_the_reference_recipe():
    from utils.data import setup_grid_distortions
    specs = setup_grid_distortions()
    assert len(specs) == 31 and specs[0][0] == "identity" and specs == od.setup_grid_distortions()
    assert [s[0] for s in specs[1:6]] == ["brightness", "contrast", "saturation", "hue", "gamma"]
    assert specs[1][1] == 0.5 and specs[4][1] == -0.5 and abs(specs[-1][1] - 1.5) < 1e-12      # linspace(-0.5, 0.5, 6)


def test_oracle_distortions_basic_properties():
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (3, 20, 30), generator=g, dtype=torch.uint8)
    assert torch.equal(od.apply(img, "brightness", 1.0), img) and torch.equal(od.apply(img, "saturation", 1.0), img)
    assert torch.equal(od.apply(img, "gamma", 1.0), img)
    assert (od.apply(img, "hue", 0.0).int() - img.int()).abs().max() <= 1          # hsv round trip
    gray = od.apply(img, "saturation", 0.0)
    assert torch.equal(gray[0], gray[1]) and torch.equal(gray[1], gray[2])
    with pytest.raises(ValueError):
        od.adjust_hue(img, 0.7)


def test_cpu_affinity_slices(monkeypatch):
    from utils import sharding as sh
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity")
    calls = []
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: calls.append(list(cpus)))
    assert sh.pin_rank_to_cpus(0, 1) is None and not calls
    assert sh.pin_rank_to_cpus(2, 8) == [4, 5] and calls[-1] == [4, 5]
    assert sh.pin_rank_to_cpus(1, 2) == list(range(8, 16))


@pytest.fixture(scope="module")
def hip():
    import ct_hip
    ct_hip.lib()
    return ct_hip


@pytest.mark.gpu
def test_hip_distortions_vs_oracle(hip):
    g = torch.Generator().manual_seed(3)
    coarse = torch.randint(0, 256, (3, 12, 17), generator=g).float()
    img = torch.nn.functional.interpolate(coarse[None], size=(271, 483), mode="bilinear", align_corners=True)[0]
    img = (img + torch.randint(-20, 21, img.shape, generator=g)).clamp(0, 255).to(torch.uint8)
    img[:, :4, :4] = 0
    img[:, 4:8, :4] = 255
    img[:, 8:12, :4] = torch.tensor([255, 0, 0], dtype=torch.uint8).view(3, 1, 1)
    worst = {}
    for name, param in od.setup_grid_distortions():
        want = od.apply(img, name, param)
        got_f, got_u = hip.distort_u8(img.cuda(), name, param, want_u8=True)
        got_u = got_u.cpu()
        assert torch.equal(got_f.cpu(), got_u.float() / 255)
        d = (got_u.int() - want.int()).abs()
        worst[(name, round(float(param), 2))] = (int(d.max()), float((d > 0).float().mean()))
        if name in ("identity", "brightness", "saturation"):
            assert int(d.max()) == 0, (name, param)                                # plain float32 arithmetic: bit exact
        else:
            # contrast (float32 mean of 131 k values: torch's summation order), hue and gamma (powf / hsv round trip): the
            # truncating float -> uint8 cast turns a last-bit difference into one grey level on a few pixels
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-3, (name, param, worst[(name, round(float(param), 2))])
    print("\n[distortions] (max |diff| in grey levels, fraction of values differing):", {k: v for k, v in worst.items() if v[0]})
    with pytest.raises(ValueError):
        hip.distort_u8(img.cuda(), "hue", 0.7)


@pytest.mark.gpu
def test_file_datasets_and_prefetch(tmp_path, hip):
    from PIL import Image
    from utils.data import ArtificialTestDataset, DataModule, RealWorldTestDataset, prefetch
    rng = np.random.default_rng(0)
    (tmp_path / "Test").mkdir()
    (tmp_path / "Real-World Test" / "scene1").mkdir(parents=True)
    imgs = {}
    for name in ("a_L", "a_R", "b_L", "b_R"):
        imgs[name] = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
        Image.fromarray(imgs[name]).save(tmp_path / "Test" / (name + ".png"))
    for name in ("s_L", "s_LD", "s_R"):
        imgs[name] = rng.integers(0, 256, (33, 47, 3), dtype=np.uint8)
        Image.fromarray(imgs[name]).save(tmp_path / "Real-World Test" / "scene1" / (name + ".png"))
    ds = ArtificialTestDataset(tmp_path / "Test")
    assert len(ds) == 2 * 31
    s = ds[31 + 4]                                    # image b, 5th entry of the grid = hue at -0.5
    gt = torch.from_numpy(imgs["b_L"].transpose(2, 0, 1).copy())
    assert torch.equal(s["gt"].cpu(), gt.float() / 255) and torch.equal(s["reference"].cpu(), torch.from_numpy(imgs["b_R"].transpose(2, 0, 1).copy()).float() / 255)
    want = od.apply(gt, "hue", -0.5).float() / 255
    assert (s["target"].cpu() - want).abs().max() <= 1 / 255 + 1e-7
    rw = RealWorldTestDataset(tmp_path / "Real-World Test")
    assert len(rw) == 1 and torch.equal(rw[0]["target"].cpu(), torch.from_numpy(imgs["s_LD"].transpose(2, 0, 1).copy()).float() / 255)
    dm = DataModule(data_dir=str(tmp_path))
    assert len(dm.test_dataloader()) == 2 and len(dm.test_frames()) == 62
    # the prefetcher yields exactly what indexing yields, in order
    idx = [0, 7, 33, 61]
    for (i, smp), j in zip(prefetch(ds, idx, torch.device("cuda", 0)), idx):
        ref = ds[j]
        assert i == j and all(torch.equal(smp[k], ref[k]) for k in ("gt", "reference", "target"))


@pytest.mark.gpu
def test_cli_on_the_artificial_grid(capsys):
    from utils import cli
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "color-transfer_amd", "configs", "others.yaml")
    table = cli.main(["test", "--config", cfg, "--data.synthetic", "artificial", "--data.n_frames", "1", "--data.height", "96", "--data.width", "128"])
    assert table.shape == (31, 4) and torch.isfinite(table).all()        # PSNR, SSIM, FSIM, iCID
    assert float(table[0, 0]) > float(table[1, 0]) - 50       # sanity: the identity sample is not worse than everything else
    assert "Test SSIM" in capsys.readouterr().out


def test_synthetic_video_pool_cycles_for_every_stride():
    """the pinned chunk a group takes is a hash of its first frame -- it cycles through the pool whatever the stride
    between a rank's groups is (first_frame % pool handed every group of a rank the same chunk), chunks are made on first use, and the
    per-frame path returns slot 0 of the chunk the grouped path would hand over for a group starting there"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "color-transfer_amd"))
    from utils.data import SyntheticStereoVideoU8
    ds = SyntheticStereoVideoU8(n_frames=1000, height=4, width=6, group=8, pool=8)
    assert ds._chunks is None
    for world in (1, 2, 8):
        for rank in range(world):
            firsts = [rank + world * 8 * c for c in range(32)]            # first frames of a rank's groups (frames rank, rank + world, ...)
            used = {ds._pool_index(f) for f in firsts}
            assert len(used) >= 6, (world, rank, used)
    c = ds.host_chunk(64)
    assert c.shape == (3, 8, 4, 6, 3) and c.dtype == torch.uint8
    assert sum(x is not None for x in ds._chunks) == 1                    # only what was touched
    ds.prepare(range(0, 1000))                                            # a timed caller makes every chunk of its groups first
    assert all(x is not None for x in ds._chunks)
    f = ds[64]
    assert torch.equal(f["target"], c[0, 0].permute(2, 0, 1).float() / 255) and torch.equal(f["gt"], c[2, 0].permute(2, 0, 1).float() / 255)
