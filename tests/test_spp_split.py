"""The N>1 path on CPU: 2 gloo ranks split the sample range, one all_reduce, checked against a
single-rank render of the same samples. The renderer here is the CPU oracle (tests only)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sample_range_is_a_partition():
    sp = importlib.import_module("path-tracer_amd.spp_split")
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 7, 8, 1000):
            got = []
            for r in range(world):
                f, c = sp.sample_range(r, world, total, first_sample=5)
                got += list(range(f, f + c))
            assert got == list(range(5, 5 + total))
    assert sp.sample_range(7, 8, 1000) == (875, 125)
    with pytest.raises(ValueError):
        sp.sample_range(2, 2, 10)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pt = importlib.import_module("path-tracer_amd")
    sp = importlib.import_module("path-tracer_amd.spp_split")
    import orc
    desc, cam = pt.load_scene("geoms", width=40, height=30, upload=False)
    accum = torch.zeros((cam.width * cam.height, 4), dtype=torch.float32)

    def render_fn(first, count):
        a = accum.numpy()
        orc.render(desc, cam, first, count, 77, accum=a, nthreads=2)

    synced = []
    mine = sp.render_split(render_fn, accum, total_samples=6, sync_fn=lambda: synced.append(1))
    assert synced == [1]                       # the sync hook runs between the render and the collective
    partial = torch.zeros_like(accum)
    orc.render(desc, cam, mine[0], mine[1], 77, accum=partial.numpy(), nthreads=2)
    np.save(os.path.join(out_dir, f"partial{rank}.npy"), partial.numpy())
    np.save(os.path.join(out_dir, f"sum{rank}.npy"), accum.numpy())
    # progressive: rounds of 4 global samples, a reduced copy after each; the local buffer keeps only this rank's samples
    accum2 = torch.zeros_like(accum)
    frames = []

    def render_fn2(first, count):
        orc.render(desc, cam, first, count, 77, accum=accum2.numpy(), nthreads=2)

    last = sp.render_progressive(render_fn2, accum2, total_samples=6, every=4,
                                 on_image=lambda img, done: frames.append((done, img.clone())))
    assert [d for d, _ in frames] == [4, 6] and torch.equal(frames[-1][1], last)
    np.save(os.path.join(out_dir, f"prog4_{rank}.npy"), frames[0][1].numpy())
    np.save(os.path.join(out_dir, f"prog6_{rank}.npy"), last.numpy())
    np.save(os.path.join(out_dir, f"proglocal{rank}.npy"), accum2.numpy())
    dist.destroy_process_group()


def test_two_rank_split_matches_single_rank(tmp_path, pt, orc):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "sum0.npy"), np.load(tmp_path / "sum1.npy")
    p0, p1 = np.load(tmp_path / "partial0.npy"), np.load(tmp_path / "partial1.npy")
    assert np.array_equal(s0, s1)               # every rank holds the reduced image
    assert np.array_equal(s0, p0 + p1)          # exactly the sum of the two partials
    desc, cam = pt.load_scene("geoms", width=40, height=30, upload=False)
    full, _ = orc.render(desc, cam, 0, 6, 77)
    # same sample set; only the float summation order differs ((s0+s1+s2)+(s3+s4+s5) vs sequential)
    assert np.allclose(s0, full, rtol=1e-5, atol=1e-6)
    # progressive rounds: after round 1 every rank holds samples [0, 4), after round 2 all six
    first4, _ = orc.render(desc, cam, 0, 4, 77)
    for r in (0, 1):
        assert np.allclose(np.load(tmp_path / f"prog4_{r}.npy"), first4, rtol=1e-5, atol=1e-6)
        assert np.allclose(np.load(tmp_path / f"prog6_{r}.npy"), full, rtol=1e-5, atol=1e-6)
    # the local buffers stayed partial: they sum to the final image exactly
    assert np.array_equal(np.load(tmp_path / "proglocal0.npy") + np.load(tmp_path / "proglocal1.npy"), np.load(tmp_path / "prog6_0.npy"))
