"""The N > 1 render path with the HIP renderer (the collective must not
overtake the render kernels). Two fresh rank processes share GPU 0 and reduce over gloo - what a
one-GPU box can exercise of `render.py --gpus N`; the RCCL run itself is the driver's (bench.py)."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch   # before the first HIP call of libptx_hip.so: torch brings its own HIP runtime, and initialised second it finds no GPU

import gpuops

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    return float((np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max())


def test_render_split_with_the_hip_renderer_orders_render_before_collective(pt, ctx):
    """Single process, world size 1: render_split / render_progressive drive the REAL asynchronous
    ptx_render_samples into a torch-owned buffer and call ptx_render_sync before they touch it. The
    result equals an unsplit render bit for bit (one rank: same samples, same order)."""
    import torch
    sp = importlib.import_module("path-tracer_amd.spp_split")
    L = pt.lib()
    desc, cam = pt.load_scene("teapot", width=320, height=180)
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    want, _ = gpuops.render(ctx, scene, cam, 0, 24, 3)

    accum = torch.zeros((cam.width * cam.height, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    gpuops.render_init(ctx, cam, C.c_void_p(accum.data_ptr()))
    synced = []

    def render_fn(first, count):
        pt.check(L.ptx_render_samples(ctx, scene, first, count, 3, 0))

    def sync_fn():
        pt.check(L.ptx_render_sync(ctx))
        synced.append(1)

    assert sp.render_split(render_fn, accum, 24, sync_fn=sync_fn) == (0, 24)
    assert synced == [1]
    assert np.array_equal(accum.cpu().numpy().view(np.uint32), want.view(np.uint32))

    accum.zero_()
    torch.cuda.synchronize()
    frames = []
    last = sp.render_progressive(render_fn, accum, 24, every=10, sync_fn=sync_fn,
                                 on_image=lambda img, done: frames.append((done, img.cpu().numpy().copy())))
    assert [d for d, _ in frames] == [10, 20, 24]
    gpuops.render_init(ctx, cam)
    first10, _ = gpuops.render(ctx, scene, cam, 0, 10, 3)
    assert np.array_equal(frames[0][1].view(np.uint32), first10.view(np.uint32))
    assert np.array_equal(last.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("every", [0, 6])
def test_two_rank_hip_render_matches_single_rank(pt, ctx, tmp_path, every):
    """render.py --gpus 2, started as a child process: its torch.distributed.run starts two fresh rank
    processes; both render their sample ranges with the HIP kernels on GPU 0, reduce over gloo, and rank 0
    writes the buffer and the PNG. Against this process's single-rank HIP render of the same 16
    samples: <= 1e-5 relative (float summation order only)."""
    acc_path, png_path = tmp_path / "acc.npy", tmp_path / "img.png"
    cmd = [sys.executable, os.path.join(ROOT, "path-tracer_amd", "render.py"), "--scene", "teapot", "--width", "256",
           "--height", "144", "--spp", "16", "--seed", "9", "--gpus", "2", "--backend", "gloo", "--all-on-device", "0",
           "--every", str(every), "--out-accum", str(acc_path), "--out", str(png_path)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
    got = np.load(acc_path)
    desc, cam = pt.load_scene("teapot", width=256, height=144)
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    want, _ = gpuops.render(ctx, scene, cam, 0, 16, 9)
    assert got.shape == want.shape and want[:, :3].max() > 0
    assert _rel(got, want) <= 1e-5, _rel(got, want)
    # the split really happened: the reduced buffer is the sum of the two ranks' ranges
    if every == 0:
        a, _ = gpuops.render(ctx, scene, cam, 0, 8, 9)
        b, _ = gpuops.render(ctx, scene, cam, 8, 8, 9)
        assert np.array_equal(got.view(np.uint32), (a + b).view(np.uint32))
    # the PNG is the tone-mapped, mirrored image of that buffer
    w, h, ch = C.c_int(), C.c_int(), C.c_int()
    px = C.POINTER(C.c_float)()
    H = pt.host()
    assert H.pth_load_image_float(str(png_path).encode(), C.byref(w), C.byref(h), C.byref(ch), C.byref(px)) == 0
    assert (w.value, h.value) == (256, 144)
    H.pth_free(px)
