"""Float accumulation checkpoint / resume and the reference's autosave rule and file naming
(reference qt/mainwindow.cpp:93-99)."""
import ctypes as C
import os

import numpy as np
import pytest


def test_autosave_rule_and_name(pt):
    H = pt.host()
    # reference: (iter % every == 0 || iter == 1) && iter > 0
    due = [i for i in range(0, 21) if H.pth_autosave_due(i, 5)]
    assert due == [1, 5, 10, 15, 20]
    assert not H.pth_autosave_due(5, 0)
    buf = C.create_string_buffer(256)
    assert H.pth_autosave_name(b"images", b"teapot", 250, 7, 3, 2019, buf, 256) == 0
    assert buf.value == b"images/teapot_07-03-2019_250spp.png"      # <folder>/<name>_dd-MM-yyyy_<iter>spp.png
    assert H.pth_autosave_name(b"images", b"x", 1, 0, 0, 0, buf, 256) == 0 and buf.value.endswith(b"_1spp.png")
    assert H.pth_autosave_name(b"images", b"teapot", 250, 7, 3, 2019, buf, 8) != 0


def test_checkpoint_file_round_trip_and_rejects_garbage(pt, tmp_path):
    H = pt.host()
    rng = np.random.default_rng(0)
    a = rng.random((6 * 5, 4), dtype=np.float32)
    fp = a.ctypes.data_as(C.POINTER(C.c_float))
    path = str(tmp_path / "a.ptxacc").encode()
    assert H.pth_checkpoint_write_file(path, 6, 5, 37, 0xDEADBEEF12345678, fp) == 0
    w, h, n, seed = C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64()
    b = np.zeros_like(a)
    assert H.pth_checkpoint_read_file(path, C.byref(w), C.byref(h), C.byref(n), C.byref(seed), b.ctypes.data_as(C.POINTER(C.c_float)), b.size) == 0
    assert (w.value, h.value, n.value, seed.value) == (6, 5, 37, 0xDEADBEEF12345678)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not os.path.exists(path.decode() + ".tmp")
    # too small a buffer, truncated file, trailing bytes, wrong magic: all refused with a message
    assert H.pth_checkpoint_read_file(path, C.byref(w), C.byref(h), C.byref(n), C.byref(seed), b.ctypes.data_as(C.POINTER(C.c_float)), 8) != 0
    raw = open(path, "rb").read()
    for bad in (raw[:-4], raw + b"x", b"NOTACKPT" + raw[8:], raw[:20], b""):
        q = tmp_path / "bad.ptxacc"
        q.write_bytes(bad)
        assert H.pth_checkpoint_read_file(str(q).encode(), C.byref(w), C.byref(h), C.byref(n), C.byref(seed), None, 0) != 0
        assert H.pth_last_error()
    assert H.pth_checkpoint_read_file(str(tmp_path / "missing").encode(), C.byref(w), C.byref(h), C.byref(n), C.byref(seed), None, 0) != 0


@pytest.mark.gpu
def test_resume_is_bit_identical_to_an_uninterrupted_run(pt, ctx, tmp_path):
    import gpuops
    H, L = pt.host(), pt.lib()
    desc, cam = pt.load_scene("teapot", width=96, height=64)
    scene = H.pth_scene_device()
    H.pth_set_seed(21)
    assert H.pth_kernel_init() == 0
    whole, _ = gpuops.render(ctx, scene, cam, 0, 12, 21)
    part, _ = gpuops.render(ctx, scene, cam, 0, 5, 21)
    path = str(tmp_path / "teapot.ptxacc").encode()
    assert H.pth_checkpoint_save(path, 5) == 0, H.pth_last_error()
    pt.check(L.ptx_accum_clear(ctx))
    done = C.c_int32()
    assert H.pth_checkpoint_load(path, C.byref(done)) == 0, H.pth_last_error()
    assert done.value == 5
    resumed, _ = gpuops.render(ctx, scene, cam, 5, 7, 21, clear=False)
    assert np.array_equal(whole.view(np.uint32), resumed.view(np.uint32))
    # a checkpoint of another seed or resolution is refused
    H.pth_set_seed(22)
    assert H.pth_checkpoint_load(path, C.byref(done)) != 0 and b"seed" in H.pth_last_error()
    H.pth_set_seed(21)
    pt.load_scene("teapot", width=64, height=64)
    assert H.pth_kernel_init() == 0
    assert H.pth_checkpoint_load(path, C.byref(done)) != 0 and b"resolution" in H.pth_last_error()
    H.pth_set_seed(0)


@pytest.mark.gpu
def test_checkpoint_of_another_scene_bounce_limit_or_texture_switch_is_refused(pt, ctx, tmp_path):
    """The header records what the samples depend on besides seed and resolution (scene content hash, bounce limit,
    texture switch): a resume that would mix two different images in one buffer is refused with a message."""
    import gpuops
    H = pt.host()
    H.pth_set_seed(5)
    desc, cam = pt.load_scene("teapot", width=64, height=48)
    assert H.pth_kernel_init() == 0
    gpuops.render(ctx, H.pth_scene_device(), cam, 0, 2, 5)
    path = str(tmp_path / "a.ptxacc").encode()
    assert H.pth_checkpoint_save(path, 2) == 0, H.pth_last_error()
    done = C.c_int32()
    assert H.pth_checkpoint_load(path, C.byref(done)) == 0 and done.value == 2
    pt.load_scene("teapot", width=64, height=48, bounces=3)
    assert H.pth_kernel_init() == 0
    assert H.pth_checkpoint_load(path, C.byref(done)) != 0 and b"max_ray_bounces" in H.pth_last_error()
    pt.load_scene("teapot", width=64, height=48, textures=1)
    assert H.pth_kernel_init() == 0
    assert H.pth_checkpoint_load(path, C.byref(done)) != 0 and b"textures" in H.pth_last_error()
    pt.load_scene("geoms", width=64, height=48)
    assert H.pth_kernel_init() == 0
    assert H.pth_checkpoint_load(path, C.byref(done)) != 0 and b"different scene" in H.pth_last_error()
    # the same scene seen by a moved camera is another image: refused, and accepted again once the camera is back
    pt.load_scene("teapot", width=64, height=48)
    assert H.pth_kernel_init() == 0
    assert H.pth_checkpoint_load(path, C.byref(done)) == 0
    assert H.pth_camera_control(b"h_ang", 10.0) == 0
    assert H.pth_checkpoint_load(path, C.byref(done)) != 0 and b"camera" in H.pth_last_error()
    assert H.pth_camera_control(b"reset", 0.0) == 0
    assert H.pth_checkpoint_load(path, C.byref(done)) == 0
    H.pth_set_seed(0)
