"""The one-sample-per-call loop of the reference (runCUDA: src/kernel.cu:59-103, called by qt/cudarunthread.cpp:21-41) through
ptx_render_sample_ahead: every call must leave the accumulation buffer - and the tone-mapped "PBO" - bit for bit as one pass
per call leaves it, whatever window of samples the library traced ahead, and a camera change, a batch render or a knob in
between must end the window."""
import ctypes as C

import numpy as np
import pytest

import gpuops

pytestmark = pytest.mark.gpu


def _accum(pt, ctx, cam):
    acc = np.empty((cam.width * cam.height, 4), np.float32)
    pt.check(pt.lib().ptx_render_sync(ctx))
    pt.check(pt.lib().ptx_accum_download(ctx, acc.ctypes.data_as(C.POINTER(C.c_float))))
    return acc


@pytest.fixture
def look_ahead(pt):
    H = pt.host()
    before = H.pth_look_ahead()
    yield H.pth_set_look_ahead
    H.pth_set_look_ahead(before)


@pytest.mark.parametrize("scene_name", ["teapot", "geoms"])
def test_loop_with_look_ahead_equals_one_pass_per_call_and_the_oracle(pt, orc, ctx, look_ahead, scene_name):
    """40 iterations of runCUDA with windows of 1, 2, 4, 8, 8, 8, 8 (+1) samples traced ahead against the same loop with one
    pass per call: the tone-mapped image after EVERY iteration and the float buffer at the end are identical; the buffer
    equals one ptx_render_samples(0, 40) and the oracle's 40 samples. teapot has a multi-node mesh (traversal queues, retry
    queue); geoms stays inside the segment launch of bounce 0."""
    H, L = pt.host(), pt.lib()
    desc, cam = pt.load_scene(scene_name, width=96, height=64)
    scene = H.pth_scene_device()
    n_pix, iters = cam.width * cam.height, 40
    pbo = np.zeros((n_pix, 4), np.uint8)
    pp = pbo.ctypes.data_as(C.POINTER(C.c_ubyte))

    def loop(ahead):
        look_ahead(ahead)
        assert H.pth_look_ahead() == ahead
        assert H.pth_kernel_init() == 0
        frames = []
        for it in range(1, iters + 1):
            assert H.pth_run_iteration(it, pp) == it
            frames.append(pbo.copy())
        return frames, _accum(pt, ctx, cam)

    f1, a1 = loop(1)
    f8, a8 = loop(8)
    for it, (x, y) in enumerate(zip(f1, f8), 1):
        assert np.array_equal(x, y), f"tone-mapped image differs after iteration {it}"
    assert np.array_equal(a1.view(np.uint32), a8.view(np.uint32))
    f32, a32 = loop(32)
    assert np.array_equal(a1.view(np.uint32), a32.view(np.uint32)) and np.array_equal(f1[-1], f32[-1])
    batch, _ = gpuops.render(ctx, scene, cam, 0, iters, 0)
    assert np.array_equal(a1.view(np.uint32), batch.view(np.uint32))
    ref, _ = orc.render(desc, cam, 0, iters, 0)
    assert np.array_equal(a1.view(np.uint32), ref.view(np.uint32))


def test_camera_change_inside_a_window_restarts_like_the_reference(pt, orc, ctx, look_ahead):
    """Six iterations leave a window of four samples half used; the camera moves (SceneState setters, as the GUI does) and
    the loop restarts at iteration 1: nothing traced for the old camera may reach the buffer."""
    H = pt.host()
    look_ahead(16)
    desc, cam0 = pt.load_scene("teapot", width=64, height=48)
    assert H.pth_kernel_init() == 0
    pbo = np.zeros((64 * 48, 4), np.uint8)
    pp = pbo.ctypes.data_as(C.POINTER(C.c_ubyte))
    for it in range(1, 6):            # windows [0], [1, 2], [3 .. 6]: samples 5 and 6 stay parked
        assert H.pth_run_iteration(it, pp) == it
    H.pth_camera_control(b"h_ang", -20.0)
    H.pth_camera_control(b"pos_y", float(np.float32(cam0.position[1]) + np.float32(0.5)))
    cam1 = pt.PtxCamera()
    assert H.pth_camera_get(C.byref(cam1)) == 0
    assert H.pth_run_iteration(6, pp) == 1
    for it in (2, 3, 4, 5):
        assert H.pth_run_iteration(it, pp) == it
    ref, _ = orc.render(desc, cam1, 0, 5, 0)
    assert np.array_equal(_accum(pt, ctx, cam1).view(np.uint32), ref.view(np.uint32))


def test_sample_ahead_through_the_c_abi_any_order_of_calls(pt, ctx):
    """ptx_render_sample_ahead adds exactly the sample asked for, whatever the calls around it: a jump (window of one again),
    a batch render in between (its pass reuses the parked planes' memory), a knob, another seed, a sample asked for twice.
    The same sequence through ptx_render_samples(sample, 1) gives the same bits (float addition in the same order)."""
    H, L = pt.host(), pt.lib()
    desc, cam = pt.load_scene("teapot", width=80, height=56)
    scene = H.pth_scene_device()
    gpuops.render_init(ctx, cam)
    # (kind, first sample, count, seed)
    seq = [("one", s, 1, 0) for s in (0, 1, 2, 3, 4, 5, 6)]            # windows [0] [1,2] [3..6]
    seq += [("one", 10, 1, 0), ("one", 11, 1, 0), ("one", 12, 1, 0)]    # jump: [10], then [11, 12]
    seq += [("many", 13, 5, 0)]                                          # a batch pass drops the window ...
    seq += [("one", s, 1, 0) for s in (18, 19, 20, 21)]                   # ... and the ramp starts again: [18] [19,20] [21..24]
    seq += [("tune", 0, 0, 0)]                                           # a knob ends the window [21..24]
    seq += [("one", 22, 1, 0), ("one", 23, 1, 0), ("one", 23, 1, 0)]     # [22] [23, 24], 23 again from the window
    seq += [("one", 24, 1, 7), ("one", 25, 1, 7), ("one", 26, 1, 7)]     # another seed is another window
    seq += [("one", 3, 1, 0)]                                            # backwards

    def run(ahead):
        pt.check(L.ptx_accum_clear(ctx))
        for kind, first, count, seed in seq:
            if kind == "tune":
                pt.check(L.ptx_ctx_tune(ctx, b"wide_scap", -1))
            elif kind == "many" or ahead <= 0:
                pt.check(L.ptx_render_samples(ctx, scene, first, count, seed, 0))
            else:
                pt.check(L.ptx_render_sample_ahead(ctx, scene, first, seed, ahead))
        return _accum(pt, ctx, cam)

    plain = run(0)
    for ahead in (1, 4, 64):
        assert np.array_equal(plain.view(np.uint32), run(ahead).view(np.uint32)), ahead
    # refusals
    assert L.ptx_render_sample_ahead(ctx, scene, -1, 0, 8) != 0
    assert L.ptx_render_sample_ahead(ctx, None, 0, 0, 8) != 0


def test_sample_ahead_in_megakernel_and_debug_modes_renders_its_own_sample(pt, orc, ctx):
    """The AABB debug view and the forced megakernel have no parked radiance: the call falls back to one pass per sample."""
    H, L = pt.host(), pt.lib()
    desc, cam = pt.load_scene("teapot", width=64, height=48, aabb=1)
    scene = H.pth_scene_device()
    gpuops.render_init(ctx, cam)
    pt.check(L.ptx_accum_clear(ctx))
    for s in range(5):
        pt.check(L.ptx_render_sample_ahead(ctx, scene, s, 0, 16))
    ref, _ = orc.render(desc, cam, 0, 5, 0)
    assert np.array_equal(_accum(pt, ctx, cam).view(np.uint32), ref.view(np.uint32))


def test_page_locked_display_buffer_receives_the_same_image(pt, ctx):
    """ptx_host_alloc (what kernelAllocPBO hands the render loop in place of the reference's mapped pixel buffer): the
    tone-mapped image written into it equals the one written into ordinary memory; freeing NULL is allowed."""
    H, L = pt.host(), pt.lib()
    desc, cam = pt.load_scene("geoms", width=64, height=48)
    scene = H.pth_scene_device()
    gpuops.render_init(ctx, cam)
    gpuops.render(ctx, scene, cam, 0, 6, 0)
    n = cam.width * cam.height
    plain = np.zeros((n, 4), np.uint8)
    pt.check(L.ptx_tonemap(ctx, 6, C.c_float(cam.gamma_corr), plain.ctypes.data_as(C.POINTER(C.c_ubyte))))
    raw = C.c_void_p()
    pt.check(L.ptx_host_alloc(ctx, n * 4, C.byref(raw)))
    assert raw.value
    try:
        pt.check(L.ptx_tonemap(ctx, 6, C.c_float(cam.gamma_corr), C.cast(raw, C.POINTER(C.c_ubyte))))
        locked = np.ctypeslib.as_array(C.cast(raw, C.POINTER(C.c_ubyte)), shape=(n, 4)).copy()
    finally:
        pt.check(L.ptx_host_free(ctx, raw))
    assert plain.any() and np.array_equal(plain, locked)
    pt.check(L.ptx_host_free(ctx, None))
    assert L.ptx_host_alloc(ctx, 0, C.byref(raw)) != 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_call_sequences_equal_one_pass_per_call(pt, ctx, seed):
    """Seeded random sequences of what a caller can do between two iterations - the next sample, a jump, a repeat, a batch
    render, a knob, a camera change back and forth, a cleared buffer, another seed - through ptx_render_sample_ahead with
    windows up to 8, against the same sequence with one pass per sample."""
    H, L = pt.host(), pt.lib()
    desc, cam_a = pt.load_scene("teapot", width=72, height=40)
    scene = H.pth_scene_device()
    cam_b = pt.PtxCamera()
    C.memmove(C.byref(cam_b), C.byref(cam_a), C.sizeof(cam_a))
    cam_b.position[0] = cam_a.position[0] + 0.5
    rng = np.random.default_rng(seed)
    ops, s = [], 0
    for _ in range(70):
        r = rng.random()
        if r < 0.62:
            ops.append(("one", s, 0)); s += 1
        elif r < 0.70:
            s = int(rng.integers(0, 200)); ops.append(("one", s, 0)); s += 1
        elif r < 0.76:
            ops.append(("one", max(0, s - 1), 0))
        elif r < 0.82:
            n = int(rng.integers(1, 6)); ops.append(("many", s, n)); s += n
        elif r < 0.87:
            ops.append(("tune", 0, 0))
        elif r < 0.93:
            ops.append(("camera", int(rng.integers(0, 2)), 0))
        elif r < 0.96:
            ops.append(("clear", 0, 0))
        else:
            ops.append(("one", s, 5)); s += 1

    def run(ahead):
        gpuops.render_init(ctx, cam_a)
        pt.check(L.ptx_accum_clear(ctx))
        for kind, a, b in ops:
            if kind == "one" and ahead > 0:
                pt.check(L.ptx_render_sample_ahead(ctx, scene, a, b, ahead))
            elif kind == "one":
                pt.check(L.ptx_render_samples(ctx, scene, a, 1, b, 0))
            elif kind == "many":
                pt.check(L.ptx_render_samples(ctx, scene, a, b, 0, 0))
            elif kind == "tune":
                pt.check(L.ptx_ctx_tune(ctx, b"wf_batch", -1))
            elif kind == "camera":
                pt.check(L.ptx_render_set_camera(ctx, C.byref(cam_b if a else cam_a)))
            elif kind == "clear":
                pt.check(L.ptx_accum_clear(ctx))
        return _accum(pt, ctx, cam_a)

    plain = run(0)
    assert plain.any()
    assert np.array_equal(plain.view(np.uint32), run(8).view(np.uint32))


def test_textured_headline_scene_through_the_loop(pt, orc, ctx, look_ahead):
    """The headline scene's own ingredients - textured sky cube and floor, the car's 22 textures and 83 materials (a 20 000-triangle
    car here), mirrors and glass, the retry queue - through 24 calls of runCUDA with windows up to 8, into a page-locked display
    buffer: the float buffer equals one pass of 24 samples and the oracle, the displayed image equals the oracle's tone map."""
    H, L = pt.host(), pt.lib()
    look_ahead(8)
    desc, cam = pt.load_scene("lamborghini", width=160, height=90, textures=1, car_tris=20000)
    scene = H.pth_scene_device()
    n = cam.width * cam.height
    assert H.pth_kernel_init() == 0
    raw = C.c_void_p()
    pt.check(L.ptx_host_alloc(ctx, n * 4, C.byref(raw)))
    try:
        pp = C.cast(raw, C.POINTER(C.c_ubyte))
        for it in range(1, 25):
            assert H.pth_run_iteration(it, pp) == it
        shown = np.ctypeslib.as_array(pp, shape=(n, 4)).copy()
    finally:
        pt.check(L.ptx_host_free(ctx, raw))
    loop = _accum(pt, ctx, cam)
    batch, st = gpuops.render(ctx, scene, cam, 0, 24, 0)
    assert np.array_equal(loop.view(np.uint32), batch.view(np.uint32))
    ref, st_o = orc.render(desc, cam, 0, 24, 0)
    assert np.array_equal(loop.view(np.uint32), ref.view(np.uint32))
    assert st_o.hit_textured > 1000 and st_o.hit_mesh > 1000
    out_o = np.empty((n, 4), np.uint8)
    assert orc.lib().orc_tonemap(orc.fptr(ref), n, 24, cam.gamma_corr, out_o.ctypes.data_as(C.POINTER(C.c_ubyte))) == 0
    assert np.abs(shown.astype(int) - out_o.astype(int)).max() <= 1
