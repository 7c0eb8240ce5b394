"""Do the two kernel classes of a step overlap when two half-batches run on two HIP streams? Two contexts on one
GPU, each with its own stream, state and accumulation buffer, render disjoint sample ranges of the headline frame
at the same time (launches are asynchronous: one host thread enqueues both); compared with one context doing the
same samples alone. Usage (GPU box): python tools/overlap_probe.py [steps] [spp_per_step]
Environment knobs forwarded to ptx_ctx_tune on both contexts: TRAV_BLOCKS."""
import sys, os, importlib, time, ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (HIP runtime order)
pt = importlib.import_module("path-tracer_amd")

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L, H = pt.lib(), pt.host()
desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1)
ctx1, scene1 = H.pth_context(), H.pth_scene_device()
ctx2 = C.c_void_p(); pt.check(L.ptx_ctx_create(0, C.byref(ctx2)))
scene2 = C.c_void_p(); pt.check(L.ptx_scene_upload(ctx2, desc, C.byref(scene2)))
for c in (ctx1, ctx2):
    pt.check(L.ptx_render_init(c, C.byref(cam), None))
    pt.check(L.ptx_accum_clear(c))


def run(label, plan, tb=None):
    """plan: list of (ctx, scene, first, count) enqueued per step"""
    for c in (ctx1, ctx2):
        pt.check(L.ptx_ctx_tune(c, b"trav_blocks", -1 if tb is None else tb))
    for warm in (True, False):
        for c in (ctx1, ctx2):
            pt.check(L.ptx_render_sync(c)); pt.check(L.ptx_stats_reset(c))
        t0 = time.time()
        for s in range(1 if warm else steps):
            for (c, sc, first, count) in plan:
                pt.check(L.ptx_render_samples(c, sc, s * spp + first, count, 0, 0))
        for c in (ctx1, ctx2):
            pt.check(L.ptx_render_sync(c))
        dt = time.time() - t0
    rays = 0
    for c in (ctx1, ctx2):
        st = pt.PtxStats(); pt.check(L.ptx_stats_get(c, C.byref(st))); rays += st.rays
    print(f"{label:48s} {dt / steps * 1e3:8.2f} ms/step  {rays / dt / 1e6:8.0f} Mrays/s", flush=True)


h = spp // 2
run("one context, %d spp per call" % spp, [(ctx1, scene1, 0, spp)])
run("one context, 2 x %d spp per step" % h, [(ctx1, scene1, 0, h), (ctx1, scene1, h, h)])
run("two contexts, %d spp each per step" % h, [(ctx1, scene1, 0, h), (ctx2, scene2, h, h)])
for tb in (2, 3, 4):
    run("two contexts, %d spp each, trav_blocks %d" % (h, tb), [(ctx1, scene1, 0, h), (ctx2, scene2, h, h)], tb)
run("two contexts, %d spp each per step" % spp, [(ctx1, scene1, 0, spp), (ctx2, scene2, spp, spp)])
