"""Can a small kernel on a second stream run while the persistent kernels of a render pass hold the chip? Context A renders
passes of the headline frame back to back; context B (own stream) tone-maps its own buffer into page-locked memory, and the
latency of those calls is compared with the same calls on an idle GPU. (Feasibility of tracing the NEXT look-ahead window
while the calls of the current one are served: DESIGN.md 3.6.)  Usage (GPU box): python tools/corun_probe.py [spp_per_pass]"""
import sys, os, importlib, time, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L, H = pt.lib(), pt.host()
desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1)
ctxA, sceneA = H.pth_context(), H.pth_scene_device()
ctxB = C.c_void_p(); pt.check(L.ptx_ctx_create(0, C.byref(ctxB)))
for c in (ctxA, ctxB):
    pt.check(L.ptx_render_init(c, C.byref(cam), None))
n_pix = cam.width * cam.height
raw = C.c_void_p(); pt.check(L.ptx_host_alloc(ctxB, n_pix * 4, C.byref(raw)))
pbo = C.cast(raw, C.POINTER(C.c_ubyte))


def tonemaps(n):
    ts = []
    for i in range(n):
        t0 = time.perf_counter()
        pt.check(L.ptx_tonemap(ctxB, 1 + i, C.c_float(0.5), pbo))
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    return ts


pt.check(L.ptx_render_samples(ctxA, sceneA, 0, spp, 0, 0)); pt.check(L.ptx_render_sync(ctxA))
idle = tonemaps(200)
print(f"idle GPU:          tone map + copy  median {idle[100]:.3f} ms  p90 {idle[180]:.3f}  max {idle[-1]:.3f}")
t0 = time.perf_counter()
passes = 40
for s in range(passes):
    pt.check(L.ptx_render_samples(ctxA, sceneA, (1 + s) * spp, spp, 0, 0))   # asynchronous: all enqueued at once
busy = tonemaps(200)
t_tm = time.perf_counter() - t0
pt.check(L.ptx_render_sync(ctxA))
t_all = time.perf_counter() - t0
print(f"under {passes} passes of {spp}: tone map + copy  median {busy[100]:.3f} ms  p90 {busy[180]:.3f}  max {busy[-1]:.3f}   (200 calls took {1e3 * t_tm:.1f} ms, the passes {1e3 * t_all:.1f} ms)")
pt.check(L.ptx_host_free(ctxB, raw))
