"""Print the wavefront pipeline's per-bounce live-path and traversal-queue sizes for a scene."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("path-tracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "lamborghini"
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
desc, cam = pt.load_scene(name, width=w, height=h, textures=1)
L, H = pt.lib(), pt.host()
ctx, scene = H.pth_context(), H.pth_scene_device()
pt.check(L.ptx_render_init(ctx, C.byref(cam), None))
pt.check(L.ptx_render_samples(ctx, scene, 0, 8, 0, 0))
pt.check(L.ptx_render_sync(ctx))
n = cam.max_ray_bounces
live, q = (C.c_int32 * (n + 1))(), (C.c_int32 * (n + 1))()
pt.check(L.ptx_wavefront_counts(ctx, n, live, q))
print("paths", cam.width * cam.height * 8, "live", list(live), "queued", list(q))
