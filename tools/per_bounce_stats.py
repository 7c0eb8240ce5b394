"""Work per bounce: counters of renders limited to 1..5 bounces, differenced."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("path-tracer_amd")
L, H = pt.lib(), pt.host()
prev = None
for nb in range(1, 6):
    desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1, bounces=nb)
    ctx, scene = H.pth_context(), H.pth_scene_device()
    pt.check(L.ptx_render_init(ctx, C.byref(cam), None))
    pt.check(L.ptx_stats_reset(ctx))
    pt.check(L.ptx_render_samples(ctx, scene, 0, 4, 0, 1))
    pt.check(L.ptx_render_sync(ctx))
    st = pt.PtxStats(); pt.check(L.ptx_stats_get(ctx, C.byref(st)))
    d = st.as_dict()
    live, q = (C.c_int32 * (nb + 1))(), (C.c_int32 * (nb + 1))()
    pt.check(L.ptx_wavefront_counts(ctx, nb, live, q))
    cur = d
    if prev:
        diff = {k: cur[k] - prev[k] for k in cur}
    else:
        diff = cur
    nq = q[nb - 1]
    print(f"bounce {nb-1}: rays {diff['rays']} queued {nq} per queued ray: inner {diff['n_inner']/max(1,nq):.1f} tri {(diff['n_tri']-14*diff['rays'])/max(1,nq):.1f} leaf {(diff['n_leaf']-2*diff['rays'])/max(1,nq):.1f} lane_util {diff['lane_steps_active']/max(1,diff['lane_steps_total']):.3f}")
    prev = cur
