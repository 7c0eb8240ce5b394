"""Per-bounce traversal work and lane-use counters of the bench workload (stats variant).
usage: python tools/per_bounce_stats.py [spp]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("path-tracer_amd")
L, H = pt.lib(), pt.host()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prev = None
for maxb in range(1, 6):
    desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1, bounces=maxb)
    ctx, scene = H.pth_context(), H.pth_scene_device()
    pt.check(L.ptx_render_init(ctx, C.byref(cam), None))
    pt.check(L.ptx_stats_reset(ctx))
    pt.check(L.ptx_render_samples(ctx, scene, 0, spp, 0, 1))
    pt.check(L.ptx_render_sync(ctx))
    st = pt.PtxStats(); pt.check(L.ptx_stats_get(ctx, C.byref(st))); d = st.as_dict()
    cur = d if prev is None else {k: d[k] - prev[k] for k in d}
    r = max(1, cur["trav_rays"])
    print(f"bounce {maxb-1}: trav_rays {cur['trav_rays']:>9} inner/ray {cur['trav_inner']/r:6.1f} tri/ray {cur['trav_tri']/r:6.1f} leaf/ray {cur['trav_leaf']/r:5.1f} "
          f"step-eff {cur['lane_steps_active']/max(1,cur['lane_steps_total']):.3f} diag0/1 {cur['diag0']/max(1,cur['diag1']):.3f} diag2/3 {cur['diag2']/max(1,cur['diag3']):.3f} raw diag {cur['diag0']:.4g} {cur['diag1']:.4g} {cur['diag2']:.4g} {cur['diag3']:.4g}", flush=True)
    prev = d
