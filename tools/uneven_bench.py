"""Tree-quality decisions re-taken on UNEVEN geometry (tools/uneven_mesh.py): Mrays/s at 1920x1080 and what a traversal
visits (wide nodes, work units) for the two candidates of the wide tree (wide_build 1 = collapsed from the reference's KD tree,
2 = surface-area heuristic, 0 = the library's choice) and for the two ways of cutting a leaf's triangles into groups of eight (grp_mode 1 = recursive
bisection, the default; 0 = strips along the leaf's longest axis), next to the evenly tessellated car.
Usage (GPU box): python tools/uneven_bench.py"""
import sys, os, importlib, time, tempfile, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
pt = importlib.import_module("path-tracer_amd")
import gpuops, uneven_mesh

L = pt.lib()
ctx = pt.host().pth_context()
path, ntri = uneven_mesh.write_scene(tempfile.mkdtemp(), width=1920, height=1080)
L.ptx_scene_wide_sah_objects.argtypes = [C.c_void_p]
for scene_name, label in ((path, "uneven mesh (%d triangles)" % ntri), ("lamborghini", "procedural car")):
    for mode, build in ((1, 0), (0, 0), (1, 1), (1, 2)):
        pt.check(L.ptx_ctx_tune(ctx, b"grp_mode", mode))
        pt.check(L.ptx_ctx_tune(ctx, b"wide_build", build))
        desc, cam = pt.load_scene(scene_name, 1920, 1080, textures=1)
        scene = pt.host().pth_scene_device()
        gpuops.render_init(ctx, cam)
        gpuops.render(ctx, scene, cam, 0, 128, 0)   # warm-up with the pass size of the timed run (the state buffers are sized by it)
        t0 = time.time()
        acc, st = gpuops.render(ctx, scene, cam, 128, 256, 0, batch=128)
        dt = time.time() - t0
        pt.check(L.ptx_ctx_tune(ctx, b"trav_variant", 14))
        _, sx = gpuops.render(ctx, scene, cam, 0, 16, 0, stats=True)
        pt.check(L.ptx_ctx_tune(ctx, b"trav_variant", 15))
        _, se = gpuops.render(ctx, scene, cam, 0, 16, 0, stats=True)   # the reference's own loop, counted
        pt.check(L.ptx_ctx_tune(ctx, b"trav_variant", -1))
        d, e = sx.as_dict(), se.as_dict()
        tr = max(1, d["trav_rays"])
        print("%-34s grp_mode %d wide_build %d (SAH tree on %d meshes): %7.0f Mrays/s; stack overflows %d; per traversal %.2f wide-node visits, %.2f work units; %.1f %% of segments traverse (reference loop: %.1f inner steps, %.1f triangle tests)" %
              (label, mode, build, L.ptx_scene_wide_sah_objects(scene), st.rays / dt / 1e6, d["diag3"], d["x_item_lanes"] / tr, d["x_units"] / tr, 100.0 * tr / max(1, d["rays"]),
               e["trav_inner"] / max(1, e["trav_rays"]), e["trav_tri"] / max(1, e["trav_rays"])), flush=True)
pt.check(L.ptx_ctx_tune(ctx, b"grp_mode", -1))
pt.check(L.ptx_ctx_tune(ctx, b"wide_build", -1))
