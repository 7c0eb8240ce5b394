"""Three schedules of the same samples against each other on PRODUCTION rays, every shipped scene at its bench size: the
default (wide traversal, pair boxes, bounce loop in the launch), the strict schedule (trav_variant 15: the reference's
visiting order, every triangle of every leaf and of every single-leaf mesh) and the one-lane-per-pixel megakernel (the
reference's loops as written). Any pixel whose float accumulation differs is printed; tools/strict_diff.py then asks the oracle.
Usage (GPU box): python tools/schedule_parity.py [spp]"""
import sys, os, importlib, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = pt.host().pth_context()
L = pt.lib()
total_bad = 0
for name, w, h in (("lamborghini", 1920, 1080), ("geoms", 1024, 768), ("teapot", 1920, 1080), ("sky_map", 3840, 2160),
                   ("example", 512, 512), ("blender_comp", 1920, 1080)):
    desc, cam = pt.load_scene(name, width=w, height=h, textures=1)
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    out = {}
    for label, key, val in (("default", b"trav_variant", 0), ("strict", b"trav_variant", 15), ("megakernel", b"megakernel", 1)):
        pt.check(L.ptx_ctx_tune(ctx, key, val))
        t0 = time.time()
        acc, st = gpuops.render(ctx, scene, cam, 0, spp, 11, batch=64)
        out[label] = (acc, st.rays, time.time() - t0)
        pt.check(L.ptx_ctx_tune(ctx, key, -1))
    line = f"{name} {w}x{h} {spp} spp: " + ", ".join(f"{k} {v[1]} segments {v[2]:.2f} s" for k, v in out.items())
    for other in ("strict", "megakernel"):
        bad = np.nonzero((out["default"][0].view(np.uint32) != out[other][0].view(np.uint32)).any(axis=1))[0]
        total_bad += len(bad)
        line += f"; default vs {other}: {len(bad)} pixels differ" + (f" (first {bad[:6].tolist()}, x {(bad[:6] % w).tolist()}, y {(bad[:6] // w).tolist()})" if len(bad) else "")
    print(line, flush=True)
print("TOTAL differing pixels:", total_bad)
