"""Differential run of the schedules over the render path's PARAMETERS: scenes x bounce limits x odd frame sizes x sample
offsets x texture / preview switches, default schedule against the strict schedule and the megakernel (tools/schedule_parity.py
does the bench sizes at high sample counts). Prints every configuration whose accumulation buffers differ.
Usage (GPU box): python tools/schedule_fuzz.py [spp]"""
import sys, os, importlib, itertools, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = pt.host().pth_context()
L = pt.lib()
n_cfg = n_bad = 0
t0 = time.time()
for name in ("lamborghini", "blender_comp", "sky_map", "teapot", "geoms"):
    for (w, h), bounces, (textures, preview), first in itertools.product(
            ((333, 217), (1921, 1081), (64, 1), (1, 1)), (0, 1, 3, 8, 20, 64), ((1, 0), (0, 0), (1, 1)), (0, 1 << 30)):
        if (bounces in (20, 64) or preview) and (w, h) == (1921, 1081):
            continue   # keep the run short: the deep limits and the preview mode at the small sizes only
        desc, cam = pt.load_scene(name, width=w, height=h, textures=textures, preview=preview, bounces=bounces)
        scene = pt.host().pth_scene_device()
        gpuops.render_init(ctx, cam)
        out = {}
        for label, key, val in (("default", b"trav_variant", 0), ("strict", b"trav_variant", 15), ("megakernel", b"megakernel", 1)):
            pt.check(L.ptx_ctx_tune(ctx, key, val))
            acc, st = gpuops.render(ctx, scene, cam, first, spp, 5, batch=5)
            out[label] = (acc, st.rays)
            pt.check(L.ptx_ctx_tune(ctx, key, -1))
        n_cfg += 1
        for other in ("strict", "megakernel"):
            bad = np.nonzero((out["default"][0].view(np.uint32) != out[other][0].view(np.uint32)).any(axis=1))[0]
            if len(bad) or out["default"][1] != out[other][1]:
                n_bad += 1
                print(f"DIFFER {name} {w}x{h} bounces {bounces} textures {textures} preview {preview} first {first}: default vs {other}: "
                      f"{len(bad)} pixels, segments {out['default'][1]} vs {out[other][1]}, first pixels {bad[:5].tolist()}", flush=True)
    print(f"{name}: {n_cfg} configurations so far, {n_bad} differ, {time.time() - t0:.0f} s", flush=True)
print(f"TOTAL {n_cfg} configurations, {n_bad} differ")
