"""Default traversal (variant 0) against the strict kernel alone (variant 15: every triangle of every leaf the reference visits) on the PRODUCTION ray distribution: the
headline frame at S samples per pixel (default 1000 = 4.4e9 path segments); prints how many pixels of the float
accumulation buffers differ. Usage (GPU box): python tools/production_parity.py [spp]"""
import sys, os, importlib, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = pt.host().pth_context()
desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1)
scene = pt.host().pth_scene_device()
gpuops.render_init(ctx, cam)
out = {}
for variant in (15, 0):
    pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", variant))
    t0 = time.time()
    out[variant], st = gpuops.render(ctx, scene, cam, 0, spp, 7, batch=128)
    print(f"variant {variant}: {st.rays} segments in {time.time() - t0:.2f} s", flush=True)
pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", -1))
bad = (out[15].view(np.uint32) != out[0].view(np.uint32)).any(axis=1)
print(f"{spp} spp: {int(bad.sum())} of {len(bad)} pixels differ between the default traversal and the strict kernel")
