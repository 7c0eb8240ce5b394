"""Where the default schedule and the strict schedule (trav_variant 15) differ on PRODUCTION rays, and which of them the
oracle agrees with: the headline frame one sample at a time through both, then the oracle on the samples that differ.
Usage (GPU box): python tools/strict_diff.py [samples] [first sample] [oracle threads]"""
import sys, os, importlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops, orc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
seed = 7
ctx = pt.host().pth_context()
desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1)
scene = pt.host().pth_scene_device()
gpuops.render_init(ctx, cam)
found = []
for s in range(first, first + S):
    out = {}
    for variant in (0, 15):
        pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", variant))
        out[variant], st = gpuops.render(ctx, scene, cam, s, 1, seed)
        out[variant] = (out[variant], st.rays)
    bad = np.nonzero((out[0][0].view(np.uint32) != out[15][0].view(np.uint32)).any(axis=1))[0]
    if len(bad):
        found.append((s, bad, out[0][0][bad].copy(), out[15][0][bad].copy(), out[0][1], out[15][1]))
        print(f"sample {s}: pixels {bad.tolist()} differ; segments default {out[0][1]} strict {out[15][1]}", flush=True)
pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", -1))
print(f"{len(found)} of {S} samples hold a differing pixel", flush=True)
agree = {"default": 0, "strict": 0, "neither": 0}
for s, bad, a0, a15, r0, r15 in found[:12]:
    ao, st = orc.render(desc, cam, s, 1, seed, nthreads=threads)
    for k, p in enumerate(bad):
        d_ok = np.array_equal(ao[p].view(np.uint32), a0[k].view(np.uint32))
        s_ok = np.array_equal(ao[p].view(np.uint32), a15[k].view(np.uint32))
        agree["default" if d_ok else "strict" if s_ok else "neither"] += 1
        print(f"sample {s} pixel {p} (x {p % cam.width}, y {p // cam.width}): oracle {ao[p][:3]} default {a0[k][:3]} strict {a15[k][:3]} -> oracle agrees with "
              f"{'default' if d_ok else 'strict' if s_ok else 'NEITHER'}; oracle segments {st.rays}", flush=True)
print("oracle agrees with:", agree)
