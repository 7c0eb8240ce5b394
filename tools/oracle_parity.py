"""The default schedule against the CPU ORACLE on production rays, at a size the suite cannot afford: the headline frame at S
samples per pixel (default 256: 1.1e9 segments, about 2.5 minutes of 16 host threads), accumulated in chunks of 16 samples on
both sides (same float order), compared bit for bit at the end. Usage (GPU box): python tools/oracle_parity.py [spp] [threads]"""
import sys, os, importlib, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops, orc

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
seed = 7
ctx = pt.host().pth_context()
desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1)
scene = pt.host().pth_scene_device()
gpuops.render_init(ctx, cam)
acc_g, st_g = gpuops.render(ctx, scene, cam, 0, spp, seed, batch=128)
acc_o = np.zeros((cam.width * cam.height, 4), np.float32)
rays_o, t0 = 0, time.time()
for s in range(0, spp, 16):
    n = min(16, spp - s)
    acc_o, st = orc.render(desc, cam, s, n, seed, accum=acc_o, nthreads=threads)
    rays_o += st.rays
    print(f"oracle: samples {s}..{s + n - 1} done, {rays_o} segments, {time.time() - t0:.0f} s", flush=True)
bad = np.nonzero((acc_o.view(np.uint32) != acc_g.view(np.uint32)).any(axis=1))[0]
print(f"{spp} spp of the headline frame: oracle {rays_o} segments, default schedule {st_g.rays}; {len(bad)} of {len(acc_o)} pixels differ"
      + (f" (first {bad[:8].tolist()})" if len(bad) else ""))
