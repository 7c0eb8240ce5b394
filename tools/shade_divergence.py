"""Shading divergence of the segment kernel on the bench workload (the material-sorting question):
how many BSDF branches a shading wave runs and how its lanes split by material type.
Usage (GPU box): python tools/shade_divergence.py [scene] [spp]"""
import sys, os, importlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops
name = sys.argv[1] if len(sys.argv) > 1 else "lamborghini"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = pt.host().pth_context()
desc, cam = pt.load_scene(name, 1920, 1080, textures=1)
scene = pt.host().pth_scene_device()
gpuops.render_init(ctx, cam)
acc, st = gpuops.render(ctx, scene, cam, 0, spp, 0, stats=True)
d = st.as_dict()
w, l = max(1, d["sh_waves"]), max(1, d["sh_lanes"])
print(f"{name}: {d['rays']} rays, {l} shaded hits in {w} waves: {l / w:.1f} shading lanes per wave, "
      f"{d['sh_branches'] / w:.2f} of 3 BSDF branches run per shading wave")
print("lanes by material type (0 lumi, 1 diffuse, 2 specular, 3 transparent): " +
      ", ".join(f"{k}: {100.0 * d['sh_type%d' % k] / l:.1f}%" for k in range(4)))
