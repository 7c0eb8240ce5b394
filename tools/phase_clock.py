"""Temporary: per-phase clocks of the leaf phase (needs the instrumented build)."""
import sys, os, importlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops
ctx = pt.host().pth_context()
desc, cam = pt.load_scene("lamborghini", 1920, 1080)
scene = pt.host().pth_scene_device()
gpuops.render_init(ctx, cam)
acc, st = gpuops.render(ctx, scene, cam, 0, 8, 0, stats=True)
d = [st.diag0, st.diag1, st.diag2, st.diag3]
print("dense fill %.3g  leftover fill %.3g  steps %.3g  leaf phase total %.3g (other = %.3g)" % (d[0], d[1], d[2], d[3], d[3] - d[0] - d[1] - d[2]))
print("rounds %d  leaf lanes/round %.1f  units/round %.1f  units/leaf %.2f" % (st.lane_steps_active, st.lane_steps_total / st.lane_steps_active, st.trav_rays / st.lane_steps_active, st.trav_rays / st.lane_steps_total))
