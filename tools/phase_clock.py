"""Where the cooperative traversal kernel spends its time: the counting variant sums a constant-rate
clock per phase into PtxStats.diag0..3 (include/ptx_hip.h). Usage (GPU box): python tools/phase_clock.py [scene]"""
import sys, os, importlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops
ctx = pt.host().pth_context()
desc, cam = pt.load_scene(sys.argv[1] if len(sys.argv) > 1 else "lamborghini", 1920, 1080)
scene = pt.host().pth_scene_device()
gpuops.render_init(ctx, cam)
acc, st = gpuops.render(ctx, scene, cam, 0, 128, 0, stats=True)   # one 128-sample pass, as in bench.py
d = [st.diag0, st.diag1, st.diag2, st.diag3]
tot = float(d[0] + d[1] + d[2]) or 1.0
print("ticks : commit+refill %.3g  descent %.3g  leaf phase %.3g (of which unit steps %.3g)" % tuple(d))
print("share : commit+refill %.1f%%  descent %.1f%%  group boxes + hand-over %.1f%%  unit steps %.1f%%" %
      (100 * d[0] / tot, 100 * d[1] / tot, 100 * (d[2] - d[3]) / tot, 100 * d[3] / tot))
print("per traversal: %.1f inner steps, %.1f leaves, %.1f triangles in those leaves" %
      (st.trav_inner / max(st.trav_rays, 1), st.trav_leaf / max(st.trav_rays, 1), st.trav_tri / max(st.trav_rays, 1)))
