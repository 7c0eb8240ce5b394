"""Execution counters of the wide traversal kernel's counting variant (variant 14) on the bench workload
(one pass). Usage (GPU box): python tools/wide_diag.py [spp]"""
import sys, os, importlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ctx = pt.host().pth_context()
desc, cam = pt.load_scene("lamborghini", 1920, 1080, textures=1)
scene = pt.host().pth_scene_device()
pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", 14))
gpuops.render_init(ctx, cam)
acc, st = gpuops.render(ctx, scene, cam, 0, spp, 0, stats=True)
d = st.as_dict()
tr = max(1, d["trav_rays"])
it = max(1, d["x_iters"])
print("traversals", tr)
print("per traversal: wide-node visits %.2f  units %.2f  stacked nodes dropped at the pop %.2f  retried rays %.6f (%d)" %
      (d["x_item_lanes"] / tr, d["x_units"] / tr, d["x_wait_lanes"] / tr, d["x_half_lanes"] / tr, d["x_half_lanes"]))
print("per visit iteration: lanes visiting %.1f of 64 (iterations %d)" % (d["x_item_lanes"] / it, it))
print("unit steps %d: fill %.1f of 16; drains %d (%.1f units each); turns %d (%.1f iterations per turn)" %
      (d["x_unit_steps"], d["x_units"] / max(1, d["x_unit_steps"]), d["x_unit_phases"], d["x_units"] / max(1, d["x_unit_phases"]),
       d["x_turns"], it / max(1, d["x_turns"])))
tot = float(d["diag0"] + d["diag1"] + d["diag2"]) or 1.0
print("clock share: turns %.1f%%  node visits %.1f%%  unit steps %.1f%%;  retried for a full stack: %d, for a winning hit that is not robust: %d" %
      (100 * d["diag0"] / tot, 100 * d["diag1"] / tot, 100 * d["diag2"] / tot, d["diag3"], d["lane_steps_active"]))
