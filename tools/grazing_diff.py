"""The grazing rays on which the wide traversal and the oracle disagree (tools/grazing_stress.py), one by one:
which triangle each found, at which t, and how well conditioned those hits are.
Usage (GPU box): python tools/grazing_diff.py [scene] [rays]"""
import sys, os, importlib, ctypes
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
pt = importlib.import_module("path-tracer_amd")
import gpuops, orc
import grazing_stress as gs

scene_name = sys.argv[1] if len(sys.argv) > 1 else "lamborghini"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
ctx = pt.host().pth_context()
rng = np.random.default_rng(5)
if scene_name.startswith("soup"):
    import tempfile
    tmp = tempfile.mkdtemp()
    gs.write_soup(os.path.join(tmp, "soup.obj"), 20000, rng, offset=float(scene_name[4:] or 0))
    with open(os.path.join(tmp, "s.json"), "w") as f:
        f.write('{"camera": {"pixelWidth": 64, "pixelHeight": 48, "direction": [0,0,-1], "up": [0,1,0], "position": [0,0,20],'
                ' "screenDist": 1.5, "screenHeight": 2}, "objects": [{"type": "mesh", "render": true, "src": "%s"}]}' % os.path.join(tmp, "soup.obj"))
    scene_name = os.path.join(tmp, "s.json")
pt.check(pt.lib().ptx_ctx_tune(ctx, b"grp_pad_ppm", 10))      # same order of uploads as the stress tool, so the rays are the same
desc, cam = pt.load_scene(scene_name, 64, 48)
scene = pt.host().pth_scene_device()
d = desc.contents
obj = max(range(d.num_objects), key=lambda i: d.objects[i].num_nodes)
o, dr, tgt = gs.grazing_rays(desc, obj, n, rng)
ho = orc.trace(desc, o, dr, obj)
pt.check(pt.lib().ptx_ctx_tune(ctx, b"grp_pad_ppm", -1))
desc, cam = pt.load_scene(scene_name, 64, 48)
scene = pt.host().pth_scene_device()
d = desc.contents
tris = np.ctypeslib.as_array(ctypes.cast(d.triangles, ctypes.POINTER(pt.PtxTriangle * d.num_triangles)).contents)
tris = tris[d.objects[obj].first_triangle: d.objects[obj].first_triangle + d.objects[obj].num_triangles]
res = {}
for variant in (0, 15):
    pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", variant))
    res[variant] = gpuops.trace_queued(ctx, scene, o, dr, obj)
pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", -1))
hw, he = res[0], res[15]
bad = np.nonzero((ho["object"] != hw["object"]) | (ho["prim"] != hw["prim"]) | (ho["t_bits"] != hw["t_bits"]))[0]
print(len(bad), "rays differ (wide vs oracle);", int(((ho["prim"] != he["prim"]) | (ho["t_bits"] != he["t_bits"])).sum()), "differ exact vs oracle")

def cond(i, prim):
    if prim < 0 or prim >= len(tris): return "-"
    t = tris[prim]
    a, e1, e2 = (t[k].astype(np.float64) for k in ("a", "e1", "e2"))
    dd, oo = dr[i].astype(np.float64), o[i].astype(np.float64)
    h = np.cross(dd, e2); det = e1 @ h
    nrm = np.linalg.norm(np.cross(e1, e2))
    s = oo - a; u = (s @ h) / det; q = np.cross(s, e1); v = (dd @ q) / det; tt = (e2 @ q) / det
    return "det/|n| %.2e (cos of incidence)  t64 %.7g u %.4f v %.4f  |e| %.3g %.3g" % (abs(det) / max(nrm, 1e-300), tt, u, v, np.linalg.norm(e1), np.linalg.norm(e2))

for i in bad[:12]:
    to, tw = ho["t_bits"][i:i + 1].view(np.float32)[0], hw["t_bits"][i:i + 1].view(np.float32)[0]
    print("ray %d  oracle: prim %d t %.7g   wide: prim %d t %.7g" % (i, int(ho["prim"][i].astype(np.int32)), to, int(hw["prim"][i].astype(np.int32)), tw))
    print("    oracle's triangle:", cond(i, int(ho["prim"][i].astype(np.int32))))
    print("    wide's triangle:  ", cond(i, int(hw["prim"][i].astype(np.int32))))
