"""Adversarial cameras for the PAIR boxes of single-leaf meshes (segment kernels): a floor of two triangles inside a closed
luminous box, seen from a hair above the floor's plane along and across its edges and diagonal, so that millions of camera
rays graze the triangles' planes beside their edges - the regime where the reference's triangle test accepts rounding
noise. Default schedule against the megakernel (every triangle, no boxes of its own).
Usage (GPU box): python tools/pair_box_stress.py [spp]"""
import sys, os, importlib, tempfile, pathlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tmp = pathlib.Path(tempfile.mkdtemp())
c = np.array([[x, y, z] for x in (-40.0, 47.0) for y in (-3.0, 51.0) for z in (-44.0, 39.0)])
faces = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
lines = ["mtllib m.mtl", "usemtl sky.lumi"] + ["v %.9g %.9g %.9g" % tuple(q) for q in c]
for f in faces:
    lines += ["f %d %d %d" % (f[0] + 1, f[1] + 1, f[2] + 1), "f %d %d %d" % (f[0] + 1, f[2] + 1, f[3] + 1)]
(tmp / "box.obj").write_text("\n".join(lines) + "\n")
(tmp / "floor.obj").write_text("mtllib m.mtl\nusemtl ground.spec\nv -20 0 -20\nv 20 0 -20\nv 20 0 20\nv -20 0 20\nf 1 3 2\nf 1 4 3\n")
(tmp / "m.mtl").write_text("newmtl sky.lumi\nKd 1 1 1\nKe 0.9 0.8 0.7\nnewmtl ground.spec\nKd 0.7 0.6 0.5\nNs 300\nd 0.5\n")
ctx = pt.host().pth_context()
L = pt.lib()
total = 0
cams = []
for height in (1e-3, 1e-5, 3e-2):
    cams += [((19.99, height, 15.0), (0.0, 0.0, -1.0)), ((25.0, height, 15.0), (-1.0, 0.0, -1.0)), ((-19.0, height, -19.5), (1.0, 0.0, 1.0)),
             ((0.0, height, 0.0), (1.0, 0.0, 0.3)), ((30.0, height, 0.0), (-1.0, 0.0, 0.0)), ((-39.0, height, -43.0), (1.0, 0.0, 1.0))]
for pos, dr in cams:
    dn = np.array(dr) / np.linalg.norm(dr)
    (tmp / "s.json").write_text(
        '{"camera": {"pixelWidth": 1920, "pixelHeight": 1080, "direction": [%.9g,%.9g,%.9g], "up": [0,1,0], "position": [%.9g,%.9g,%.9g],'
        ' "screenDist": 1.5, "screenHeight": 2}, "objects": ['
        '{"type": "mesh", "render": true, "src": "%s"}, {"type": "mesh", "render": true, "src": "%s"}]}'
        % (*dn, *pos, tmp / "box.obj", tmp / "floor.obj"))
    desc, cam = pt.load_scene(str(tmp / "s.json"))
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    out = {}
    for label, key, val in (("default", b"megakernel", 0), ("megakernel", b"megakernel", 1)):
        pt.check(L.ptx_ctx_tune(ctx, key, val))
        out[label] = gpuops.render(ctx, scene, cam, 0, spp, 9, batch=64)
        pt.check(L.ptx_ctx_tune(ctx, key, -1))
    bad = np.nonzero((out["default"][0].view(np.uint32) != out["megakernel"][0].view(np.uint32)).any(axis=1))[0]
    total += len(bad)
    print(f"camera {pos} -> {dr}: {out['default'][1].rays} / {out['megakernel'][1].rays} segments, {len(bad)} pixels differ"
          + (f" (first {bad[:6].tolist()})" if len(bad) else ""), flush=True)
print("TOTAL differing pixels:", total)
