"""Stress of the group-box culling (leaf modes 5/6): rays at grazing incidence that pass within a hair of
triangle edges - where the reference's Moeller-Trumbore test itself is noisiest - through the PRODUCTION
traversal kernel vs the oracle, for several paddings of the group boxes (ptx_ctx_tune "grp_pad_ppm").
Usage (GPU box): python tools/grazing_stress.py [scene] [rays]"""
import sys, os, importlib, ctypes
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import gpuops, orc


def grazing_rays(desc, obj, n, rng):
    d = desc.contents
    ob = d.objects[obj]
    tris = np.ctypeslib.as_array(ctypes.cast(d.triangles, ctypes.POINTER(pt.PtxTriangle * d.num_triangles)).contents)
    tris = tris[ob.first_triangle: ob.first_triangle + ob.num_triangles]
    t = rng.integers(0, len(tris), n)
    a = tris["a"][t].astype(np.float64); e1 = tris["e1"][t].astype(np.float64); e2 = tris["e2"][t].astype(np.float64)
    nrm = np.cross(e1, e2); nl = np.linalg.norm(nrm, axis=1, keepdims=True); ok = nl[:, 0] > 1e-20
    nrm = nrm / np.maximum(nl, 1e-30)
    edge = rng.integers(0, 3, n)
    p0 = np.where((edge == 0)[:, None], a, np.where((edge == 1)[:, None], a + e1, a + e2))
    ev = np.where((edge == 0)[:, None], e1, np.where((edge == 1)[:, None], e2 - e1, -e2))
    w = rng.random((n, 1))
    P = p0 + w * ev
    m = np.cross(ev, nrm); m /= np.maximum(np.linalg.norm(m, axis=1, keepdims=True), 1e-30)    # in-plane, across the edge
    delta = 10.0 ** rng.uniform(-7, -2, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    Q = P + delta * m
    ang = rng.uniform(0, 2 * np.pi, (n, 1))
    evn = ev / np.maximum(np.linalg.norm(ev, axis=1, keepdims=True), 1e-30)
    g = np.cos(ang) * evn + np.sin(ang) * m                                                  # random in-plane direction
    sin_a = 10.0 ** rng.uniform(-6, -1, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    dr = g * np.sqrt(1 - sin_a ** 2) + nrm * sin_a
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    L = rng.uniform(0.05, 30.0, (n, 1))
    o = Q - L * dr
    return o[ok].astype(np.float32), dr[ok].astype(np.float32), t[ok]


def write_soup(path, n_tris, rng, offset=0.0):
    """A triangle soup: small randomly oriented triangles in a 10-unit cube (little occlusion, so a grazing
    ray's nearest hit is usually the triangle it was aimed at)."""
    c = rng.uniform(-5, 5, (n_tris, 3)) + offset
    size = 10.0 ** rng.uniform(-2, -0.5, (n_tris, 1))
    v = c[:, None, :] + size[:, None, :] * rng.normal(size=(n_tris, 3, 3))
    with open(path, "w") as f:
        f.write("mtllib soup.mtl\nusemtl white.diff\n")
        for tri in v:
            for q in tri:
                f.write("v %.9g %.9g %.9g\n" % tuple(q))
        for i in range(n_tris):
            f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    with open(os.path.join(os.path.dirname(path), "soup.mtl"), "w") as f:
        f.write("newmtl white.diff\nKd 0.8 0.8 0.8\n")


def main():
    scene_name = sys.argv[1] if len(sys.argv) > 1 else "lamborghini"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
    ctx = pt.host().pth_context()
    rng = np.random.default_rng(5)
    if scene_name.startswith("soup"):
        import tempfile
        tmp = tempfile.mkdtemp()
        write_soup(os.path.join(tmp, "soup.obj"), 20000, rng, offset=float(scene_name[4:] or 0))
        with open(os.path.join(tmp, "s.json"), "w") as f:
            f.write('{"camera": {"pixelWidth": 64, "pixelHeight": 48, "direction": [0,0,-1], "up": [0,1,0], "position": [0,0,20],'
                    ' "screenDist": 1.5, "screenHeight": 2}, "objects": [{"type": "mesh", "render": true, "src": "%s"}]}' % os.path.join(tmp, "soup.obj"))
        scene_name = os.path.join(tmp, "s.json")
    first = True
    for ppm in (10, 0, 1, 100, 1000):
        pt.check(pt.lib().ptx_ctx_tune(ctx, b"grp_pad_ppm", ppm))
        desc, cam = pt.load_scene(scene_name, 64, 48)
        scene = pt.host().pth_scene_device()
        d = desc.contents
        obj = max(range(d.num_objects), key=lambda i: d.objects[i].num_nodes)
        if first:
            o, dr, tgt = grazing_rays(desc, obj, n, rng)
            ho = orc.trace(desc, o, dr, obj)
            print(f"{scene_name}: {len(o)} grazing rays, oracle hits {(ho['object'] == obj).sum()}, "
                  f"of which on the triangle aimed at {((ho['object'] == obj) & (ho['prim'] == tgt)).sum()}", flush=True)
            first = False
        for variant in (0, 7):
            pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", variant))
            hg = gpuops.trace_queued(ctx, scene, o, dr, obj)
            bad = (ho["object"] != hg["object"]) | (ho["prim"] != hg["prim"]) | (ho["t_bits"] != hg["t_bits"])
            print(f"  pad {ppm:5d} ppm  variant {variant}: {int(bad.sum())} rays differ", flush=True)
    pt.check(pt.lib().ptx_ctx_tune(ctx, b"grp_pad_ppm", -1))
    pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", -1))


if __name__ == "__main__":
    main()
