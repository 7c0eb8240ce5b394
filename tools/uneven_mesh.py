"""A second deterministic stand-in mesh, UNEVENLY tessellated (the procedural car of host/procedural.cpp is even): a hull of
large flat panels (edges of 2-4 units, two triangles per face), panels cut into long slivers (4 x 0.006 units), 40 densely
tessellated blobs (edges of ~0.02 units: a 100:1 spread of edge lengths, 10^4:1 of areas) and 1500 needles (aspect 100:1).
Used by tests/test_gpu_wide.py (parity on uneven geometry) and tools/uneven_bench.py (tree-quality decisions re-taken on it).
write_scene(dir, seed) -> path of the scene JSON."""
import os
import numpy as np


def _blob(c, r, nu, nv, rng):
    u = np.linspace(0, 2 * np.pi, nu + 1)
    v = np.linspace(0.02, np.pi - 0.02, nv + 1)
    uu, vv = np.meshgrid(u, v)
    bump = 1.0 + 0.15 * np.sin(5 * uu + rng.uniform(0, 6)) * np.sin(4 * vv + rng.uniform(0, 6))
    p = np.stack([np.cos(uu) * np.sin(vv), np.cos(vv), np.sin(uu) * np.sin(vv)], -1) * (r * bump)[..., None] + c
    tris = []
    for j in range(nv):
        a = p[j, :-1]; b = p[j, 1:]; cc = p[j + 1, :-1]; d = p[j + 1, 1:]
        tris.append(np.stack([a, b, d], 1)); tris.append(np.stack([a, d, cc], 1))
    return np.concatenate(tris)


def triangles(seed=3):
    rng = np.random.default_rng(seed)
    groups = []   # (material name, (n, 3, 3) array)
    # hull: six large panels, two triangles each
    lo, hi = np.array([-2.0, 0.0, -1.0]), np.array([2.0, 1.2, 1.0])
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    faces = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    hull = []
    for f in faces[:5]:   # one side stays open so that rays get inside
        a, b, c, d = corners[list(f)]
        hull += [[a, b, c], [a, c, d]]
    groups.append(("panel.diff", np.array(hull)))
    # a panel cut into 300 long slivers
    xs = np.linspace(-1.8, 1.8, 2); zs = np.linspace(-0.9, 0.9, 301)
    sl = []
    for k in range(300):
        a = [xs[0], 0.4, zs[k]]; b = [xs[1], 0.4, zs[k]]; c = [xs[1], 0.4, zs[k + 1]]; d = [xs[0], 0.4, zs[k + 1]]
        sl += [[a, b, c], [a, c, d]]
    groups.append(("shelf.spec", np.array(sl)))
    # dense detail
    blobs = []
    for i in range(40):
        c = np.array([rng.uniform(-1.7, 1.7), rng.uniform(0.55, 1.05), rng.uniform(-0.8, 0.8)])
        blobs.append(_blob(c, rng.uniform(0.08, 0.16), 48, 24, rng))
    groups.append(("detail.diff", np.concatenate(blobs[:28])))
    groups.append(("chrome.spec", np.concatenate(blobs[28:36])))
    groups.append(("glass.trans", np.concatenate(blobs[36:])))
    # needles
    base = rng.uniform([-1.8, 0.05, -0.9], [1.8, 0.35, 0.9], (1500, 3))
    dirv = rng.normal(size=(1500, 3)); dirv /= np.linalg.norm(dirv, axis=1, keepdims=True)
    side = np.cross(dirv, rng.normal(size=(1500, 3))); side /= np.linalg.norm(side, axis=1, keepdims=True)
    ln = rng.uniform(0.8, 1.6, (1500, 1)); wd = ln / 100.0
    groups.append(("needle.diff", np.stack([base, base + dirv * ln, base + dirv * ln * 0.5 + side * wd], 1)))
    return groups


def write_scene(directory, seed=3, width=96, height=64):
    os.makedirs(directory, exist_ok=True)
    groups = triangles(seed)
    with open(os.path.join(directory, "uneven.obj"), "w") as f:
        f.write("mtllib uneven.mtl\n")
        n = 0
        for name, t in groups:
            f.write("usemtl %s\n" % name)
            f.write("".join("v %.9g %.9g %.9g\n" % tuple(p) for p in t.reshape(-1, 3)))
            f.write("".join("f %d %d %d\n" % (n + 3 * i + 1, n + 3 * i + 2, n + 3 * i + 3) for i in range(len(t))))
            n += 3 * len(t)
    with open(os.path.join(directory, "uneven.mtl"), "w") as f:
        f.write("newmtl panel.diff\nKd 0.7 0.7 0.75\nnewmtl shelf.spec\nKd 0.8 0.8 0.8\nNs 700\nd 0.6\nnewmtl detail.diff\nKd 0.8 0.5 0.3\n"
                "newmtl chrome.spec\nKd 0.9 0.9 0.9\nNs 900\nd 0.9\nnewmtl glass.trans\nKd 0.95 0.95 0.95\nNi 1.5\nd 0.1\nnewmtl needle.diff\nKd 0.3 0.6 0.4\n")
    path = os.path.join(directory, "scene_uneven.json")
    with open(path, "w") as f:
        f.write('{"camera": {"pixelWidth": %d, "pixelHeight": %d, "direction": [-0.55, -0.12, -0.8], "up": [0,1,0], "position": [2.6, 1.0, 3.4],'
                ' "screenDist": 1.6, "screenHeight": 2}, "objects": ['
                '{"type": "sphere", "render": true, "radius": 40, "position": [0, 0, 0], "material": {"type": "lumi", "Kd": [1], "Ke": [2.5]}},'
                '{"type": "sphere", "render": true, "radius": 0.5, "position": [-2.9, 0.5, 0.3], "material": {"type": "spec", "Kd": [0.9], "Ns": 800, "d": 0.8}},'
                '{"type": "mesh", "render": true, "src": "%s"}]}' % (width, height, os.path.join(directory, "uneven.obj")))
    return path, sum(len(t) for _, t in groups)


if __name__ == "__main__":
    import sys
    p, n = write_scene(sys.argv[1] if len(sys.argv) > 1 else "/tmp/uneven")
    print(p, n, "triangles")
