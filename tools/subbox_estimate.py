"""Estimate (CPU, numpy) what sub-boxes inside the reference's KD leaves would save.

For the biggest mesh of a scene: every leaf's triangles are sorted along the leaf box's longest axis
and cut into groups of GROUP; a sample of rays (primary rays of the scene camera plus random
secondary rays leaving random triangles) is slab-tested against every leaf box and, where it hits,
against the leaf's group boxes. Prints leaves hit per ray, groups per leaf, the fraction of group
boxes hit given the leaf box was hit, and the triangle slots tested with / without the groups.
The leaf set per ray is a superset of what the traversal visits (no distance culling here).
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib
pt = importlib.import_module("path-tracer_amd")

GROUP = int(os.environ.get("GROUP", "8"))
NRAYS = int(os.environ.get("NRAYS", "1500"))
PAD = 5e-5


def slab(bmin, bmax, o, d):
    """rays [R,3] vs boxes [B,3] -> hit mask [R,B] (plain slab test, boxes behind the origin rejected)"""
    inv = 1.0 / d
    t0 = (bmin[None, :, :] - o[:, None, :]) * inv[:, None, :]
    t1 = (bmax[None, :, :] - o[:, None, :]) * inv[:, None, :]
    tn = np.minimum(t0, t1).max(axis=2)
    tf = np.maximum(t0, t1).min(axis=2)
    return (tn <= tf) & (tf >= 0)


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "lamborghini"
    desc, cam = pt.load_scene(scene, 1920, 1080, upload=False)
    d = desc.contents
    objs = [d.objects[i] for i in range(d.num_objects)]
    ob = max(objs, key=lambda o: o.num_nodes)
    nodes = np.ctypeslib.as_array(C_cast(d.nodes, d.num_nodes, pt.PtxKDNode))[ob.first_node: ob.first_node + ob.num_nodes]
    tris = np.ctypeslib.as_array(C_cast(d.triangles, d.num_triangles, pt.PtxTriangle))[ob.first_triangle: ob.first_triangle + ob.num_triangles]
    leaf_tris = np.ctypeslib.as_array(d.leaf_tris, shape=(d.num_leaf_tris,))[ob.first_leaf_tri: ob.first_leaf_tri + ob.num_leaf_tris]
    a = tris["a"].astype(np.float64); b = a + tris["e1"]; c = a + tris["e2"]
    tmin = np.minimum(np.minimum(a, b), c) - PAD
    tmax = np.maximum(np.maximum(a, b), c) + PAD
    cen = (a + b + c) / 3.0
    leaf = (nodes["left"] == -1) & (nodes["right"] == -1) & (nodes["num_tris"] > 0)
    lidx = np.nonzero(leaf)[0]
    print(f"{scene}: object with {ob.num_triangles} triangles, {ob.num_nodes} nodes, {len(lidx)} leaves, "
          f"mean {nodes['num_tris'][lidx].mean():.1f} tris/leaf, max {nodes['num_tris'][lidx].max()}")
    lmin = nodes["bmin"][lidx].astype(np.float64); lmax = nodes["bmax"][lidx].astype(np.float64)
    # groups
    gmin, gmax, gleaf, gcount = [], [], [], []
    for li, n in enumerate(lidx):
        ids = leaf_tris[nodes["first_tri"][n]: nodes["first_tri"][n] + nodes["num_tris"][n]]
        ax = int(np.argmax(lmax[li] - lmin[li]))
        ids = ids[np.argsort(cen[ids, ax], kind="stable")]
        for s in range(0, len(ids), GROUP):
            g = ids[s: s + GROUP]
            gmin.append(tmin[g].min(axis=0)); gmax.append(tmax[g].max(axis=0)); gleaf.append(li); gcount.append(len(g))
    gmin = np.array(gmin); gmax = np.array(gmax); gleaf = np.array(gleaf); gcount = np.array(gcount)
    print(f"groups of {GROUP}: {len(gleaf)} ({len(gleaf) / len(lidx):.2f} per leaf)")
    rng = np.random.default_rng(1)
    # rays: half primary, half secondary
    pos = np.array(cam.position[:], dtype=np.float64); dr = np.array(cam.direction[:]); up = np.array(cam.up[:]); rt = np.array(cam.right[:])
    n1 = NRAYS // 2
    px = rng.uniform(-1, 1, n1) * cam.screen_half[0]; py = rng.uniform(-1, 1, n1) * cam.screen_half[1]
    d1 = cam.screen_dist * dr[None, :] + px[:, None] * rt[None, :] + py[:, None] * up[None, :]
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    o1 = np.repeat(pos[None, :], n1, axis=0)
    t = rng.integers(0, len(tris), NRAYS - n1)
    o2 = cen[t] + 1e-3 * rng.normal(size=(len(t), 3))
    d2 = rng.normal(size=(len(t), 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    for name, o, dd in (("primary", o1, d1), ("secondary", o2, d2)):
        dd = np.where(np.abs(dd) < 1e-12, 1e-12, dd)
        hl = slab(lmin, lmax, o, dd)                  # [R, L]
        hg = slab(gmin, gmax, o, dd)                  # [R, G]
        hg &= hl[:, gleaf]
        leaves_hit = hl.sum()
        groups_in_hit_leaves = hl[:, gleaf].sum()
        groups_hit = hg.sum()
        slots_now = (hl * (32 * np.ceil(nodes["num_tris"][lidx] / 32.0))[None, :]).sum()
        slots_grp = (hg * GROUP).sum()
        tris_now = (hl * nodes["num_tris"][lidx][None, :]).sum()
        print(f"{name:9s}: {leaves_hit / len(o):6.1f} leaf boxes hit per ray; group boxes hit {groups_hit / max(groups_in_hit_leaves, 1):.2%} "
              f"of those in hit leaves ({groups_hit / max(leaves_hit, 1):.2f} per leaf); triangle slots {slots_now / len(o):.0f} -> {slots_grp / len(o):.0f} "
              f"(real triangles {tris_now / len(o):.0f})")


def C_cast(ptr, n, struct):
    import ctypes
    arr = ctypes.cast(ptr, ctypes.POINTER(struct * n)).contents
    return arr


if __name__ == "__main__":
    main()
