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


def test_oracle_bsdfs_obey_the_optics_they_restate(pt, orc):
    """The oracle's three BSDFs (reference: src/surfaces.cuh:14-137) against their definitions in float64 - an
    independent statement of WHAT they compute, not of how: sampled directions are unit vectors inside the stated cone
    about the stated axis, new origins are the hit point moved along the normal by the reference's biases, the
    refracted direction obeys Snell's law, and the refractive mask factors are the Schlick / energy ratios."""
    rng = np.random.default_rng(7)
    n = 3000
    cases = (pt.PtxBsdfCase * n)()

    def unit(v):
        return v / np.linalg.norm(v)

    spec = []
    for i in range(n):
        c = cases[i]
        nrm = unit(rng.normal(size=3))
        d = unit(rng.normal(size=3))
        if np.dot(d, nrm) > 0 and i % 3 != 2:   # rays arrive from outside for the opaque materials
            d = d - 2 * np.dot(d, nrm) * nrm
        pnt = rng.uniform(-5, 5, 3)
        mt = 1 + i % 3
        refl, ior, sharp = rng.uniform(0.05, 0.95), rng.uniform(1.1, 2.0), rng.uniform(0.0, 1.0)
        u = rng.uniform(1e-3, 1.0, 4)
        for k in range(3):
            c.dir[k], c.normal[k], c.point[k] = d[k], nrm[k], pnt[k]
        c.mat_type, c.reflect_factor, c.refract_index, c.sharpness = mt, refl, ior, sharp
        for k in range(4):
            c.u[k] = u[k]
        spec.append((np.float32(d).astype(np.float64), np.float32(nrm).astype(np.float64), np.float32(pnt).astype(np.float64),
                     mt, float(np.float32(refl)), float(np.float32(ior)), float(np.float32(sharp)), np.float32(u).astype(np.float64)))
    assert orc.lib().orc_bsdf_eval(n, cases) == 0
    seen = {1: 0, 2: 0, 3: 0, "tir": 0, "refr": 0, "refl": 0}
    for i, (d, nrm, pnt, mt, refl, ior, sharp, u) in enumerate(spec):
        c = cases[i]
        od, oo, om = np.array(c.out_dir[:], np.float64), np.array(c.out_origin[:], np.float64), np.array(c.out_mask[:], np.float64)
        assert abs(np.linalg.norm(od) - 1.0) < 1e-5, i
        if mt == 1:   # Diffuse_BRDF: cosine-ish cone of half-angle pi/2 about the normal, origin + 5e-5 n
            assert np.dot(od, nrm) > -1e-3, i
            assert np.allclose(oo, pnt + 5e-5 * nrm, atol=2e-6), i
            assert np.array_equal(om, [1, 1, 1])
            seen[1] += 1
        elif mt == 2:   # Specular_Diffuse_BRDF: u0 > reflect_factor -> diffuse; else a cone about the mirror direction
            if u[0] > refl:
                assert np.dot(od, nrm) > -1e-3 and np.allclose(oo, pnt + 5e-5 * nrm, atol=2e-6), i
            else:
                mirror = d - 2 * np.dot(nrm, d) * nrm
                max_cone = (np.pi / 2) * (1.0 - sharp)
                scatt = np.pi / 2 - np.arccos(min(1.0, abs(np.dot(d, nrm))))
                ang = min(max_cone, scatt)
                # rand_cone_Dir: sin of the deviation is at most sqrt(sin(ang)) (r^2 = sin(ang) * u, |offset| = r, w = sqrt(1 - r^2 + 1e-4))
                cosdev = np.dot(od, mirror)
                assert cosdev > np.sqrt(max(0.0, 1.0 - np.sin(ang))) - 2e-3, (i, cosdev, ang)
                assert np.allclose(oo, pnt + 1e-4 * nrm, atol=2e-6), i
            seen[2] += 1
        else:   # Refractive_BRDF
            n_or = nrm if np.dot(nrm, d) < 0 else -nrm
            outside = np.dot(nrm, n_or) > 0
            nn = (1.0 / ior) if outside else ior
            cos_ray = np.dot(d, n_or)
            cos2 = 1.0 - nn * nn * (1.0 - cos_ray * cos_ray)
            mirror = d - 2 * np.dot(n_or, d) * n_or
            if cos2 < 0:   # total internal reflection
                assert np.allclose(od, mirror, atol=2e-5) and np.allclose(oo, pnt + 0.05 * n_or, atol=2e-5), i
                assert np.array_equal(om, [1, 1, 1])
                seen["tir"] += 1
            else:
                refl0 = ((ior - 1.0) / (ior + 1.0)) ** 2
                # transmitted direction by Snell: n1 sin(i) = n2 sin(t), in the plane of incidence, on the far side
                trans = unit(d * nn - nrm * ((1.0 if outside else -1.0) * (cos_ray * nn + np.sqrt(cos2))))
                pw = 1.0 - (-cos_ray if outside else np.dot(trans, nrm))
                reflectance = refl0 + (1.0 - refl0) * pw ** 5
                energy = refl + (1.0 - 2.0 * refl) * reflectance
                if u[0] < refl:
                    assert np.allclose(od, d - 2 * np.dot(nrm, d) * nrm, atol=2e-5), i
                    assert np.allclose(om, reflectance / energy, rtol=2e-4), i
                    assert np.allclose(oo, pnt + 0.05 * n_or, atol=2e-5), i
                    seen["refl"] += 1
                else:
                    assert np.allclose(od, trans, atol=2e-5), i
                    sin_i, sin_t = np.linalg.norm(np.cross(d, n_or)), np.linalg.norm(np.cross(od, n_or))
                    assert abs(sin_t - nn * sin_i) < 2e-5, i   # Snell
                    assert np.dot(od, n_or) < 0, i
                    assert np.allclose(om, (1.0 - reflectance) / (1.0 - energy), rtol=2e-4), i
                    assert np.allclose(oo, pnt - 0.0005 * n_or, atol=2e-5), i
                    seen["refr"] += 1
            seen[3] += 1
    assert min(seen[1], seen[2], seen["refr"], seen["refl"]) > 50 and seen["tir"] > 5, seen


def test_oracle_sphere_hits_agree_with_the_quadratic(pt, orc):
    """testSphereIntersection (reference: src/intersect.cuh:59-83) against the quadratic solved in float64: nearest root
    beyond the 1e-4 epsilon, else the far one; point on the sphere, normal radial."""
    desc, cam = pt.load_scene("geoms", width=64, height=48, upload=False)
    o, d = orc.generate_rays(cam, 1, 5)
    checked = 0
    for oi in range(desc.contents.num_objects):
        ob = desc.contents.objects[oi]
        if ob.type != 0:
            continue
        h = orc.trace(desc, o, d, oi)
        c, r = np.array(ob.position[:], np.float64), float(ob.radius)
        for i in range(0, len(o), 3):
            oo, dd = o[i].astype(np.float64), d[i].astype(np.float64)
            op = c - oo
            b = np.dot(op, dd)
            disc = b * b - np.dot(op, op) + r * r
            if disc < -1e-6 * r * r:
                assert h["hit"][i] == 0, (oi, i)
                continue
            if disc < 1e-6 * r * r:
                continue   # tangent within float noise
            t = b - np.sqrt(disc)
            if t < 1e-4:
                t = b + np.sqrt(disc)
            if abs(t - 1e-4) < 1e-5:
                continue
            if t < 1e-4:
                assert h["hit"][i] == 0, (oi, i)
                continue
            assert h["hit"][i] == 1, (oi, i)
            p = oo + t * dd
            assert np.allclose(h["point"][i].view(np.float32), p, atol=2e-3 * max(1.0, t)), (oi, i)
            assert np.allclose(h["normal"][i].view(np.float32), (p - c) / r, atol=2e-3), (oi, i)
            checked += 1
    assert checked > 30

