"""GPU parity: HIP path vs the CPU oracle on the same seeded inputs, through the C ABI.
Bar: bit-exact (traversal, hit ids, leaf visits, ray generation, BSDFs, radiance, accumulation)."""
import ctypes as C

import numpy as np
import pytest

import gpuops

pytestmark = pytest.mark.gpu

SCENES = ["example", "geoms", "teapot", "sky_map", "blender_comp"]


def _secondary_rays(orc_hits, origins, dirs, rng, n):
    """Rays leaving surfaces in random directions (incoherent), plus axis-parallel ones with inf inverses."""
    hit = orc_hits["hit"] == 1
    pts = orc_hits["point"].view(np.float32)[hit]
    idx = rng.integers(0, len(pts), n)
    o = pts[idx] + rng.normal(0, 1e-3, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    axis = np.eye(3, dtype=np.float32)
    d[:6] = np.concatenate([axis, -axis])
    d[6:9, :] = axis * np.float32(1.0)
    d[6:9, :] += np.roll(axis, 1, axis=1) * np.float32(0.5)  # one zero component
    return o.astype(np.float32), d.astype(np.float32)


@pytest.mark.parametrize("name", SCENES)
def test_trace_bit_exact(pt, orc, ctx, name):
    desc, cam = pt.load_scene(name, width=160, height=120, textures=1)
    scene = pt.host().pth_scene_device()
    o, d = orc.generate_rays(cam, -1, 0)
    ho = orc.trace(desc, o, d)
    hg = gpuops.trace(ctx, scene, o, d)
    for k in ho:
        assert np.array_equal(ho[k], hg[k]), f"{name}: primary rays differ in {k}"
    assert ho["hit"].sum() > 0
    rng = np.random.default_rng(1234)
    o2, d2 = _secondary_rays(ho, o, d, rng, 8192)
    ho2 = orc.trace(desc, o2, d2)
    hg2 = gpuops.trace(ctx, scene, o2, d2)
    for k in ho2:
        assert np.array_equal(ho2[k], hg2[k]), f"{name}: secondary rays differ in {k}"
    # per-object queries (rayIntersectsObject)
    for obj in range(desc.contents.num_objects):
        a = orc.trace(desc, o2[:2048], d2[:2048], obj)
        b = gpuops.trace(ctx, scene, o2[:2048], d2[:2048], obj)
        for k in a:
            assert np.array_equal(a[k], b[k]), f"{name}: object {obj} differs in {k}"


@pytest.mark.parametrize("name", SCENES)
def test_generate_rays_bit_exact(pt, orc, ctx, name):
    desc, cam = pt.load_scene(name, width=96, height=64, upload=True)
    for sample in (-1, 0, 7):
        oo, od = orc.generate_rays(cam, sample, 99)
        go, gd = gpuops.generate_rays(ctx, cam, sample, 99)
        assert np.array_equal(oo.view(np.uint32), go.view(np.uint32))
        assert np.array_equal(od.view(np.uint32), gd.view(np.uint32)), f"sample {sample}"


def test_bsdf_bit_exact(pt, orc, ctx):
    rng = np.random.default_rng(7)
    n = 4096
    cases = (pt.PtxBsdfCase * n)()
    for i in range(n):
        c = cases[i]
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        if i % 16 == 0:
            nrm = np.array([0.0, 0.0, 1.0 if i % 32 else -1.0])  # degenerate ONB branch
        if i % 16 == 1:
            d = -nrm  # normal incidence
        c.dir[:] = d.astype(np.float32); c.normal[:] = nrm.astype(np.float32)
        c.point[:] = rng.normal(size=3).astype(np.float32)
        c.mat_type = 1 + i % 3
        c.reflect_factor = float(rng.random()); c.refract_index = float(1.0 + rng.random())
        c.sharpness = float(rng.random())
        c.u[:] = ((rng.integers(0, 1 << 24, 4) + 1) / float(1 << 24)).astype(np.float32)
    ref = (pt.PtxBsdfCase * n)()
    C.memmove(ref, cases, C.sizeof(cases))
    assert orc.lib().orc_bsdf_eval(n, ref) == 0
    pt.check(pt.lib().ptx_bsdf_eval(ctx, n, cases))
    a = np.frombuffer(ref, dtype=np.uint32).reshape(n, -1)
    b = np.frombuffer(cases, dtype=np.uint32).reshape(n, -1)
    bad = np.nonzero((a != b).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} BSDF cases differ, first {bad[:5]}"


@pytest.mark.parametrize("name,tex", [("example", 0), ("geoms", 0), ("teapot", 0), ("sky_map", 1), ("blender_comp", 0),
                                      ("blender_comp_open", 1), ("lamborghini_door", 1)])
def test_radiance_and_accumulation_bit_exact(pt, orc, ctx, name, tex):
    desc, cam = pt.load_scene(name, width=64, height=48, textures=tex)
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    for s in (0, 5):
        ro = orc.radiance_sample(desc, cam, s, 42)
        rg = gpuops.radiance_sample(ctx, scene, cam, s, 42)
        assert np.array_equal(ro.view(np.uint32), rg.view(np.uint32)), f"{name}: radiance of sample {s} differs"
    acc_o, st_o = orc.render(desc, cam, 0, 8, 42)
    acc_g, st_g = gpuops.render(ctx, scene, cam, 0, 8, 42, stats=True)
    assert np.array_equal(acc_o.view(np.uint32), acc_g.view(np.uint32)), f"{name}: accumulation differs"
    # one launch of 8 samples == 8 launches of 1 sample == reference schedule
    acc_g1, _ = gpuops.render(ctx, scene, cam, 0, 8, 42, batch=1)
    assert np.array_equal(acc_g.view(np.uint32), acc_g1.view(np.uint32))
    for k in ("rays", "paths", "n_inner", "n_box", "n_tri", "n_sphere", "n_leaf", "hit_mesh", "hit_sphere", "hit_textured",
              "trav_rays", "trav_inner", "trav_box", "trav_tri", "trav_leaf"):
        assert getattr(st_o, k) == getattr(st_g, k), f"{name}: counter {k}: oracle {getattr(st_o, k)} gpu {getattr(st_g, k)}"
    assert acc_o[:, :3].max() > 0


def test_preview_and_aabb_modes(pt, orc, ctx):
    for kw in ({"preview": 1}, {"aabb": 1}):
        desc, cam = pt.load_scene("teapot", width=64, height=48, **kw)
        scene = pt.host().pth_scene_device()
        gpuops.render_init(ctx, cam)
        acc_o, _ = orc.render(desc, cam, 0, 2, 3)
        acc_g, _ = gpuops.render(ctx, scene, cam, 0, 2, 3)
        assert np.array_equal(acc_o.view(np.uint32), acc_g.view(np.uint32)), kw
        assert acc_o[:, :3].max() > 0


def test_tonemap_matches_oracle(pt, orc, ctx):
    desc, cam = pt.load_scene("geoms", width=64, height=48)
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    acc, _ = gpuops.render(ctx, scene, cam, 0, 4, 1)
    n = cam.width * cam.height
    out_g = np.empty((n, 4), np.uint8)
    pt.check(pt.lib().ptx_tonemap(ctx, 4, 0.5, out_g.ctypes.data_as(C.POINTER(C.c_ubyte))))
    out_o = np.empty((n, 4), np.uint8)
    assert orc.lib().orc_tonemap(orc.fptr(acc), n, 4, 0.5, out_o.ctypes.data_as(C.POINTER(C.c_ubyte))) == 0
    # powf differs by at most 1 ulp between libm and the device library: allow 1 LSB on truncation
    assert np.abs(out_g.astype(int) - out_o.astype(int)).max() <= 1
    assert (out_g[:, 3] == 0).all()


def test_full_size_lamborghini_matches_oracle(pt, orc, ctx):
    """The bench workload itself (1920x1080, ~400k triangles), 1 spp, every traversal variant: accumulation
    buffer and work counters bit-identical to the oracle."""
    desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1)
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    acc_o, st_o = orc.render(desc, cam, 3, 1, 7, nthreads=16)
    try:
        for variant in (0, 7, 9, 13, 14, 15):
            pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", variant))
            for stats in (True, False):
                acc_g, st_g = gpuops.render(ctx, scene, cam, 3, 1, 7, stats=stats)
                diff = np.nonzero((acc_o.view(np.uint32) != acc_g.view(np.uint32)).any(axis=1))[0]
                assert len(diff) == 0, f"variant {variant} stats {stats}: {len(diff)} pixels differ, first {diff[:5]}"
                # (variant 14 counts what the WIDE kernel executes, PtxStats.x_*, not the reference's node and triangle counters)
                keys = ("rays", "paths") if not stats or variant == 14 else ("rays", "paths", "n_inner", "n_box", "n_tri", "n_sphere", "n_leaf", "hit_mesh", "hit_sphere",
                                                            "hit_textured", "trav_rays", "trav_inner", "trav_box", "trav_tri", "trav_leaf")
                for k in keys:
                    assert getattr(st_o, k) == getattr(st_g, k), f"variant {variant}: counter {k}: oracle {getattr(st_o, k)} gpu {getattr(st_g, k)}"
    finally:
        pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", -1))
    # and a longer run of the default schedule: 24 samples in one pass (about 105 M path segments)
    acc_o8, st_o8 = orc.render(desc, cam, 10, 24, 7, nthreads=16)
    acc_g8, st_g8 = gpuops.render(ctx, scene, cam, 10, 24, 7)
    diff = np.nonzero((acc_o8.view(np.uint32) != acc_g8.view(np.uint32)).any(axis=1))[0]
    assert len(diff) == 0, f"24 samples: {len(diff)} pixels differ, first {diff[:5]}"
    assert st_o8.rays == st_g8.rays and st_o8.paths == st_g8.paths


def test_fast_reciprocal_is_exact(pt, ctx):
    """The triangle test replaces 1.0f / det by v_rcp_f32 + two fma Newton steps. That is only legitimate if the
    result is the IEEE quotient bit for bit, so the device compares EVERY float in the admitted range
    (|det| from the rejection threshold 5e-8 up to 2^126): 2.5 billion values, no mismatch allowed."""
    c, m, f = C.c_uint64(), C.c_uint64(), C.c_uint32()
    pt.check(pt.lib().ptx_selftest_rcp(ctx, 5.0e-8, 2.0 ** 126, C.byref(c), C.byref(m), C.byref(f)))
    assert c.value == 2521989336 and m.value == 0, f"{m.value} of {c.value} reciprocals differ, first bits {f.value:#010x}"
    # outside the range the shortcut is wrong, which is why pt_rcp_det falls back to the division there
    pt.check(pt.lib().ptx_selftest_rcp(ctx, 2.0 ** 126 * 1.0001, 3.0e38, C.byref(c), C.byref(m), C.byref(f)))
    assert m.value > 0


@pytest.mark.parametrize("name", ["teapot", "geoms"])
def test_degenerate_rays_bit_exact(pt, orc, ctx, name):
    """Rays the render path never makes but the entry point accepts: zero and axis-parallel directions
    (inverse = +-inf, 0 * inf = NaN in the slab test), unnormalised tiny / huge directions, origins at 1e30,
    NaN origins, infinite directions, origins on box planes. Every field must equal the oracle's."""
    desc, cam = pt.load_scene(name, width=32, height=24)
    scene = pt.host().pth_scene_device()
    rng = np.random.default_rng(4)
    n = 4096
    o = rng.normal(0, 3, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[0:64] = 0
    d[64:128, 0] = 0; d[128:192, 1] = 0; d[192:256, 2] = 0; d[256:320, :2] = 0
    d[320:384] *= 1e-20; d[384:448] *= 1e18
    o[448:512] = 1e30; o[512:520] = np.nan; d[520:528] = np.inf
    o[528:600] = 0
    ho, hg = orc.trace(desc, o, d), gpuops.trace(ctx, scene, o, d)
    for k in ho:
        bad = np.nonzero((ho[k] != hg[k]).reshape(n, -1).any(axis=1))[0]
        assert len(bad) == 0, f"{name}: {k} differs for {len(bad)} rays, first {bad[:8]}"
    assert ((ho["hit"] == 1) & (ho["object"] == 0xFFFFFFFF)).any()   # the hit-without-object case is covered


def _mesh_rays(desc, cam, orc, obj, rng, n_rand=6000):
    """Adversarial rays for one mesh: primary rays, random rays through its box, rays aimed exactly AT triangle
    vertices and edge midpoints (several triangles share them: equal-t ties, group-box faces), axis-parallel
    rays through vertices (inverse direction +-inf), rays starting on the surface, degenerate directions."""
    import ctypes
    d = desc.contents
    ob = d.objects[obj]
    tris = np.ctypeslib.as_array(ctypes.cast(d.triangles, ctypes.POINTER(pt_mod.PtxTriangle * d.num_triangles)).contents)
    tris = tris[ob.first_triangle: ob.first_triangle + ob.num_triangles]
    a = tris["a"].astype(np.float32); b = (a + tris["e1"]).astype(np.float32); c = (a + tris["e2"]).astype(np.float32)
    lo = np.minimum(np.minimum(a, b), c).min(axis=0); hi = np.maximum(np.maximum(a, b), c).max(axis=0)
    ext = hi - lo
    O, D = [], []
    o, dd = orc.generate_rays(cam, -1, 0)
    O.append(o); D.append(dd)
    # random rays through the box
    p0 = (lo - ext + rng.random((n_rand, 3)) * 3 * ext).astype(np.float32)
    p1 = (lo + rng.random((n_rand, 3)) * ext).astype(np.float32)
    dr = p1 - p0; dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    O.append(p0); D.append(dr.astype(np.float32))
    # rays aimed at vertices and edge midpoints from random origins
    t = rng.integers(0, len(a), 3000)
    tgt = np.concatenate([a[t[:1000]], (0.5 * (a[t[1000:2000]] + b[t[1000:2000]])).astype(np.float32), c[t[2000:]]])
    p0 = (lo - 0.5 * ext + rng.random((3000, 3)) * 2 * ext).astype(np.float32)
    dr = tgt - p0; dr /= np.maximum(np.linalg.norm(dr, axis=1, keepdims=True), 1e-30)
    O.append(p0); D.append(dr.astype(np.float32))
    # axis-parallel rays through vertices
    for ax in range(3):
        for sgn in (1.0, -1.0):
            v = a[rng.integers(0, len(a), 400)].copy()
            dirv = np.zeros((400, 3), np.float32); dirv[:, ax] = sgn
            v[:, ax] = lo[ax] - ext[ax] if sgn > 0 else hi[ax] + ext[ax]
            O.append(v); D.append(dirv)
    # rays leaving the surface (secondary-ray-like), incl. grazing directions in the triangle's plane
    t = rng.integers(0, len(a), 3000)
    w = rng.random((3000, 2)).astype(np.float32); w[w.sum(axis=1) > 1] = 1 - w[w.sum(axis=1) > 1]
    p = (a[t] + w[:, :1] * tris["e1"][t] + w[:, 1:] * tris["e2"][t]).astype(np.float32)
    dr = rng.normal(size=(3000, 3)).astype(np.float32)
    dr[:500] = tris["e1"][t[:500]]
    dr /= np.maximum(np.linalg.norm(dr, axis=1, keepdims=True), 1e-30)
    O.append(p); D.append(dr.astype(np.float32))
    # degenerate
    p0 = rng.normal(0, 3, (256, 3)).astype(np.float32); dr = rng.normal(0, 1, (256, 3)).astype(np.float32)
    dr[0:32] = 0; dr[32:64] *= 1e-20; dr[64:96] *= 1e18; p0[96:112] = np.nan; dr[112:128] = np.inf; p0[128:160] = 1e30
    O.append(p0); D.append(dr)
    return np.concatenate(O).astype(np.float32), np.concatenate(D).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["teapot", "lamborghini", "blender_comp", "blender_comp_open", "lamborghini_door"])
def test_production_traversal_kernel_ray_by_ray(pt, orc, ctx, name):
    """ptx_trace_rays_queued: the wavefront pipeline's own traversal kernel (every leaf mode) against the oracle's
    rayIntersectsObject, ray by ray, on rays chosen to sit on triangle vertices / edges / box faces."""
    global pt_mod
    pt_mod = pt
    desc, cam = pt.load_scene(name, width=64, height=48)
    scene = pt.host().pth_scene_device()
    d = desc.contents
    meshes = [i for i in range(d.num_objects) if d.objects[i].num_nodes > 1]
    assert meshes
    rng = np.random.default_rng(11)
    try:
        for obj in meshes[:3]:
            o, dd = _mesh_rays(desc, cam, orc, obj, rng)
            ho = orc.trace(desc, o, dd, obj)
            assert (ho["object"] == obj).sum() > 1000
            for variant in (0, 7, 9, 13, 14, 15):
                pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", variant))
                hg = gpuops.trace_queued(ctx, scene, o, dd, obj)
                for k in ("object", "prim", "t_bits", "bary"):
                    bad = np.nonzero((ho[k] != hg[k]).reshape(len(o), -1).any(axis=1))[0]
                    assert len(bad) == 0, (f"{name} object {obj} variant {variant}: {k} differs for {len(bad)} of {len(o)} rays, "
                                           f"first {bad[:6]}: oracle {ho[k][bad[:3]]} gpu {hg[k][bad[:3]]}")
    finally:
        pt.check(pt.lib().ptx_ctx_tune(ctx, b"trav_variant", -1))


@pytest.mark.gpu
def test_group_box_culling_under_grazing_rays(pt, orc, ctx, tmp_path):
    """Leaf modes 5/6 skip groups of triangles whose padded box the ray misses. The regime where that could drop a
    hit the reference reports is the reference's own triangle test accepting a ray just OUTSIDE a triangle, at
    grazing incidence: rays aimed there (tools/grazing_stress.py) must still equal the oracle, with the default
    padding and with none beyond the absolute 5e-5."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("grazing_stress", os.path.join(os.path.dirname(__file__), "..", "tools", "grazing_stress.py"))
    gs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gs)
    rng = np.random.default_rng(9)
    gs.write_soup(str(tmp_path / "soup.obj"), 20000, rng, offset=300.0)
    (tmp_path / "s.json").write_text(
        '{"camera": {"pixelWidth": 64, "pixelHeight": 48, "direction": [0,0,-1], "up": [0,1,0], "position": [300,300,320],'
        ' "screenDist": 1.5, "screenHeight": 2}, "objects": [{"type": "mesh", "render": true, "src": "%s"}]}' % (tmp_path / "soup.obj"))
    lib = pt.lib()
    try:
        o = dd = ho = None
        for ppm in (-1, 0):
            pt.check(lib.ptx_ctx_tune(ctx, b"grp_pad_ppm", ppm))
            desc, cam = pt.load_scene(str(tmp_path / "s.json"))
            scene = pt.host().pth_scene_device()
            if o is None:
                o, dd, tgt = gs.grazing_rays(desc, 0, 150000, rng)
                ho = orc.trace(desc, o, dd, 0)
                assert ((ho["object"] == 0) & (ho["prim"] == tgt)).sum() > 10000   # the rays do reach the triangles aimed at
            for variant in (0, 7, 9, 13, 15):
                pt.check(lib.ptx_ctx_tune(ctx, b"trav_variant", variant))
                hg = gpuops.trace_queued(ctx, scene, o, dd, 0)
                for k in ("object", "prim", "t_bits", "bary"):
                    bad = np.nonzero((ho[k] != hg[k]).reshape(len(o), -1).any(axis=1))[0]
                    assert len(bad) == 0, f"padding {ppm} ppm, variant {variant}: {k} differs for {len(bad)} rays, first {bad[:5]}"
    finally:
        pt.check(lib.ptx_ctx_tune(ctx, b"grp_pad_ppm", -1))
        pt.check(lib.ptx_ctx_tune(ctx, b"trav_variant", -1))


_BOX_FACES = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]


@pytest.mark.parametrize("direction,up,zero_axis", [((0, 0, -1), (0, 1, 0), 0), ((0, 0, -1), (1, 0, 0), 1), ((-1, 0, 0), (0, 1, 0), 2)])
def test_rays_with_a_zero_direction_component_through_single_leaf_meshes(pt, orc, ctx, tmp_path, direction, up, zero_axis):
    """A camera whose centre column looks exactly along a coordinate plane (65 x 64 pixels of 1/32: the column's offset
    is 0 in floating point; no jitter): those rays have a direction component of exactly 0, an infinite inverse, and
    every box test on their way must treat the axis as 'inside or outside for good'. The single-leaf meshes (a closed
    luminous box around everything, a floor) are culled by pair boxes in the segment kernels; until late round 3 that
    test formed plane * inf - o * inf = NaN and let such rays through the whole box: one black pixel in ten samples of
    the headline frame (tools/strict_diff.py). Every schedule must equal the oracle, pixel by pixel."""
    c = np.array([[x, y, z] for x in (-40.0, 47.0) for y in (-3.0, 51.0) for z in (-44.0, 39.0)])
    lines = ["mtllib m.mtl", "usemtl sky.lumi"] + ["v %.9g %.9g %.9g" % tuple(q) for q in c]
    for f in _BOX_FACES:
        lines += ["f %d %d %d" % (f[0] + 1, f[1] + 1, f[2] + 1), "f %d %d %d" % (f[0] + 1, f[2] + 1, f[3] + 1)]
    (tmp_path / "box.obj").write_text("\n".join(lines) + "\n")
    (tmp_path / "floor.obj").write_text("mtllib m.mtl\nusemtl ground.diff\nv -20 0 -20\nv 20 0 -20\nv 20 0 20\nv -20 0 20\nf 1 3 2\nf 1 4 3\n")
    (tmp_path / "m.mtl").write_text("newmtl sky.lumi\nKd 1 1 1\nKe 0.9 0.8 0.7\nnewmtl ground.diff\nKd 0.7 0.6 0.5\n")
    pos = [3.25, 2.5, 5.75]
    # a tessellated ball in front of the camera (3120 triangles: a multi-node mesh, so the column's rays also walk the wide
    # tree and the strict kernel with an infinite inverse), centred ON the camera axis
    centre = np.array(pos) + 8.0 * np.array(direction, float)
    nu, nv = 40, 40
    th, ph = np.meshgrid(np.linspace(0.0, np.pi, nv + 1), np.linspace(0.0, 2.0 * np.pi, nu + 1), indexing="ij")
    pts = centre + 2.5 * np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], axis=-1)
    lines = ["v %.9g %.9g %.9g" % tuple(q) for q in pts.reshape(-1, 3)]
    idx = lambda i, j: i * (nu + 1) + j + 1
    for i in range(nv):
        for j in range(nu):
            if i > 0:
                lines.append("f %d %d %d" % (idx(i, j), idx(i, j + 1), idx(i + 1, j)))   # outward normals
            if i < nv - 1:
                lines.append("f %d %d %d" % (idx(i, j + 1), idx(i + 1, j + 1), idx(i + 1, j)))
    (tmp_path / "ball.obj").write_text("\n".join(lines) + "\n")
    (tmp_path / "s.json").write_text(
        '{"camera": {"pixelWidth": 65, "pixelHeight": 64, "direction": [%d,%d,%d], "up": [%d,%d,%d], "position": [%g,%g,%g],'
        ' "screenDist": 1.5, "screenHeight": 2}, "objects": ['
        '{"type": "sphere", "render": true, "radius": 1.0, "position": [0, 1, 0], "material": {"type": "spec", "Kd": [0.9], "Ns": [900], "d": [0.2]}},'
        '{"type": "mesh", "render": true, "src": "%s"}, {"type": "mesh", "render": true, "src": "%s"}, {"type": "mesh", "render": true, "src": "%s"}]}'
        % (*direction, *up, *pos, tmp_path / "box.obj", tmp_path / "floor.obj", tmp_path / "ball.obj"))
    desc, cam = pt.load_scene(str(tmp_path / "s.json"))
    assert desc.contents.num_nodes > 10, "the ball is meant to be a multi-node mesh"
    cam.aa_jitter = 0.0
    scene = pt.host().pth_scene_device()
    o, d = orc.generate_rays(cam, 0, 0)
    col = d.reshape(64, 65, 3)[:, 32, :]
    assert (col[:, zero_axis] == 0.0).all(), "the centre column's rays are meant to have an exactly zero component"
    gpuops.render_init(ctx, cam)
    acc_o, st_o = orc.render(desc, cam, 0, 8, 3)
    assert (acc_o[:, :3].sum(axis=1) > 0).mean() > 0.98, "closed luminous box: hardly a pixel of the oracle stays dark"
    try:
        for key, val in ((b"trav_variant", 0), (b"trav_variant", 15), (b"trav_variant", 13), (b"seg_mode", 0), (b"seg_compact", 0), (b"megakernel", 1)):
            pt.check(pt.lib().ptx_ctx_tune(ctx, key, val))
            acc_g, st_g = gpuops.render(ctx, scene, cam, 0, 8, 3)
            pt.check(pt.lib().ptx_ctx_tune(ctx, key, -1))
            bad = np.nonzero((acc_o.view(np.uint32) != acc_g.view(np.uint32)).any(axis=1))[0]
            assert len(bad) == 0, f"{key.decode()}={val}: {len(bad)} pixels differ, first {bad[:5]} (x = {bad[:5] % 65})"
            assert st_o.rays == st_g.rays
    finally:
        for key in (b"trav_variant", b"seg_mode", b"seg_compact", b"megakernel"):
            pt.check(pt.lib().ptx_ctx_tune(ctx, key, -1))
