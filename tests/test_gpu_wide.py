"""The wide traversal (csrc/pt_wide.h, the default since round 2) and its exits to the strict kernel.

The wide kernel walks its own 4-wide tree and returns the minimum t over all triangles; which of two triangles
with the SAME t wins is the reference's visiting order, which only the reference-order kernel (wf_traverse_coop_kernel; strict leaf loops on the retry queue) knows. Rays
in that situation, and rays whose stack is full, go through the retry queue. These tests force both exits and
compare with the oracle / the strict kernel bit for bit; the counters of the wide kernel's own counting twin
(variant 14) prove that the exits were taken. All through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import gpuops

pytestmark = pytest.mark.gpu


def _tune(pt, ctx, key, value):
    pt.check(pt.lib().ptx_ctx_tune(ctx, key, value))


def test_equal_t_from_two_blocks_goes_to_the_exact_kernel(pt, orc, ctx, tmp_path):
    """100 identical triangles form one leaf of four 32-slot blocks: every hit has the same t in all of them, the
    wide kernel cannot know which the reference tests first, and the strict kernel must decide (the first)."""
    # ONE mesh: a bumpy floor that makes the builder split, and 100 copies of one triangle above it (equal centroids:
    # they stay together in a leaf that spans several blocks)
    g = 12
    lines = ["mtllib d.mtl", "usemtl white.diff"]
    for z in range(g + 1):
        for x in range(g + 1):
            lines.append(f"v {x - g / 2} {0.05 * ((x * 7 + z * 3) % 5)} {z - g / 2}")
    vid = lambda x, z: z * (g + 1) + x + 1
    for z in range(g):
        for x in range(g):
            lines.append(f"f {vid(x, z)} {vid(x + 1, z + 1)} {vid(x + 1, z)}")
            lines.append(f"f {vid(x, z)} {vid(x, z + 1)} {vid(x + 1, z + 1)}")
    nv = (g + 1) * (g + 1)
    lines += ["usemtl mirror.spec", "v -3 0.5 -3", "v 3 0.5 -3", "v 0 4 -3"] + [f"f {nv + 1} {nv + 2} {nv + 3}"] * 100
    (tmp_path / "both.obj").write_text("\n".join(lines) + "\n")
    (tmp_path / "d.mtl").write_text("newmtl white.diff\nKd 0.8 0.8 0.8\nnewmtl mirror.spec\nKd 0.9 0.9 0.9\nNs 900\nd 0.8\n")
    (tmp_path / "s.json").write_text(
        '{"camera": {"pixelWidth": 96, "pixelHeight": 64, "direction": [0,-0.5,-1], "up": [0,1,0], "position": [0,6,12],'
        ' "screenDist": 1.5, "screenHeight": 2}, "objects": ['
        '{"type": "sphere", "render": true, "radius": 4, "position": [0, 12, 0], "material": {"type": "lumi", "Kd": [1], "Ke": [6]}},'
        '{"type": "mesh", "render": true, "src": "%s"}]}' % (tmp_path / "both.obj"))
    desc, cam = pt.load_scene(str(tmp_path / "s.json"))
    d0 = desc.contents
    assert d0.num_nodes > 3 and max(d0.nodes[i].num_tris for i in range(d0.num_nodes)) >= 100
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    acc_o, st_o = orc.render(desc, cam, 0, 4, 6)
    try:
        _tune(pt, ctx, b"trav_variant", 14)
        acc_g, st_g = gpuops.render(ctx, scene, cam, 0, 4, 6, stats=True)
        d = st_g.as_dict()
        assert np.array_equal(acc_o.view(np.uint32), acc_g.view(np.uint32))
        assert d["x_half_lanes"] > 50, "no ray went through the retry queue"      # rays retried
        assert d["diag3"] == 0                                                    # none of them for a full stack
        _tune(pt, ctx, b"trav_variant", 0)
        acc_f, _ = gpuops.render(ctx, scene, cam, 0, 4, 6)
        assert np.array_equal(acc_o.view(np.uint32), acc_f.view(np.uint32))
    finally:
        _tune(pt, ctx, b"trav_variant", -1)
    # ray by ray through the production kernel: the hits name the FIRST of the identical triangles
    o, dd = orc.generate_rays(cam, -1, 0)
    ho, hq = orc.trace(desc, o, dd, 1), gpuops.trace_queued(ctx, scene, o, dd, 1)
    first_copy = 2 * g * g
    on_copies = (ho["hit"] == 1) & (ho["prim"].astype(np.int64) >= first_copy) & (ho["prim"] != 0xFFFFFFFF)
    assert on_copies.sum() > 50 and set(ho["prim"][on_copies]) == {first_copy}
    for k in ("object", "prim", "t_bits", "bary"):
        assert np.array_equal(ho[k], hq[k]), k


@pytest.mark.parametrize("scap", [1, 2, 3])
def test_full_stack_goes_to_the_exact_kernel(pt, orc, ctx, scap):
    """With room for only `scap` stack entries most rays of the car give up and are traced again by the exact
    kernel: the image must not change by a bit, and the counters must show the exit."""
    desc, cam = pt.load_scene("lamborghini", width=160, height=90, textures=1)
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    acc_o, st_o = orc.render(desc, cam, 0, 2, 5)
    try:
        _tune(pt, ctx, b"wide_scap", scap)
        _tune(pt, ctx, b"trav_variant", 14)
        acc_g, st_g = gpuops.render(ctx, scene, cam, 0, 2, 5, stats=True)
        d = st_g.as_dict()
        assert np.array_equal(acc_o.view(np.uint32), acc_g.view(np.uint32))
        assert d["diag3"] > 100 and d["x_half_lanes"] >= d["diag3"], (d["diag3"], d["x_half_lanes"])
        assert d["rays"] == st_o.rays and d["trav_rays"] == st_o.trav_rays
        _tune(pt, ctx, b"trav_variant", 0)
        acc_f, _ = gpuops.render(ctx, scene, cam, 0, 2, 5)
        assert np.array_equal(acc_o.view(np.uint32), acc_f.view(np.uint32))
    finally:
        _tune(pt, ctx, b"wide_scap", -1)
        _tune(pt, ctx, b"trav_variant", -1)


def test_wide_and_exact_kernels_agree_on_every_scene(pt, ctx):
    """Variant 0 (wide + retry), 14 (the same with its own counting twin) and 15 (the strict kernel alone) give the
    same accumulation buffer on every shipped scene with a multi-node mesh; the default stack never overflows on
    the car."""
    for name in ("lamborghini", "teapot"):
        desc, cam = pt.load_scene(name, width=192, height=108, textures=1)
        scene = pt.host().pth_scene_device()
        gpuops.render_init(ctx, cam)
        try:
            _tune(pt, ctx, b"trav_variant", 15)
            ref, _ = gpuops.render(ctx, scene, cam, 5, 3, 11)
            for variant, stats in ((0, False), (0, True), (14, False), (14, True)):
                _tune(pt, ctx, b"trav_variant", variant)
                acc, st = gpuops.render(ctx, scene, cam, 5, 3, 11, stats=stats)
                assert np.array_equal(ref.view(np.uint32), acc.view(np.uint32)), (name, variant, stats)
                if variant == 14 and stats:
                    d = st.as_dict()
                    assert d["x_item_lanes"] > 0 and d["x_units"] > 0
                    assert d["diag3"] == 0, f"{name}: {d['diag3']} rays overflowed the default stack"
        finally:
            _tune(pt, ctx, b"trav_variant", -1)


def test_fragile_hits_at_grazing_incidence_equal_the_oracle(pt, orc, ctx):
    """At cos(incidence) ~ 1e-4 the reference's t = f * dot(e2, q) is off by 0.005 ... 0.25 units: a triangle is "hit"
    in front of its own boxes and beats a well-conditioned hit behind which its box lies. 400 000 grazing rays on the
    car (tools/grazing_stress.py, the set on which the first wide kernel differed from the oracle in rays 104296,
    134791 and 302464): with the culling margin and the robust-hit check every ray equals the oracle, through the
    wide kernel, its counting twin's kernel and the strict kernel alone."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("grazing_stress", os.path.join(os.path.dirname(__file__), "..", "tools", "grazing_stress.py"))
    gs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gs)
    desc, cam = pt.load_scene("lamborghini", 64, 48)
    scene = pt.host().pth_scene_device()
    d = desc.contents
    obj = max(range(d.num_objects), key=lambda i: d.objects[i].num_nodes)
    rng = np.random.default_rng(5)
    o, dd, tgt = gs.grazing_rays(desc, obj, 400000, rng)
    ho = orc.trace(desc, o, dd, obj)
    assert (ho["object"] == obj).sum() > 300000
    # the three rays are in the set, and what makes them special is visible in the oracle's own result: its t lies
    # in FRONT of the triangle it names (double-precision t of that triangle is larger by more than 1e-3)
    tris = np.ctypeslib.as_array(C.cast(d.triangles, C.POINTER(pt.PtxTriangle * d.num_triangles)).contents)
    tris = tris[d.objects[obj].first_triangle: d.objects[obj].first_triangle + d.objects[obj].num_triangles]
    for i in (104296, 134791, 302464):
        t = tris[int(ho["prim"][i])]
        a, e1, e2 = (t[k].astype(np.float64) for k in ("a", "e1", "e2"))
        dv, ov = dd[i].astype(np.float64), o[i].astype(np.float64)
        t64 = (e2 @ np.cross(ov - a, e1)) / (e1 @ np.cross(dv, e2))
        assert t64 - float(ho["t_bits"][i:i + 1].view(np.float32)[0]) > 1e-3, i
    try:
        for variant in (0, 13, 14, 15):
            _tune(pt, ctx, b"trav_variant", variant)
            hg = gpuops.trace_queued(ctx, scene, o, dd, obj)
            for k in ("object", "prim", "t_bits", "bary"):
                bad = np.nonzero((ho[k] != hg[k]).reshape(len(o), -1).any(axis=1))[0]
                assert len(bad) == 0, f"variant {variant}: {k} differs for {len(bad)} rays, first {bad[:5]}"
    finally:
        _tune(pt, ctx, b"trav_variant", -1)


def _load_grazing_tool():
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("grazing_stress", os.path.join(os.path.dirname(__file__), "..", "tools", "grazing_stress.py"))
    gs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gs)
    return gs


# What the DEFAULT traversal (variant 0: wide tree over groups of eight triangles, distance culling with a margin) and the
# grouped reference-order kernel (variant 13: the reference's visiting order, but only the touched groups of a leaf) are known
# not to reproduce: hits the reference's triangle test accepts out of pure rounding noise (|det| ~ 1e-7 |e1||e2|: t, u, v carry
# no digits) on a triangle whose group box the ray misses, or enters beyond a well-conditioned hit. 2 000 000 adversarial
# grazing rays per scene hold 1 such ray on the car and 2 on the soup; ten such sets, 7 rays
# (tools/grazing_rate.py; they are the fixture of the test after this one). The ceilings below fence that regime: the STRICT
# kernel (variant 15, what PTX_TRAV_VARIANT=15 / ptx_ctx_tune selects: every triangle of every visited leaf) must differ in
# NONE, the other two in at most these, and every ray they differ in must be such a noise hit of the oracle (its t more than
# 1e-3 away from the same triangle's t in double).
GRAZING_CEILING = {"lamborghini": 2, "soup": 3}


def _is_noise_hit(tri, o, d, t_bits):
    """The float32 test accepted this triangle, but in double the ray does NOT hit it where float says: t off by more than
    1e-3, or the barycentrics outside the triangle by more than 1e-4 (the ray passes beside it; u, v carry no digits)."""
    a, e1, e2 = (tri[k].astype(np.float64) for k in ("a", "e1", "e2"))
    dv, ov = d.astype(np.float64), o.astype(np.float64)
    h = np.cross(dv, e2)
    det = e1 @ h
    s = ov - a
    u = (s @ h) / det
    q = np.cross(s, e1)
    v = (dv @ q) / det
    t64 = (e2 @ q) / det
    t32 = float(t_bits.view(np.float32)[0])
    return abs(t64 - t32) > 1e-3 or u < -1e-4 or v < -1e-4 or u + v > 1.0 + 1e-4


@pytest.mark.parametrize("which", ["lamborghini", "soup"])
def test_two_million_grazing_rays_exact_kernel_equals_oracle_and_default_stays_under_its_ceiling(pt, orc, ctx, tmp_path, which):
    gs = _load_grazing_tool()
    rng = np.random.default_rng(5)
    if which == "soup":
        gs.write_soup(str(tmp_path / "soup.obj"), 20000, rng)
        (tmp_path / "s.json").write_text(
            '{"camera": {"pixelWidth": 64, "pixelHeight": 48, "direction": [0,0,-1], "up": [0,1,0], "position": [0,0,20],'
            ' "screenDist": 1.5, "screenHeight": 2}, "objects": [{"type": "mesh", "render": true, "src": "%s"}]}' % (tmp_path / "soup.obj"))
        name = str(tmp_path / "s.json")
    else:
        name = "lamborghini"
    desc, cam = pt.load_scene(name, 64, 48)
    scene = pt.host().pth_scene_device()
    d = desc.contents
    obj = max(range(d.num_objects), key=lambda i: d.objects[i].num_nodes)
    o, dd, tgt = gs.grazing_rays(desc, obj, 2000000, rng)
    ho = orc.trace(desc, o, dd, obj)
    assert (ho["object"] == obj).sum() > 1500000
    tris = np.ctypeslib.as_array(C.cast(d.triangles, C.POINTER(pt.PtxTriangle * d.num_triangles)).contents)
    tris = tris[d.objects[obj].first_triangle: d.objects[obj].first_triangle + d.objects[obj].num_triangles]
    try:
        differ = {}
        for variant in (15, 13, 0):
            _tune(pt, ctx, b"trav_variant", variant)
            hg = gpuops.trace_queued(ctx, scene, o, dd, obj)
            bad = np.zeros(len(o), bool)
            for k in ("object", "prim", "t_bits", "bary"):
                bad |= (ho[k] != hg[k]).reshape(len(o), -1).any(axis=1)
            differ[variant] = np.nonzero(bad)[0]
        assert len(differ[15]) == 0, f"strict kernel differs from the oracle in {len(differ[15])} rays, first {differ[15][:5]}"
        print(f"\n{which}: of {len(o)} grazing rays the default traversal differs from the oracle in {differ[0].tolist()}, the grouped kernel in {differ[13].tolist()}")
        for v in (0, 13):
            assert len(differ[v]) <= GRAZING_CEILING[which], f"variant {v}: {len(differ[v])} rays differ (ceiling {GRAZING_CEILING[which]}): {differ[v][:10]}"
        for i in sorted(set(differ[0].tolist()) | set(differ[13].tolist())):
            assert _is_noise_hit(tris[int(ho["prim"][i])], o[i], dd[i], ho["t_bits"][i:i + 1]), \
                f"ray {i}: the oracle's hit is well conditioned and a grouped traversal misses it"
    finally:
        _tune(pt, ctx, b"trav_variant", -1)


def test_seven_hard_rays_of_twenty_million_strict_kernel_equals_the_oracle(pt, orc, ctx):
    """tests/golden/grazing_hard_rays.npz: the 7 rays of 19 654 555 adversarial grazing rays on the headline mesh (ten sets of
    tools/grazing_rate.py) on which the default traversal and the grouped kernel differ from the oracle, with the oracle's
    answers. The strict kernel (variant 15) must give exactly those; so must the render path's ptx_trace_rays; the other two
    may differ, but only where the oracle's hit is rounding noise (checked against the triangle in double)."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "grazing_hard_rays.npz"))
    o, dd = fx["o"], fx["d"]
    desc, cam = pt.load_scene("lamborghini", 64, 48)
    scene = pt.host().pth_scene_device()
    d = desc.contents
    obj = max(range(d.num_objects), key=lambda i: d.objects[i].num_nodes)
    ho = orc.trace(desc, o, dd, obj)
    for k in ("hit", "object", "prim", "t_bits", "bary"):
        assert np.array_equal(ho[k], fx[k]), f"the oracle no longer reproduces the fixture: {k}"
    tris = np.ctypeslib.as_array(C.cast(d.triangles, C.POINTER(pt.PtxTriangle * d.num_triangles)).contents)
    tris = tris[d.objects[obj].first_triangle: d.objects[obj].first_triangle + d.objects[obj].num_triangles]
    try:
        hp = gpuops.trace(ctx, scene, o, dd, obj)
        for k in ("hit", "object", "prim", "t_bits", "bary"):
            assert np.array_equal(hp[k], fx[k]), f"ptx_trace_rays: {k}"
        for variant in (15, 7, 9, 13, 0):
            _tune(pt, ctx, b"trav_variant", variant)
            hg = gpuops.trace_queued(ctx, scene, o, dd, obj)
            bad = np.zeros(len(o), bool)
            for k in ("object", "prim", "t_bits", "bary"):
                bad |= (fx[k] != hg[k]).reshape(len(o), -1).any(axis=1)
            if variant in (15, 7, 9):
                assert not bad.any(), f"strict kernel (variant {variant}) differs on hard rays {np.nonzero(bad)[0]}"
            for i in np.nonzero(bad)[0]:
                assert _is_noise_hit(tris[int(fx["prim"][i])], o[i], dd[i], fx["t_bits"][i:i + 1]), \
                    f"variant {variant} ray {i}: the oracle's hit is well conditioned and a grouped traversal misses it"
    finally:
        _tune(pt, ctx, b"trav_variant", -1)


def test_production_rays_default_and_exact_traversal_accumulate_identically(pt, ctx):
    """The production distribution: the headline frame (1920x1080, textures on, 5 bounces), 256 samples per pixel = 1.1e9
    path segments, 340 M of them traversals of the car's tree, through the default traversal and through the strict kernel
    alone (variant 15 = the reference's visiting order, every triangle of every visited leaf): the accumulation buffers are bit-identical. (The same comparison at
    1000 spp, 4.4e9 segments, is tools/production_parity.py; its result is recorded in DESIGN.md 3.2.)"""
    desc, cam = pt.load_scene("lamborghini", width=1920, height=1080, textures=1)
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    try:
        _tune(pt, ctx, b"trav_variant", 15)
        ref, st_e = gpuops.render(ctx, scene, cam, 0, 256, 7, batch=128)
        _tune(pt, ctx, b"trav_variant", 0)
        acc, st_w = gpuops.render(ctx, scene, cam, 0, 256, 7, batch=128)
        assert st_e.rays == st_w.rays > 1.0e9
        bad = np.nonzero((ref.view(np.uint32) != acc.view(np.uint32)).any(axis=1))[0]
        assert len(bad) == 0, f"{len(bad)} pixels differ, first {bad[:5]}"
        # and the one-lane-per-pixel megakernel (the reference's loops as written: no wide tree, no group or pair boxes, no
        # LDS pool), which shares no culling structure with either: the pair-box bug of rounds 2-3 showed
        # as 16 missing segments here
        _tune(pt, ctx, b"trav_variant", -1)
        _tune(pt, ctx, b"megakernel", 1)
        mk, st_m = gpuops.render(ctx, scene, cam, 0, 256, 7, batch=128)
        assert st_m.rays == st_w.rays
        bad = np.nonzero((mk.view(np.uint32) != acc.view(np.uint32)).any(axis=1))[0]
        assert len(bad) == 0, f"megakernel: {len(bad)} pixels differ, first {bad[:5]}"
    finally:
        _tune(pt, ctx, b"trav_variant", -1)
        _tune(pt, ctx, b"megakernel", -1)


@pytest.mark.parametrize("grp_mode,wide_build", [(1, 0), (0, 0), (1, 1), (1, 2)])
def test_unevenly_tessellated_mesh_matches_the_oracle(pt, orc, ctx, tmp_path, grp_mode, wide_build):
    """The headline stand-in is evenly tessellated. tools/uneven_mesh.py is its opposite: panels with edges of 2-4 units, 300
    slivers of 4 x 0.006, 40 blobs with edges of ~0.02 (100:1 in edge length, 10^4:1 in area), 1500 needles of aspect 100:1;
    diffuse, glossy and glass materials. Render and ray-by-ray hits equal the oracle through the default traversal, its
    counting twin and the strict kernel, for both ways of cutting leaves into groups (grp_mode: bisection / strips) and for both
    candidates of the wide tree (wide_build 1 = collapsed from the reference's KD tree, 2 = surface-area heuristic; 0 = the
    library's choice, which on this mesh is the latter: 32 instead of 58 node visits per traversal)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("uneven_mesh", os.path.join(os.path.dirname(__file__), "..", "tools", "uneven_mesh.py"))
    um = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(um)
    path, ntri = um.write_scene(str(tmp_path))
    try:
        _tune(pt, ctx, b"grp_mode", grp_mode)
        _tune(pt, ctx, b"wide_build", wide_build)
        desc, cam = pt.load_scene(path)
        assert desc.contents.num_triangles == ntri and desc.contents.num_nodes > 1000
        scene = pt.host().pth_scene_device()
        pt.lib().ptx_scene_wide_sah_objects.argtypes = [C.c_void_p]
        assert pt.lib().ptx_scene_wide_sah_objects(scene) == (0 if wide_build == 1 else 1)
        gpuops.render_init(ctx, cam)
        acc_o, st_o = orc.render(desc, cam, 0, 8, 5)
        assert st_o.hit_mesh > 10000
        rng = np.random.default_rng(9)
        o1, d1 = orc.generate_rays(cam, 2, 5)
        o2 = rng.uniform([-2.2, -0.2, -1.2], [2.2, 1.4, 1.2], (40000, 3)).astype(np.float32)
        d2 = rng.normal(size=(40000, 3))
        d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
        # rays ALONG the slivers and panels (in-plane directions a hair above them)
        o3 = np.stack([rng.uniform(-1.9, 1.9, 20000), 0.4 + rng.uniform(-1e-4, 1e-4, 20000), rng.uniform(-1, 1, 20000)], 1).astype(np.float32)
        d3 = np.stack([rng.normal(size=20000), rng.normal(scale=1e-4, size=20000), rng.normal(size=20000)], 1)
        d3 = (d3 / np.linalg.norm(d3, axis=1, keepdims=True)).astype(np.float32)
        o, dd = np.concatenate([o1, o2, o3]), np.concatenate([d1, d2, d3])
        obj = 2
        ho = orc.trace(desc, o, dd, obj)
        assert (ho["hit"] == 1).sum() > 20000
        for variant in (0, 13, 14, 15):
            _tune(pt, ctx, b"trav_variant", variant)
            acc_g, st_g = gpuops.render(ctx, scene, cam, 0, 8, 5)
            bad = np.nonzero((acc_o.view(np.uint32) != acc_g.view(np.uint32)).any(axis=1))[0]
            assert len(bad) == 0, f"grp_mode {grp_mode} variant {variant}: {len(bad)} pixels differ, first {bad[:5]}"
            assert st_o.rays == st_g.rays
            hq = gpuops.trace_queued(ctx, scene, o, dd, obj)
            for k in ("object", "prim", "t_bits", "bary"):
                badr = np.nonzero((ho[k] != hq[k]).reshape(len(o), -1).any(axis=1))[0]
                assert len(badr) == 0, f"grp_mode {grp_mode} variant {variant}: {k} differs for {len(badr)} rays, first {badr[:5]}"
    finally:
        _tune(pt, ctx, b"trav_variant", -1)
        _tune(pt, ctx, b"grp_mode", -1)
        _tune(pt, ctx, b"wide_build", -1)


@pytest.mark.parametrize("seed", list(range(11, 27)))
def test_random_soups_render_like_the_oracle(pt, orc, ctx, tmp_path, seed):
    """Fuzz: triangle soups of random size, density and triangle shape (needles and slivers included), a random
    camera inside or outside the cloud, mirrors and glass among the materials so that secondary rays start on
    surfaces and leave in all directions: accumulation buffers and ray-by-ray hits of the wide traversal equal
    the oracle bit for bit."""
    rng = np.random.default_rng(seed)
    n_tris = int(rng.integers(300, 6000))
    spread = float(10.0 ** rng.uniform(-0.5, 1.0))
    c = rng.uniform(-spread, spread, (n_tris, 3))
    size = 10.0 ** rng.uniform(-2.5, 0.0, (n_tris, 1, 1)) * spread
    v = c[:, None, :] + size * rng.normal(size=(n_tris, 3, 3)) * rng.uniform(0.02, 1.0, (n_tris, 1, 3))
    mats = ["white.diff", "mirror.spec", "glass.trans"]
    lines = ["mtllib m.mtl"]
    for tri in v:
        for q in tri:
            lines.append("v %.9g %.9g %.9g" % tuple(q))
    for i in range(n_tris):
        if i % 97 == 0:
            lines.append("usemtl " + mats[(i // 97) % 3])
        lines.append("f %d %d %d" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
    (tmp_path / "soup.obj").write_text("\n".join(lines) + "\n")
    (tmp_path / "m.mtl").write_text("newmtl white.diff\nKd 0.8 0.7 0.6\nnewmtl mirror.spec\nKd 0.9 0.9 0.9\nNs 600\nd 0.7\n"
                                    "newmtl glass.trans\nKd 0.95 0.95 0.95\nNi 1.5\nd 0.1\n")
    pos = rng.uniform(-1.5, 1.5, 3) * spread
    look = -pos + rng.normal(0, 0.2 * spread, 3)
    look /= np.linalg.norm(look)
    (tmp_path / "s.json").write_text(
        '{"camera": {"pixelWidth": 80, "pixelHeight": 56, "direction": [%.6f,%.6f,%.6f], "up": [0,1,0], "position": [%.6f,%.6f,%.6f],'
        ' "screenDist": 1.2, "screenHeight": 2}, "objects": ['
        '{"type": "sphere", "render": true, "radius": %.6f, "position": [0, 0, 0], "material": {"type": "lumi", "Kd": [1], "Ke": [3]}},'
        '{"type": "mesh", "render": true, "src": "%s"}]}' % (*look, *pos, 30.0 * spread, tmp_path / "soup.obj"))
    desc, cam = pt.load_scene(str(tmp_path / "s.json"))
    assert desc.contents.num_nodes > 3
    scene = pt.host().pth_scene_device()
    gpuops.render_init(ctx, cam)
    acc_o, st_o = orc.render(desc, cam, 0, 6, seed)
    acc_g, st_g = gpuops.render(ctx, scene, cam, 0, 6, seed)
    bad = np.nonzero((acc_o.view(np.uint32) != acc_g.view(np.uint32)).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} pixels differ, first {bad[:5]}"
    assert st_o.rays == st_g.rays and st_o.paths == st_g.paths
    # ray by ray: primary rays and rays from random points inside the cloud in random directions
    o1, d1 = orc.generate_rays(cam, -1, 0)
    n2 = 20000
    o2 = rng.uniform(-spread, spread, (n2, 3)).astype(np.float32)
    d2 = rng.normal(size=(n2, 3))
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    o, dd = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    ho, hq = orc.trace(desc, o, dd, 1), gpuops.trace_queued(ctx, scene, o, dd, 1)
    assert (ho["hit"] == 1).sum() > 100
    for k in ("object", "prim", "t_bits", "bary"):
        badr = np.nonzero((ho[k] != hq[k]).reshape(len(o), -1).any(axis=1))[0]
        assert len(badr) == 0, f"{k} differs for {len(badr)} rays, first {badr[:5]}"
