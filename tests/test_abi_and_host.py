"""CPU tests: the C ABI library loads and exports every declared symbol, the host layer mirrors the
reference's loader semantics, and nothing computes without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pt):
    L = pt.lib()
    header = open(os.path.join(ROOT, "include", "ptx_hip.h")).read()
    declared = set(re.findall(r"\b(ptx_[a-z_0-9]+)\s*\(", header))
    assert declared == set(pt.PTX_SYMBOLS), declared ^ set(pt.PTX_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.ptx_version()


def test_struct_sizes_match_header(pt):
    # sizes the C compiler gives the structs in include/ptx_hip.h
    assert C.sizeof(pt.PtxCamera) == 96
    assert C.sizeof(pt.PtxMaterial) == 48
    assert C.sizeof(pt.PtxTriangle) == 100
    assert C.sizeof(pt.PtxKDNode) == 40
    assert C.sizeof(pt.PtxObject) == 56
    assert C.sizeof(pt.PtxSceneDesc) == 96 and C.sizeof(pt.PtxTexture) == 16
    assert C.sizeof(pt.PtxHit) == 72
    assert C.sizeof(pt.PtxStats) == 320   # 21 + 8 execution counters + 7 shading-divergence counters + 4 segment-kernel phase clocks
    assert C.sizeof(pt.PtxBsdfCase) == 104


def test_no_cpu_fallback_without_device(pt):
    L = pt.lib()
    if L.ptx_device_count() > 0:
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = L.ptx_ctx_create(0, C.byref(ctx))
    assert rc < 0 and not ctx.value
    assert b"no HIP device" in L.ptx_last_error() or b"hipGetDeviceCount" in L.ptx_last_error()
    # the host layer refuses to load a scene for rendering without a device
    H = pt.host()
    H.pth_set_device_upload(1)
    assert H.pth_scene_load(os.path.join(ROOT, "scenes/scene_example.json").encode(), 0, 0, -1, -1, -1, -1) != 0
    assert H.pth_render_samples(0, 1, 0) < 0


def test_scene_counts_and_camera(pt):
    expect = {  # objects, triangles, nodes of the largest mesh
        "example": (4, 24, 1), "geoms": (7, 30, 1), "teapot": (7, 1572, 127), "sky_map": (7, 14, 1)}
    for name, (nobj, ntri, nnodes) in expect.items():
        desc, cam = pt.load_scene(name, upload=False)
        d = desc.contents
        assert d.num_objects == nobj and d.num_triangles == ntri, name
        assert max(d.objects[i].num_nodes for i in range(nobj)) == nnodes
        assert cam.preview_mode == 0 and cam.aabb_mode == 0 and cam.max_ray_bounces == 5
        assert cam.aa_jitter == 2.0 and cam.gamma_corr == 0.5
    desc, cam = pt.load_scene("teapot", upload=False)
    # teapot tree statistics measured on the reference build in SURVEY.md section 6: 127 nodes, 64 leaves, depth 6
    ob = [desc.contents.objects[i] for i in range(desc.contents.num_objects) if desc.contents.objects[i].num_nodes == 127][0]
    nodes = [desc.contents.nodes[ob.first_node + i] for i in range(127)]
    leaves = [n for n in nodes if n.left == -1 and n.right == -1]
    assert len(leaves) == 64 and ob.max_depth == 6
    sizes = [n.num_tris for n in leaves]
    assert sum(sizes) == 1560 and max(sizes) == 28
    assert abs(np.mean(sizes) - 24.4) < 0.05


def test_resolution_override_matches_init_formula(pt):
    _, cam = pt.load_scene("example", width=512, height=512, upload=False)
    assert (cam.width, cam.height) == (512, 512)
    assert cam.pixel_size == np.float32(2.0) / np.float32(512)
    assert tuple(cam.screen_half) == (1.0, 1.0)
    _, cam = pt.load_scene("lamborghini", width=1920, height=1080, upload=False, car_tris=4000)
    assert cam.screen_half[0] == np.float32(np.float32(1.0) * np.float32(1920) / np.float32(1080))


def test_render_false_objects_are_skipped_and_cap(pt):
    desc, _ = pt.load_scene("lamborghini", upload=False, car_tris=4000)
    d = desc.contents
    assert d.num_objects == 5  # 2 of 4 spheres have render:false
    assert [d.objects[i].type for i in range(5)] == [0, 0, 1, 1, 1]
    # unknown transform keys (xrotate / xtranslate) are ignored: sky cube = translate(0,70,0) * scale(100)
    sky = d.objects[2]
    ys = [d.triangles[sky.first_triangle + t].a[1] for t in range(sky.num_triangles)]
    assert min(ys) == pytest.approx(70 - 535.3058, rel=1e-6)


def test_material_semantics(pt):
    desc, _ = pt.load_scene("teapot", upload=False)
    d = desc.contents
    sph = d.materials[d.objects[3].first_material]  # "spec", d 0.8, Ns 990
    assert sph.type == pt.MAT_SPECULAR and sph.sharpness == np.float32(990.0) / np.float32(1000.0)
    assert sph.reflect_factor == np.float32(0.8) and sph.texture == -1
    tea = d.objects[6]
    mats = [d.materials[tea.first_material + i] for i in range(tea.num_materials)]
    assert mats[0].type == pt.MAT_DIFFUSE and mats[0].color[0] == np.float32(0.6)  # DefaultMaterial
    assert mats[1].type == pt.MAT_SPECULAR and mats[1].sharpness == np.float32(995.0) / np.float32(1000.0)
    # mtllib fallback: scenes/example/cube.obj names materials.mtl, cube.mtl sits beside it
    desc, _ = pt.load_scene("example", upload=False)
    d = desc.contents
    cube = d.objects[3]
    m = d.materials[cube.first_material + d.triangles[cube.first_triangle].material]
    assert tuple(np.float32(x) for x in m.color) == (np.float32(0.95), np.float32(0.95), np.float32(0.4))


def test_quads_are_fan_triangulated_in_file_order(pt):
    desc, _ = pt.load_scene("example", upload=False)
    d = desc.contents
    box = d.objects[2]  # cornell box: 6 quads -> 12 triangles, transform translate(0,0,2.5)*scale(1,1,2)
    t0, t1 = d.triangles[box.first_triangle], d.triangles[box.first_triangle + 1]
    # back wall quad (-5,0,-5) (5,0,-5) (5,10,-5) (-5,10,-5): z -> -5*2 + 2.5 = -7.5
    assert tuple(t0.a) == (-5.0, 0.0, -7.5) and tuple(t1.a) == (-5.0, 0.0, -7.5)
    assert tuple(t0.e1) == (10.0, 0.0, 0.0) and tuple(t0.e2) == (10.0, 10.0, 0.0)
    assert tuple(t1.e1) == (10.0, 10.0, 0.0) and tuple(t1.e2) == (0.0, 10.0, 0.0)
    assert tuple(t0.na) == (0.0, 0.0, 1.0)


def test_unknown_object_type_and_bad_json_fail(pt, tmp_path):
    H = pt.host()
    H.pth_set_device_upload(0)
    p = tmp_path / "bad.json"
    p.write_text('{"camera": {"pixelWidth": 4, "pixelHeight": 4}, "objects": [{"type": "cone", "render": true}]}')
    assert H.pth_scene_load(str(p).encode(), 0, 0, -1, -1, -1, -1) != 0
    p.write_text('{"camera": ')
    assert H.pth_scene_load(str(p).encode(), 0, 0, -1, -1, -1, -1) != 0
    assert b"Line" in H.pth_last_error()
    assert H.pth_scene_load(b"/nonexistent/scene.json", 0, 0, -1, -1, -1, -1) != 0
    assert b"Unable to open file" in H.pth_last_error()
    p.write_text('{"camera": {"pixelWidth": 4, "pixelHeight": 4, "direction": [0,0,-1], "up": [0,1,0], "position": [0,0,0], '
                 '"screenDist": 1, "screenHeight": 1}, "objects": [{"type": "mesh", "render": true, "src": "./scenes/none.obj"}]}')
    assert H.pth_scene_load(str(p).encode(), 0, 0, -1, -1, -1, -1) != 0
    # empty scene (no objects) loads
    p.write_text('{"camera": {"pixelWidth": 4, "pixelHeight": 4, "direction": [0,0,-1], "up": [0,1,0], "position": [0,0,0], '
                 '"screenDist": 1, "screenHeight": 1}}')
    assert H.pth_scene_load(str(p).encode(), 0, 0, -1, -1, -1, -1) == 0
    assert H.pth_scene_desc().contents.num_objects == 0


def test_png_writer_roundtrip(pt, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    p = str(tmp_path / "o.png").encode()
    assert pt.host().pth_write_png(p, 17, 13, 3, img.ctypes.data_as(C.POINTER(C.c_ubyte)), 1) == 0
    back = np.asarray(Image.open(p.decode()))
    assert np.array_equal(back, img[:, ::-1])  # mirror_h = what the reference's viewer shows


def test_obj_reader_features(pt, tmp_path):
    """OBJ features beyond the shipped scenes - negative (relative) indices, `g` groups,
    several `usemtl` runs in one object, v//vn and v/vt/vn corners, a quad, and the `mtllib` fallback to
    <obj stem>.mtl when the named library does not exist. Primitive ids follow file order."""
    (tmp_path / "m.mtl").write_text("newmtl red.diff\nKd 1 0 0\nnewmtl glow.lumi\nKd 1 1 1\nKe 3 3 3\nnewmtl mirror.spec\nKd 0.5 0.5 0.5\nNs 500\nd 0.7\n")
    (tmp_path / "m.obj").write_text(
        "mtllib does_not_exist.mtl\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\n"
        "g first\nusemtl red.diff\nf 1//1 2//1 3//1\n"
        "usemtl glow.lumi\nf -4/1/1 -3/2/1 -2/3/1 -1/1/1\n"          # quad with relative indices -> 2 triangles
        "v 0 0 1\nv 1 0 1\nv 0 1 1\n"
        "g second\nusemtl mirror.spec\nf -3 -2 -1\n")
    (tmp_path / "s.json").write_text(
        '{"camera": {"pixelWidth": 8, "pixelHeight": 8, "direction": [0,0,-1], "up": [0,1,0], "position": [0.5,0.5,3],'
        ' "screenDist": 1, "screenHeight": 1}, "objects": [{"type": "mesh", "render": true, "src": "%s"}]}' % (tmp_path / "m.obj"))
    H = pt.host()
    H.pth_set_device_upload(0)
    assert H.pth_scene_load(str(tmp_path / "s.json").encode(), 0, 0, -1, -1, -1, -1) == 0, H.pth_last_error()
    d = H.pth_scene_desc().contents
    assert d.num_objects == 1 and d.num_triangles == 4
    mats = [d.materials[i] for i in range(d.num_materials)]
    tri = [d.triangles[i] for i in range(4)]
    # triangle 0: red diffuse; 1-2: the quad, luminescent, fan (0,1,2), (0,2,3); 3: mirror from the later vertices
    assert mats[tri[0].material].type == 1 and tuple(mats[tri[0].material].color) == (1.0, 0.0, 0.0)
    assert mats[tri[1].material].type == 0 and mats[tri[2].material].type == 0 and tuple(mats[tri[1].material].emittance) == (3.0, 3.0, 3.0)
    assert mats[tri[3].material].type == 2 and mats[tri[3].material].sharpness == 0.5 and abs(mats[tri[3].material].reflect_factor - 0.7) < 1e-7
    abc = np.ctypeslib.as_array(H.pth_scene_tri_abc(), shape=(4, 9))
    assert abc[1].tolist() == [0, 0, 0, 1, 0, 0, 1, 1, 0] and abc[2].tolist() == [0, 0, 0, 1, 1, 0, 0, 1, 0]
    assert abc[3].tolist() == [0, 0, 1, 1, 0, 1, 0, 1, 1]
    assert tuple(tri[1].tb) == (1.0, 0.0) and tuple(tri[2].tc) == (0.0, 0.0)
    assert tuple(tri[0].na) == (0.0, 0.0, 1.0)


def test_look_ahead_setting_of_the_one_sample_loop(pt):
    """host/kernel.h: runCUDA may trace kernelLookAhead() - 1 samples ahead of the one it adds; 16 unless PTX_LOOK_AHEAD or
    kernelSetLookAhead say otherwise, never below 1 (= one pass per call, the reference's schedule). No device needed."""
    H = pt.host()
    before = H.pth_look_ahead()
    try:
        assert before == int(os.environ.get("PTX_LOOK_AHEAD", "16")) or before >= 1
        H.pth_set_look_ahead(5)
        assert H.pth_look_ahead() == 5
        H.pth_set_look_ahead(0)
        assert H.pth_look_ahead() == 1
        H.pth_set_look_ahead(-7)
        assert H.pth_look_ahead() == 1
    finally:
        H.pth_set_look_ahead(before)
