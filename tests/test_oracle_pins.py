"""The oracle (and the product's host code) against the only pieces of the reference that compile
unchanged in this image: oracle/_ref/libref_host.so = the reference's utility.cpp, camera.cpp,
jsonResolve.cpp, jsoncpp.cpp (oracle/Makefile, target ref). Skipped if _ref was not built."""
import ctypes as C
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = ["example", "geoms", "teapot", "sky_map", "lamborghini"]


@pytest.fixture(scope="module")
def ref(orc):
    r = orc.ref()
    if r is None:
        pytest.skip("oracle/_ref not built (needs /root/reference; `make -C oracle ref`)")
    return r


def _camera_json(name):
    text = open(os.path.join(ROOT, "scenes", f"scene_{name}.json")).read()
    return json.dumps(json.loads(text)["camera"]).encode()


def test_fmedian_pinned(ref, orc, pt):
    rng = np.random.default_rng(5)
    cases = [rng.normal(size=n).astype(np.float32) for n in (1, 2, 3, 16, 17, 33, 1000, 4097)]
    cases.append(np.zeros(64, np.float32))
    cases.append(np.repeat(rng.normal(size=7).astype(np.float32), 9))  # many ties
    cases.append(np.arange(100, 0, -1).astype(np.float32))
    for a in cases:
        r = ref.ref_fmedian(orc.fptr(a.copy()), len(a))
        o = orc.lib().orc_fmedian(orc.fptr(a.copy()), len(a))
        h = pt.host().pth_fmedian(orc.fptr(a.copy()), len(a))
        assert np.float32(r).view(np.uint32) == np.float32(o).view(np.uint32) == np.float32(h).view(np.uint32)
    assert ref.ref_fmedian(orc.fptr(np.zeros(1, np.float32)), 0) == 0.0 == orc.lib().orc_fmedian(orc.fptr(np.zeros(1, np.float32)), 0)


@pytest.mark.parametrize("name", SCENES)
def test_camera_init_pinned(ref, pt, name):
    out = (C.c_float * 20)()
    iout = (C.c_int * 6)()
    assert ref.ref_camera_init(_camera_json(name), out, iout) == 0
    _, cam = pt.load_scene(name, upload=False, preview=1, car_tris=4000)
    got = list(cam.position) + list(cam.direction) + list(cam.up) + list(cam.right)
    assert np.array_equal(np.array(got, np.float32).view(np.uint32), np.array(out[:12], np.float32).view(np.uint32))
    assert cam.pixel_size == out[12] == out[13]
    assert (cam.screen_half[0], cam.screen_half[1], cam.screen_dist, cam.aa_jitter, cam.gamma_corr) == tuple(out[14:19])
    assert cam.max_ray_bounces == int(out[19])
    assert (cam.width, cam.height) == (iout[0], iout[1]) and iout[2] == cam.width * cam.height
    # reference defaults: preview on, aabb off, textures off (Camera::init)
    assert (iout[3], iout[4], iout[5]) == (1, 0, 0)
    assert cam.preview_mode == 1 and cam.aabb_mode == 0 and cam.texture_enabled == 0


def test_camera_rotate_move_pinned(ref, pt):
    H = pt.host()
    for name, h, v, mv in (("geoms", 35.0, -10.0, (0.2, 0.0, -0.4)), ("sky_map", -95.0, 20.0, (0, 1, 0)), ("example", 0.0, 0.0, (0, 0, 0))):
        out = (C.c_float * 12)()
        move = (C.c_float * 3)(*mv)
        assert ref.ref_camera_rotate(_camera_json(name), h, v, move, out) == 0
        _, cam0 = pt.load_scene(name, upload=False)
        H.pth_camera_control(b"h_ang", h)
        H.pth_camera_control(b"v_ang", v)
        H.pth_camera_control(b"pos_x", float(np.float32(cam0.position[0]) + np.float32(mv[0])))
        H.pth_camera_control(b"pos_y", float(np.float32(cam0.position[1]) + np.float32(mv[1])))
        H.pth_camera_control(b"pos_z", float(np.float32(cam0.position[2]) + np.float32(mv[2])))
        cam = pt.PtxCamera()
        assert H.pth_camera_get(C.byref(cam)) == 0
        got = list(cam.position) + list(cam.direction) + list(cam.up) + list(cam.right)
        assert np.array_equal(np.array(got, np.float32).view(np.uint32), np.array(out[:], np.float32).view(np.uint32)), name


def test_json_resolve_pinned(ref, pt, tmp_path):
    """resolveFloat3 / resolveFloat / resolveMatType through the reference vs through a sphere scene
    parsed by the product's own JSON reader."""
    H = pt.host()
    H.pth_set_device_upload(0)
    cases = [("[1, 2.5, -3e2]", None), ("[7]", None), ("[1, 2]", None), ("3", None), ("0.1", None), ("950.0", None),
             ('"spec"', 2), ('"trans"', 3), ('"lumi"', 0), ('"diff"', 1), ('"bogus"', 1), ("[]", None), ("null", None)]
    for text, mat in cases:
        f3 = (C.c_float * 3)(); f = C.c_float(); m = C.c_int()
        assert ref.ref_resolve(('{"k": %s}' % text).encode(), b"k", f3, C.byref(f), C.byref(m)) == 0
        if mat is not None:
            assert m.value == mat
        # product: put the value in a sphere's Kd / d / type and read the material back
        kd = text if text.startswith("[") or text == "null" else "[0]"
        dval = text if text[0].isdigit() else "0"
        ty = text if text.startswith('"') else '"diff"'
        p = tmp_path / "s.json"
        p.write_text('{"camera": {"pixelWidth": 4, "pixelHeight": 4, "direction": [0,0,-1], "up": [0,1,0], "position": [0,0,0],'
                     ' "screenDist": 1, "screenHeight": 1}, "objects": [{"type": "sphere", "render": true, "radius": 1,'
                     ' "position": [0,0,0], "material": {"type": %s, "Kd": %s, "d": %s}}]}' % (ty, kd, dval))
        assert H.pth_scene_load(str(p).encode(), 0, 0, -1, -1, -1, -1) == 0
        mat_got = H.pth_scene_desc().contents.materials[0]
        if text.startswith("[") or text == "null":
            assert tuple(mat_got.color) == tuple(f3), text
        if text[0].isdigit():
            assert mat_got.reflect_factor == f.value, text
        if text.startswith('"'):
            assert mat_got.type == m.value, text


def test_float3_helpers_and_wang_hash_pinned(ref, orc):
    """The float3 helpers every hot-path function is built from (dot, cross, length, normalize, reflect, scale,
    divide-by-scalar, + - *): the oracle's V3 functions against the REFERENCE's own external/include/cutil_math.h,
    and wang_hash against src/cutility.cuh, both compiled unchanged into oracle/_ref. Bit for bit (a NaN for a NaN)."""
    a, b, s = orc.vec3_cases()
    r = orc.vec3_ops(ref.ref_vec3_ops, a, b, s)
    o = orc.vec3_ops(orc.lib().orc_vec3_ops, a, b, s)
    same = (r == o) | (np.isnan(r.view(np.float32)) & np.isnan(o.view(np.float32)))
    assert same.all(), np.nonzero(~same.all(axis=1))[0][:10]
    assert np.isfinite(r.view(np.float32)).mean() > 0.9
    for seed in list(range(64)) + [12345, 0x7fffffff, 0x80000000, 0xffffffff]:
        assert ref.ref_wang_hash(seed) == orc.lib().orc_wang_hash(seed)


def test_texture_decode_equals_reference_stb(ref, orc, pt):
    """a16: pth_load_image_float (host/image_io.cpp) against stbi_loadf of the REFERENCE's own
    external/include/stb/stb_image.h (compiled unchanged into oracle/_ref, oracle/ref_stb.cpp), tolerance 0: every
    shipped texture (PNG, baseline JPEG with restart intervals, 29 TGAs) and the 148 synthetic files of
    tests/golden/make_images.py (subsampled / progressive / grey / CMYK JPEGs, every PNG colour type x depth x
    Adam7 + tRNS + palettes, raw / RLE / colour-mapped / 15-16-bit TGAs, truncated and bit-flipped files, which must
    fail or half-decode alike)."""
    load, free = orc.product_image_loader(pt.host())
    files = orc.image_files()
    assert len(files) > 170
    for rel in files:
        r = orc.decode_with(ref.ref_stbi_loadf, ref.ref_stbi_free, rel)
        g = orc.decode_with(load, free, rel)
        assert (r is None) == (g is None), rel
        if r is not None:
            assert r[:3] == g[:3], rel
            assert np.array_equal(r[3].view(np.uint32), g[3].view(np.uint32)), rel


def test_mesh_transforms_equal_reference_glm(ref, orc, pt):
    """a21: the loader's transform chains (world_load.cpp composeTransform on host/glmlite.h), transpose(inverse(M))
    and M * vec4 against the reference's vendored glm 0.9.8.5 (oracle/ref_glm.cpp), bit for bit, on the transform
    lists of every shipped scene and 160 random chains."""
    H = pt.host()
    fp = C.POINTER(C.c_float)
    H.pth_transform_compose.argtypes = [fp, C.c_int, fp, fp]
    H.pth_transform_apply.argtypes = [fp, fp, C.c_int, fp]
    pts = np.random.default_rng(22).normal(size=(64, 3)).astype(np.float32)
    cases = orc.transform_cases()
    assert len(cases) > 170
    for i, ops in enumerate(cases):
        r = orc.transform_eval(ref.ref_glm_transform, ref.ref_glm_apply, ops, pts)
        g = orc.transform_eval(H.pth_transform_compose, H.pth_transform_apply, ops, pts)
        for a, b in zip(r, g):
            fa, fb = a.view(np.float32), b.view(np.float32)
            assert ((a == b) | (np.isnan(fa) & np.isnan(fb))).all(), (i, ops)
