#!/usr/bin/env python3
"""Generates the fixtures in this directory.  Run from the repo root IN THE BUILD CONTAINER:

    make -C oracle all ref && make -C path-tracer_amd && python tests/golden/make_golden.py

Two files, two kinds of provenance:

ref_pins.json        outputs of the REFERENCE's own code, compiled unchanged from /root/reference by
                     oracle/Makefile (target ref: utility.cpp, camera.cpp, jsonResolve.cpp, jsoncpp.cpp):
                     fmedian, Camera::init / rotate / move, resolveFloat3 / resolveFloat / resolveMatType;
                     external/include/stb/stb_image.h (stbi_loadf of every shipped texture and of the synthetic
                     files of make_images.py) and external/include/glm (mesh transform chains).
                     These pin the oracle and the product's host code without needing the reference at
                     test time (it does not travel to the GPU box).
oracle_vectors.npz   outputs of the CPU oracle (oracle/pt_oracle.cpp) for the device path: KD-tree
                     digests, traversal hits, BSDF known answers, camera rays, small renders. The
                     reference's device code cannot be built here (nvcc / cuRAND / Qt absent) and it ships
                     no tests or golden vectors, so these are "parity unpinned" regression vectors: they
                     freeze the oracle, and the GPU tests compare the HIP path with them bit for bit.

Inputs are data (seeded numpy arrays, the repo's own scene JSONs); nothing from the reference is stored.
"""
import ctypes as C
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("path-tracer_amd")
import orc  # noqa: E402

SCENES = ["example", "geoms", "teapot", "sky_map", "lamborghini"]


def bits(x):
    return int(np.float32(x).view(np.uint32))


def camera_json(name):
    text = open(os.path.join(ROOT, "scenes", f"scene_{name}.json")).read()
    return json.dumps(json.loads(text)["camera"])


def fmedian_inputs():
    rng = np.random.default_rng(5)
    cases = [rng.normal(size=n).astype(np.float32) for n in (1, 2, 3, 16, 17, 33, 1000, 4097)]
    cases.append(np.zeros(64, np.float32))
    cases.append(np.repeat(rng.normal(size=7).astype(np.float32), 9))
    cases.append(np.arange(100, 0, -1).astype(np.float32))
    return cases


RESOLVE_CASES = ["[1, 2.5, -3e2]", "[7]", "[1, 2]", "3", "0.1", "950.0", '"spec"', '"trans"', '"lumi"', '"diff"', '"bogus"', "[]", "null"]
ROTATE_CASES = [("geoms", 35.0, -10.0, (0.2, 0.0, -0.4)), ("sky_map", -95.0, 20.0, (0.0, 1.0, 0.0)), ("example", 0.0, 0.0, (0.0, 0.0, 0.0)),
                ("teapot", 180.0, 45.0, (1.0, -2.0, 3.0))]


def make_ref_pins():
    R = orc.ref()
    if R is None:
        raise SystemExit("oracle/_ref/libref_host.so missing: run `make -C oracle ref` (needs /root/reference)")
    out = {"provenance": "reference sources compiled unchanged (oracle/Makefile target ref)", "fmedian": [], "camera_init": {},
           "camera_rotate": [], "resolve": []}
    for a in fmedian_inputs():
        out["fmedian"].append({"n": len(a), "sha1_of_input": hashlib.sha1(a.tobytes()).hexdigest(),
                               "median_bits": bits(R.ref_fmedian(orc.fptr(a.copy()), len(a)))})
    for name in SCENES:
        f = (C.c_float * 20)(); i = (C.c_int * 6)()
        assert R.ref_camera_init(camera_json(name).encode(), f, i) == 0
        out["camera_init"][name] = {"float_bits": [bits(x) for x in f], "ints": list(i)}
    for name, h, v, mv in ROTATE_CASES:
        f = (C.c_float * 12)()
        assert R.ref_camera_rotate(camera_json(name).encode(), h, v, (C.c_float * 3)(*mv), f) == 0
        out["camera_rotate"].append({"scene": name, "h": h, "v": v, "move": list(mv), "float_bits": [bits(x) for x in f]})
    for text in RESOLVE_CASES:
        f3 = (C.c_float * 3)(); f = C.c_float(); m = C.c_int()
        assert R.ref_resolve(('{"k": %s}' % text).encode(), b"k", f3, C.byref(f), C.byref(m)) == 0
        out["resolve"].append({"json": text, "float3_bits": [bits(x) for x in f3], "float_bits": bits(f.value), "mat_type": m.value})
    # the float3 helpers of external/include/cutil_math.h and wang_hash of src/cutility.cuh, run by the reference's
    # own headers: inputs and outputs (bit patterns) go to ref_vec3.npz
    a, b, s = orc.vec3_cases()
    bits3 = orc.vec3_ops(R.ref_vec3_ops, a, b, s)
    seeds = np.concatenate([np.arange(0, 64), np.array([12345, 0x7fffffff, 0x80000000, 0xffffffff]),
                            np.random.default_rng(7).integers(0, 2 ** 32, 200)]).astype(np.uint32)
    hashes = np.array([R.ref_wang_hash(int(x)) for x in seeds], np.uint32)
    np.savez_compressed(os.path.join(HERE, "ref_vec3.npz"), a=a, b=b, s=s, out_bits=bits3, wang_in=seeds, wang_out=hashes)
    out["vec3"] = {"file": "ref_vec3.npz", "cases": int(len(a)), "sha1_of_outputs": hashlib.sha1(bits3.tobytes()).hexdigest()}
    # texture decode: what the reference's own stb_image.h (stbi_loadf, req_comp 0) makes of every shipped
    # texture and of the synthetic fixtures of make_images.py; sha1 over the float32 array
    out["images"] = {}
    for rel in orc.image_files():
        r = orc.decode_with(R.ref_stbi_loadf, R.ref_stbi_free, rel)
        out["images"][rel] = {"fails": True} if r is None else {"w": r[0], "h": r[1], "channels": r[2], "sha1": hashlib.sha1(r[3].tobytes()).hexdigest()}
    # mesh transforms through the reference's glm (translate / rotate / scale chains, transpose(inverse(M)), M * vec4)
    pts = np.random.default_rng(22).normal(size=(64, 3)).astype(np.float32)
    out["transforms"] = []
    for ops in orc.transform_cases():
        m, nm, o1, o2 = orc.transform_eval(R.ref_glm_transform, R.ref_glm_apply, ops, pts)
        out["transforms"].append({"m16_bits": m.tolist(), "normal16_bits": nm.tolist(),
                                  "sha1_points": hashlib.sha1(o1.tobytes() + o2.tobytes()).hexdigest()})
    with open(os.path.join(HERE, "ref_pins.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("ref_pins.json:", {k: len(v) for k, v in out.items() if not isinstance(v, str)})


def kd_digest(abc):
    nodes = C.POINTER(pt.PtxKDNode)(); lt = C.POINTER(C.c_int32)(); nlt, depth = C.c_int(), C.c_int()
    a = np.ascontiguousarray(abc, np.float32)
    n = orc.lib().orc_kd_build(orc.fptr(a), len(a) // 9, C.byref(nodes), C.byref(lt), C.byref(nlt), C.byref(depth))
    nb = C.string_at(nodes, n * C.sizeof(pt.PtxKDNode)); lb = C.string_at(lt, nlt.value * 4)
    orc.lib().orc_free(nodes); orc.lib().orc_free(lt)
    return n, nlt.value, depth.value, hashlib.sha1(nb).hexdigest(), hashlib.sha1(lb).hexdigest()


def golden_meshes():
    """name -> flat float32 (a, b, c) x n array; all built from seeds or the repo's scenes."""
    rng = np.random.default_rng(11)
    out = {}
    for scene, obj in (("teapot", 6), ("example", None), ("geoms", None)):
        desc, _ = pt.load_scene(scene, upload=False)
        d = desc.contents
        abc = np.ctypeslib.as_array(pt.host().pth_scene_tri_abc(), shape=(d.num_triangles * 9,)).copy()
        for oi in range(d.num_objects):
            ob = d.objects[oi]
            if ob.type == 1 and ob.num_triangles > 0 and (obj is None or oi == obj):
                out[f"{scene}:{oi}"] = abc[ob.first_triangle * 9:(ob.first_triangle + ob.num_triangles) * 9]
    g = 64
    xs, zs = np.meshgrid(np.arange(g, dtype=np.float32), np.arange(g, dtype=np.float32))
    tris = []
    for x, z in zip(xs.ravel(), zs.ravel()):
        tris += [[x, 0, z, x + 1, 0, z, x + 1, 0, z + 1], [x, 0, z, x + 1, 0, z + 1, x, 0, z + 1]]
    out["sheet"] = np.array(tris, np.float32).ravel()
    out["random5000"] = rng.normal(0, 3, 5000 * 9).astype(np.float32)
    out["dup100"] = np.tile(rng.normal(0, 1, 9).astype(np.float32), 100)
    out["seventeen"] = rng.normal(0, 1, 17 * 9).astype(np.float32)
    return out


def trace_rays_for(desc, cam, n_secondary=768, seed=2):
    """Primaries of one sample on the (small) camera plus seeded secondary rays leaving hit points, plus
    axis-parallel directions (inv_direction = +-inf)."""
    o, d = orc.generate_rays(cam, 1, 7)
    h = orc.trace(desc, o, d)
    rng = np.random.default_rng(seed)
    idx = np.nonzero(h["hit"] == 1)[0]
    pick = rng.choice(idx, size=min(n_secondary, len(idx)), replace=len(idx) < n_secondary) if len(idx) else np.zeros(0, int)
    p = h["point"].view(np.float32)[pick]
    nrm = h["normal"].view(np.float32)[pick]
    nd = rng.normal(size=(len(pick), 3)).astype(np.float32)
    nd /= np.linalg.norm(nd, axis=1, keepdims=True).astype(np.float32)
    so = (p + nrm * np.float32(1e-3)).astype(np.float32)
    axis = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 1, 0]], np.float32)
    ao = np.tile(cam_pos(cam), (4, 1)).astype(np.float32)
    keep = np.arange(0, len(o), max(1, len(o) // 1024))[:1024]
    return np.concatenate([o[keep], so, ao]), np.concatenate([d[keep], nd, axis])


def cam_pos(cam):
    return np.array(list(cam.position), np.float32)


def make_oracle_vectors():
    vec = {}
    meta = {"provenance": "CPU oracle (oracle/pt_oracle.cpp); parity unpinned against the reference's device code", "kd": {}, "trace": [],
            "render": []}
    for name, abc in golden_meshes().items():
        n, nlt, depth, hn, hl = kd_digest(abc)
        meta["kd"][name] = {"triangles": len(abc) // 9, "nodes": n, "leaf_tris": nlt, "max_depth": depth, "sha1_nodes": hn, "sha1_leaf_tris": hl,
                            "sha1_input": hashlib.sha1(np.ascontiguousarray(abc, np.float32).tobytes()).hexdigest()}
    for scene in ("example", "geoms", "teapot", "sky_map"):
        desc, cam = pt.load_scene(scene, width=64, height=48, textures=1 if scene == "sky_map" else 0, upload=False)
        o, d = trace_rays_for(desc, cam)
        h = orc.trace(desc, o, d)
        vec[f"trace_{scene}_o"] = o; vec[f"trace_{scene}_d"] = d
        vec[f"trace_{scene}_hits"] = np.stack([h[k] for k in ("hit", "object", "prim", "t_bits", "leaf_hash", "n_inner", "n_box", "n_tri", "n_leaf")], 1)
        vec[f"trace_{scene}_pnb"] = np.concatenate([h["point"], h["normal"], h["bary"]], 1)
        meta["trace"].append({"scene": scene, "rays": len(o), "hits": int(h["hit"].sum())})
        # camera rays (sample 0 and 5, seed 42) at 5 probe pixels + the full first row
        for s in (0, 5):
            ro, rd = orc.generate_rays(cam, s, 42)
            vec[f"rays_{scene}_{s}"] = np.concatenate([ro, rd], 1).view(np.uint32)[:: 37]
        acc, st = orc.render(desc, cam, 0, 8, 42)
        vec[f"render_{scene}"] = acc.view(np.uint32)
        meta["render"].append({"scene": scene, "width": 64, "height": 48, "spp": 8, "seed": 42, "textures": 1 if scene == "sky_map" else 0,
                               "stats": {k: v for k, v in st.as_dict().items() if not k.startswith(("lane_", "diag", "x_", "sh_", "seg_"))}})
    # BSDF known answers
    rng = np.random.default_rng(3)
    n = 768
    cases = (pt.PtxBsdfCase * n)()
    for i in range(n):
        c = cases[i]
        dv = rng.normal(size=3); dv /= np.linalg.norm(dv)
        nv = rng.normal(size=3); nv /= np.linalg.norm(nv)
        if i % 16 == 0:
            nv = np.array([0.0, 0.0, 1.0 if i % 32 else -1.0])   # ONB fallback branch (axis.x == axis.y == 0)
        c.dir[:] = dv.astype(np.float32); c.normal[:] = nv.astype(np.float32)
        c.point[:] = rng.normal(size=3).astype(np.float32)
        c.mat_type = int(i % 4)
        c.reflect_factor = float(rng.random()); c.refract_index = float(1.0 + rng.random()); c.sharpness = float(rng.random())
        c.u[:] = ((rng.integers(0, 1 << 24, 4) + 1) / float(1 << 24)).astype(np.float32)
    vec["bsdf_in"] = np.frombuffer(bytes(cases), np.uint32).reshape(n, -1).copy()
    assert orc.lib().orc_bsdf_eval(n, cases) == 0
    vec["bsdf_out"] = np.frombuffer(bytes(cases), np.uint32).reshape(n, -1).copy()
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **vec)
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("oracle_vectors.npz:", os.path.getsize(os.path.join(HERE, "oracle_vectors.npz")), "bytes;", len(vec), "arrays")


if __name__ == "__main__":
    pt.host().pth_set_device_upload(0)
    make_ref_pins()
    make_oracle_vectors()
