"""ctypes harness over the MI355X path tracer's two shared libraries.

    libptx_hip.so   the C ABI of include/ptx_hip.h (HIP kernels for gfx950)
    libpt_host.so   the C++ host layer with the reference's entry points
                    (parseJsonScene / sceneState.load / buildKDTree / kernelInit / runCUDA ...),
                    reached through host/host_capi.h

This package is plumbing for tests, smoke() and bench.py. It contains no compute and no
fallback: if the libraries are not built, importing `lib()`/`host()` raises.
(The directory name has a hyphen; import it with importlib.import_module("path-tracer_amd").)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)

PTX_OK = 0
OBJ_SPHERE, OBJ_MESH = 0, 1
MAT_LUMINESCENT, MAT_DIFFUSE, MAT_SPECULAR, MAT_TRANSPARENT = 0, 1, 2, 3


class PtxCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3),
                ("right", C.c_float * 3), ("width", C.c_int32), ("height", C.c_int32),
                ("pixel_size", C.c_float), ("screen_half", C.c_float * 2), ("screen_dist", C.c_float),
                ("aa_jitter", C.c_float), ("gamma_corr", C.c_float), ("max_ray_bounces", C.c_int32),
                ("preview_mode", C.c_int32), ("aabb_mode", C.c_int32), ("texture_enabled", C.c_int32)]


class PtxMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 3), ("texture", C.c_int32),
                ("emittance", C.c_float * 3), ("reflect_factor", C.c_float), ("refract_index", C.c_float),
                ("sharpness", C.c_float), ("_pad", C.c_int32)]


class PtxTexture(C.Structure):
    _fields_ = [("rgba", C.POINTER(C.c_float)), ("width", C.c_int32), ("height", C.c_int32)]


class PtxTriangle(C.Structure):
    _fields_ = [("a", C.c_float * 3), ("e1", C.c_float * 3), ("e2", C.c_float * 3),
                ("na", C.c_float * 3), ("nb", C.c_float * 3), ("nc", C.c_float * 3),
                ("ta", C.c_float * 2), ("tb", C.c_float * 2), ("tc", C.c_float * 2), ("material", C.c_int32)]


class PtxKDNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("left", C.c_int32), ("right", C.c_int32),
                ("first_tri", C.c_int32), ("num_tris", C.c_int32)]


class PtxObject(C.Structure):
    _fields_ = [("type", C.c_int32), ("first_material", C.c_int32), ("num_materials", C.c_int32),
                ("radius", C.c_float), ("position", C.c_float * 3),
                ("first_triangle", C.c_int32), ("num_triangles", C.c_int32),
                ("first_node", C.c_int32), ("num_nodes", C.c_int32),
                ("first_leaf_tri", C.c_int32), ("num_leaf_tris", C.c_int32), ("max_depth", C.c_int32)]


class PtxSceneDesc(C.Structure):
    _fields_ = [("num_objects", C.c_int32), ("objects", C.POINTER(PtxObject)),
                ("num_materials", C.c_int32), ("materials", C.POINTER(PtxMaterial)),
                ("num_textures", C.c_int32), ("textures", C.POINTER(PtxTexture)),
                ("num_triangles", C.c_int32), ("triangles", C.POINTER(PtxTriangle)),
                ("num_nodes", C.c_int32), ("nodes", C.POINTER(PtxKDNode)),
                ("num_leaf_tris", C.c_int32), ("leaf_tris", C.POINTER(C.c_int32))]


class PtxHit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("object", C.c_int32), ("prim", C.c_int32), ("t_bits", C.c_uint32),
                ("leaf_hash", C.c_uint32), ("n_inner", C.c_uint32), ("n_box", C.c_uint32),
                ("n_tri", C.c_uint32), ("n_leaf", C.c_uint32), ("point", C.c_float * 3),
                ("normal", C.c_float * 3), ("bary", C.c_float * 3)]


class PtxStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "paths", "n_inner", "n_box", "n_tri", "n_sphere", "n_leaf",
                                          "hit_mesh", "hit_sphere", "hit_textured",
                                          "lane_steps_active", "lane_steps_total",
                                          "trav_rays", "trav_inner", "trav_box", "trav_tri", "trav_leaf",
                                          "diag0", "diag1", "diag2", "diag3",
                                          "x_iters", "x_item_lanes", "x_half_lanes", "x_wait_lanes", "x_unit_steps", "x_units",
                                          "x_unit_phases", "x_turns", "sh_waves", "sh_lanes", "sh_branches",
                                          "sh_type0", "sh_type1", "sh_type2", "sh_type3",
                                          "seg_clk0", "seg_clk1", "seg_clk2", "seg_clk3")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PtxBsdfCase(C.Structure):
    _fields_ = [("dir", C.c_float * 3), ("normal", C.c_float * 3), ("point", C.c_float * 3),
                ("mat_type", C.c_int32), ("reflect_factor", C.c_float), ("refract_index", C.c_float),
                ("sharpness", C.c_float), ("u", C.c_float * 4), ("out_origin", C.c_float * 3),
                ("out_dir", C.c_float * 3), ("out_mask", C.c_float * 3)]


# Every symbol include/ptx_hip.h declares (the "not gpu" tests check that the library exports all of them).
PTX_SYMBOLS = [
    "ptx_last_error", "ptx_version", "ptx_device_count", "ptx_device_info", "ptx_ctx_create", "ptx_ctx_destroy",
    "ptx_scene_upload", "ptx_scene_free", "ptx_scene_device_bytes", "ptx_scene_wide_sah_objects", "ptx_render_init", "ptx_render_set_camera",
    "ptx_render_cleanup", "ptx_accum_clear", "ptx_render_samples", "ptx_render_sync", "ptx_accum_device_ptr",
    "ptx_render_sample_ahead", "ptx_accum_download", "ptx_accum_upload", "ptx_tonemap", "ptx_host_alloc", "ptx_host_free", "ptx_stats_get", "ptx_stats_reset",
    "ptx_wavefront_counts", "ptx_ctx_tune", "ptx_profile_enable", "ptx_profile_get", "ptx_timer_begin", "ptx_timer_end", "ptx_trace_rays", "ptx_trace_rays_queued", "ptx_generate_rays", "ptx_bsdf_eval", "ptx_radiance_sample", "ptx_selftest_rcp", "ptx_selftest_vec3",
]

_lib = None
_host = None
_oracle = None


def _load(path, what):
    if not os.path.exists(path):
        raise RuntimeError(f"{what} is not built: {path} missing (run __graft_entry__.build() or `make -C path-tracer_amd`)")
    return C.CDLL(path)  # RTLD_LOCAL: the reference-built pin library defines the same C++ names


def lib():
    """libptx_hip.so with argument/return types set."""
    global _lib
    if _lib is None:
        L = _load(os.path.join(_HERE, "libptx_hip.so"), "the HIP library")
        vp, i32, u64, f32 = C.c_void_p, C.c_int, C.c_uint64, C.c_float
        fp = C.POINTER(C.c_float)
        L.ptx_last_error.restype = C.c_char_p
        L.ptx_version.restype = C.c_char_p
        L.ptx_device_info.argtypes = [i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(u64)]
        L.ptx_ctx_create.argtypes = [i32, C.POINTER(vp)]
        L.ptx_ctx_destroy.argtypes = [vp]
        L.ptx_scene_upload.argtypes = [vp, C.POINTER(PtxSceneDesc), C.POINTER(vp)]
        L.ptx_scene_free.argtypes = [vp, vp]
        L.ptx_scene_device_bytes.argtypes = [vp]
        L.ptx_scene_device_bytes.restype = C.c_int64
        L.ptx_render_init.argtypes = [vp, C.POINTER(PtxCamera), vp]
        L.ptx_render_set_camera.argtypes = [vp, C.POINTER(PtxCamera)]
        L.ptx_render_cleanup.argtypes = [vp]
        L.ptx_accum_clear.argtypes = [vp]
        L.ptx_render_samples.argtypes = [vp, vp, i32, i32, u64, i32]
        L.ptx_render_sample_ahead.argtypes = [vp, vp, i32, u64, i32]
        L.ptx_render_sync.argtypes = [vp]
        L.ptx_accum_device_ptr.argtypes = [vp]
        L.ptx_accum_device_ptr.restype = vp
        L.ptx_accum_download.argtypes = [vp, fp]
        L.ptx_accum_upload.argtypes = [vp, fp]
        L.ptx_tonemap.argtypes = [vp, i32, f32, C.POINTER(C.c_ubyte)]
        L.ptx_stats_get.argtypes = [vp, C.POINTER(PtxStats)]
        L.ptx_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(C.c_void_p)]
        L.ptx_host_free.argtypes = [vp, vp]
        L.ptx_stats_reset.argtypes = [vp]
        L.ptx_wavefront_counts.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ptx_ctx_tune.argtypes = [vp, C.c_char_p, i32]
        L.ptx_selftest_rcp.argtypes = [vp, C.c_float, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.ptx_profile_enable.argtypes = [vp, i32]
        L.ptx_profile_get.argtypes = [vp, fp, C.POINTER(C.c_int32), fp, C.POINTER(C.c_int32)]
        L.ptx_timer_begin.argtypes = [vp]
        L.ptx_timer_end.argtypes = [vp, fp]
        L.ptx_trace_rays.argtypes = [vp, vp, i32, fp, fp, i32, C.POINTER(PtxHit)]
        L.ptx_trace_rays_queued.argtypes = [vp, vp, i32, fp, fp, i32, C.POINTER(PtxHit)]
        L.ptx_generate_rays.argtypes = [vp, C.POINTER(PtxCamera), i32, u64, fp, fp]
        L.ptx_bsdf_eval.argtypes = [vp, i32, C.POINTER(PtxBsdfCase)]
        L.ptx_radiance_sample.argtypes = [vp, vp, i32, u64, fp]
        L.ptx_selftest_vec3.argtypes = [vp, i32, fp, fp, fp, fp]
        _lib = L
    return _lib


def host():
    """libpt_host.so (C++ host layer) with argument/return types set."""
    global _host
    if _host is None:
        lib()
        H = _load(os.path.join(_HERE, "libpt_host.so"), "the host library")
        vp, i32 = C.c_void_p, C.c_int
        H.pth_last_error.restype = C.c_char_p
        H.pth_set_asset_root.argtypes = [C.c_char_p]
        H.pth_set_seed.argtypes = [C.c_uint64]
        H.pth_set_look_ahead.argtypes = [i32]
        H.pth_look_ahead.restype = i32
        H.pth_checkpoint_save.argtypes = [C.c_char_p, C.c_int32]
        H.pth_checkpoint_load.argtypes = [C.c_char_p, C.POINTER(C.c_int32)]
        H.pth_autosave_due.argtypes = [C.c_int32, C.c_int32]
        H.pth_checkpoint_write_file.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(C.c_float)]
        H.pth_checkpoint_read_file.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.c_int64]
        H.pth_autosave_name.argtypes = [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
        H.pth_scene_load.argtypes = [C.c_char_p, i32, i32, i32, i32, i32, i32]
        H.pth_scene_desc.restype = C.POINTER(PtxSceneDesc)
        H.pth_scene_tri_abc.restype = C.POINTER(C.c_float)
        H.pth_scene_device.restype = vp
        H.pth_camera_get.argtypes = [C.POINTER(PtxCamera)]
        H.pth_camera_control.argtypes = [C.c_char_p, C.c_float]
        H.pth_log_lines.argtypes = [C.c_char_p, i32]
        H.pth_context.restype = vp
        H.pth_run_iteration.argtypes = [i32, C.POINTER(C.c_ubyte)]
        H.pth_render_samples.argtypes = [i32, i32, i32]
        H.pth_kd_build.argtypes = [C.POINTER(C.c_float), i32, C.POINTER(C.POINTER(PtxKDNode)),
                                   C.POINTER(C.POINTER(C.c_int32)), C.POINTER(i32), C.POINTER(i32)]
        H.pth_free.argtypes = [vp]
        H.pth_fmedian.argtypes = [C.POINTER(C.c_float), i32]
        H.pth_fmedian.restype = C.c_float
        H.pth_load_image_float.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                           C.POINTER(C.POINTER(C.c_float))]
        H.pth_write_png.argtypes = [C.c_char_p, i32, i32, i32, C.POINTER(C.c_ubyte), i32]
        H.pth_write_pfm.argtypes = [C.c_char_p, i32, i32, C.POINTER(C.c_float)]
        H.pth_set_asset_root(REPO_ROOT.encode())
        _host = H
    return _host


def check(rc, who="ptx"):
    """Raise with the library's message on a negative status."""
    if rc is not None and rc < 0:
        msg = lib().ptx_last_error().decode(errors="replace")
        hmsg = host().pth_last_error().decode(errors="replace") if _host is not None else ""
        raise RuntimeError(f"{who} failed ({rc}): {msg} {hmsg}".strip())
    return rc


def gpu_available():
    """True if the HIP library sees at least one device (no torch involved)."""
    try:
        return lib().ptx_device_count() > 0
    except Exception:
        return False


def load_scene(name_or_path, width=0, height=0, textures=-1, preview=0, aabb=-1, bounces=-1, upload=True,
               car_tris=None):
    """parseJsonScene + sceneState.load through the host layer. Returns (desc pointer, PtxCamera)."""
    H = host()
    path = name_or_path
    if not os.path.isabs(path) and not os.path.exists(path):
        path = os.path.join(REPO_ROOT, "scenes", f"scene_{name_or_path}.json")
    H.pth_set_device_upload(1 if upload else 0)
    # (the host keeps the budget of the procedural stand-in car between loads: always say it, so that a reduced car of one
    # caller does not become the next caller's headline mesh)
    H.pth_set_procedural_car_triangles(int(car_tris) if car_tris is not None else 400000)
    rc = H.pth_scene_load(path.encode(), width, height, textures, preview, aabb, bounces)
    if rc != PTX_OK:
        raise RuntimeError(f"scene load failed: {H.pth_last_error().decode(errors='replace')} / "
                           f"{lib().ptx_last_error().decode(errors='replace')}")
    cam = PtxCamera()
    check(H.pth_camera_get(C.byref(cam)), "pth_camera_get")
    return H.pth_scene_desc(), cam
