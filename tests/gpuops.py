"""Thin numpy wrappers over the C ABI entry points (GPU side of the parity tests)."""
import ctypes as C
import importlib

import numpy as np

pt = importlib.import_module("path-tracer_amd")
from orc import fptr, hits_to_arrays  # noqa: E402  (array helpers only; no oracle compute)


def trace_queued(ctx, scene, origins, dirs, obj):
    """The same rays through the production traversal kernel (one multi-node mesh)."""
    n = origins.shape[0]
    hits = (pt.PtxHit * n)()
    o = np.ascontiguousarray(origins, np.float32)
    d = np.ascontiguousarray(dirs, np.float32)
    pt.check(pt.lib().ptx_trace_rays_queued(ctx, scene, n, fptr(o), fptr(d), obj, hits), "ptx_trace_rays_queued")
    return hits_to_arrays(hits, n)


def trace(ctx, scene, origins, dirs, obj=-1):
    n = origins.shape[0]
    hits = (pt.PtxHit * n)()
    o = np.ascontiguousarray(origins, np.float32)
    d = np.ascontiguousarray(dirs, np.float32)
    pt.check(pt.lib().ptx_trace_rays(ctx, scene, n, fptr(o), fptr(d), obj, hits), "ptx_trace_rays")
    return hits_to_arrays(hits, n)


def generate_rays(ctx, cam, sample, seed):
    n = cam.width * cam.height
    o = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    pt.check(pt.lib().ptx_generate_rays(ctx, C.byref(cam), sample, seed, fptr(o), fptr(d)), "ptx_generate_rays")
    return o, d


def render_init(ctx, cam, external=None):
    pt.check(pt.lib().ptx_render_init(ctx, C.byref(cam), external), "ptx_render_init")


def render(ctx, scene, cam, first, count, seed, stats=False, clear=True, batch=None):
    L = pt.lib()
    if clear:
        pt.check(L.ptx_accum_clear(ctx))
        pt.check(L.ptx_stats_reset(ctx))
    batch = batch or count
    s = first
    while s < first + count:
        n = min(batch, first + count - s)
        pt.check(L.ptx_render_samples(ctx, scene, s, n, seed, 1 if stats else 0), "ptx_render_samples")
        s += n
    pt.check(L.ptx_render_sync(ctx))
    acc = np.empty((cam.width * cam.height, 4), np.float32)
    pt.check(L.ptx_accum_download(ctx, fptr(acc)))
    st = pt.PtxStats()
    pt.check(L.ptx_stats_get(ctx, C.byref(st)))
    return acc, st


def radiance_sample(ctx, scene, cam, sample, seed):
    rgb = np.empty((cam.width * cam.height, 3), np.float32)
    pt.check(pt.lib().ptx_radiance_sample(ctx, scene, sample, seed, fptr(rgb)), "ptx_radiance_sample")
    return rgb
