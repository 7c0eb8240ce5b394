#!/usr/bin/env python3
"""Rate of the REFERENCE'S OWN render loop through the drop-in boundary: kernelInit once, then runCUDA(pbo, scene, i) for
i = 1, 2, ... (qt/cudarunthread.cpp:14-56), one sample and one tone-mapped host image per call, against the batched
entry point the headline is measured on (ptx_render_samples, 128 samples per pass).

    python tools/reference_loop_rate.py [--scene lamborghini] [--width 1920 --height 1080] [--iters 200]

Prints, per mode: ms per iteration, samples/s, Mrays/s (device counters). Modes:
  loop        pth_run_iteration (= runCUDA): 1 spp + tone map + host copy, synchronous, as the reference's thread does
  render1     ptx_render_samples(1 spp) + ptx_render_sync only (no display)
  tonemap     ptx_tonemap alone (kernel + 4 bytes per pixel to the host)
  batch       ptx_render_samples(128 spp) per call (the fast path)
  sweep       pass time against samples per pass
  verify      the loop's float buffer after --iters calls against one pass of as many samples (bit for bit, at full size)
  ahead=N     sets the look-ahead of the following loops (pth_set_look_ahead; 1 = one pass per call)
"""
import argparse
import ctypes as C
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="lamborghini")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--modes", default="loop,render1,tonemap,batch")
    ap.add_argument("--pinned", type=int, default=0, help="1: the host image is page-locked memory (ptx_host_alloc)")
    args = ap.parse_args()
    import numpy as np
    pt = importlib.import_module("path-tracer_amd")
    L, H = pt.lib(), pt.host()
    if not pt.gpu_available():
        raise SystemExit("needs a HIP device")
    H.pth_set_device(0)
    desc, cam = pt.load_scene(args.scene, width=args.width, height=args.height)
    ctx = H.pth_context()
    pt.check(H.pth_kernel_init(), "kernelInit")
    n_pix = cam.width * cam.height
    if args.pinned:   # page-locked host image from the library (ptx_host_alloc = what kernelAllocPBO hands the render loop)
        raw = C.c_void_p()
        pt.check(L.ptx_host_alloc(ctx, n_pix * 4, C.byref(raw)), "ptx_host_alloc")
        pbo_ptr = C.cast(raw, C.POINTER(C.c_ubyte))
    else:
        pbo = np.zeros((n_pix, 4), np.uint8)
        pbo_ptr = pbo.ctypes.data_as(C.POINTER(C.c_ubyte))
    stats = pt.PtxStats()

    def rays():
        pt.check(L.ptx_stats_get(ctx, C.byref(stats)))
        return stats.rays

    def timed(name, fn, iters, spp_per_iter):
        for i in range(3):
            fn(i + 1)
        pt.check(L.ptx_render_sync(ctx))
        pt.check(L.ptx_stats_reset(ctx))
        t0 = time.perf_counter()
        for i in range(iters):
            fn(i + 4)
        pt.check(L.ptx_render_sync(ctx))
        dt = time.perf_counter() - t0
        r = rays()
        print(f"{name:8s} {1e3 * dt / iters:8.3f} ms per call  {iters * spp_per_iter * n_pix / dt / 1e9:7.3f} Gsamples/s  "
              f"{r / dt / 1e6:9.1f} Mrays/s", flush=True)

    def loop(i):
        rc = H.pth_run_iteration(i, pbo_ptr)
        if rc < 0:
            raise RuntimeError(H.pth_last_error().decode())

    def render1(i):
        pt.check(H.pth_render_samples(i - 1, 1, 0))
        pt.check(L.ptx_render_sync(ctx))

    def tonemap(i):
        pt.check(L.ptx_tonemap(ctx, i, C.c_float(0.5), pbo_ptr))

    def batch(i):
        pt.check(H.pth_render_samples((i - 1) * 128, 128, 0))

    print(f"{args.scene} {cam.width}x{cam.height}, host image {'pinned' if args.pinned else 'pageable'}")
    def sweep():   # pass time against samples per pass (what a look-ahead window pays)
        for n in (1, 2, 4, 8, 16, 32, 64, 128):
            def passn(i, n=n):
                pt.check(H.pth_render_samples((i - 1) * n, n, 0))
                pt.check(L.ptx_render_sync(ctx))
            timed(f"pass{n}", passn, max(4, args.iters // (4 * n) + 2), n)

    for m in args.modes.split(","):   # in the order given; a mode may be named twice (the first loop also pays for the buffers' growth)
        if m == "loop":
            timed("loop", loop, args.iters, 1)
        elif m == "looptimes":   # per-call times of one loop: the calls that trace a window against those that find their sample parked
            ts = []
            for i in range(1, args.iters + 1):
                t0 = time.perf_counter()
                loop(i)
                ts.append(1e3 * (time.perf_counter() - t0))
            big = [(i + 1, round(t, 2)) for i, t in enumerate(ts) if t > 1.5]
            small = sorted(t for t in ts if t <= 1.5)
            print("calls over 1.5 ms (iteration, ms):", big)
            print(f"the other {len(small)} calls: median {small[len(small) // 2]:.3f} ms, max {small[-1]:.3f} ms, sum {sum(small):.1f} ms; all: {sum(ts):.1f} ms")
        elif m == "render1":
            timed("render1", render1, args.iters, 1)
        elif m == "tonemap":
            timed("tonemap", tonemap, args.iters, 0)
        elif m == "sweep":
            sweep()
        elif m == "batch":
            timed("batch", batch, max(2, args.iters // 50), 128)
        elif m == "verify":   # the loop's buffer after args.iters calls against ONE ptx_render_samples call of as many samples, full size
            assert H.pth_kernel_init() == 0
            pt.check(L.ptx_stats_reset(ctx))
            for i in range(1, args.iters + 1):
                loop(i)
            a = np.empty((n_pix, 4), np.float32)
            pt.check(L.ptx_render_sync(ctx)); pt.check(L.ptx_accum_download(ctx, a.ctypes.data_as(C.POINTER(C.c_float))))
            r_loop = rays()
            assert H.pth_kernel_init() == 0
            pt.check(L.ptx_stats_reset(ctx))
            pt.check(H.pth_render_samples(0, args.iters, 0))
            b = np.empty((n_pix, 4), np.float32)
            pt.check(L.ptx_render_sync(ctx)); pt.check(L.ptx_accum_download(ctx, b.ctypes.data_as(C.POINTER(C.c_float))))
            diff = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=1).sum())
            print(f"verify: {args.iters} runCUDA calls (look-ahead {H.pth_look_ahead()}) against one pass of {args.iters} samples: "
                  f"{diff} of {n_pix} pixels differ; segments traced {r_loop} (loop, incl. ahead of the last call) / {rays()} (pass)")
            if diff:
                raise SystemExit(1)
        elif m.startswith("ahead="):
            H.pth_set_look_ahead(int(m[6:]))
            print(f"look-ahead {H.pth_look_ahead()}")
        else:
            raise SystemExit(f"unknown mode {m}")
    H.pth_shutdown()


if __name__ == "__main__":
    main()
