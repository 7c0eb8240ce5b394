#!/usr/bin/env python3
"""Headless multi-GPU render driver: one process per GPU, samples split across ranks, the float
accumulation buffers summed with one all-reduce (RCCL over xGMI), image written by rank 0.

    python path-tracer_amd/render.py --scene lamborghini --width 1920 --height 1080 --spp 1000 \
           --gpus 8 --out car.png                       # one all-reduce at the end
    python path-tracer_amd/render.py --scene sky_map --width 3840 --height 2160 --spp 1024 \
           --gpus 8 --every 128 --out sky.png           # progressive: a reduced image every 128 spp

With --gpus N > 1 and no RANK in the environment the script starts N ranks of itself through
torch.distributed.run (as a child process; this parent never touches a GPU). Under torchrun it
reads RANK / LOCAL_RANK / WORLD_SIZE. The reference has no multi-GPU path (reference:
src/main.cpp:31-32); the render loop per rank is the reference's (qt/cudarunthread.cpp:19-53) with
runCUDA replaced by ptx_render_samples on the rank's own sample range (spp_split.py).

--backend gloo --all-on-device 0 puts every rank on GPU 0 and reduces over gloo: the N > 1 code
path on a one-GPU box (tests/test_multi_gpu_render.py). PyTorch is plumbing here: the device buffer
the collective runs on, and torch.distributed.
"""
import argparse
import ctypes as C
import importlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scene", default="lamborghini", help="scene name (scenes/scene_<name>.json) or a path")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=64, help="total samples per pixel over all GPUs")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--every", type=int, default=0, help="progressive: all-reduce a copy every K global samples (0: once, at the end)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--textures", type=int, default=-1)
    ap.add_argument("--bounces", type=int, default=-1)
    ap.add_argument("--out", default="", help="tone-mapped PNG (mirrored like the reference's viewer), written by rank 0")
    ap.add_argument("--out-accum", default="", help="reduced float accumulation buffer as .npy (rank 0)")
    ap.add_argument("--progress-dir", default="", help="with --every: rank 0 writes <dir>/<spp>spp.png after every round")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--all-on-device", type=int, default=-1, help="development: every rank uses this one GPU")
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """The parent of an N-rank job: start torch.distributed.run as a CHILD process and hand its exit
    code on. Nothing here touches a GPU."""
    port = args.master_port or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def tonemap_host(img, spp, gamma):
    """writeImageToPBO on the host for progressive previews (reference: src/kernel.cu:40-54)."""
    import numpy as np
    rgb = np.clip(img[:, :3] / np.float32(spp), 0.0, 1.0).astype(np.float32)
    out = np.zeros((img.shape[0], 4), np.uint8)
    out[:, :3] = (np.float32(255.0) * np.power(rgb, np.float32(gamma))).astype(np.uint8)
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    device = args.all_on_device if args.all_on_device >= 0 else local_rank

    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist
    pt = importlib.import_module("path-tracer_amd")
    sp = importlib.import_module("path-tracer_amd.spp_split")
    L, H = pt.lib(), pt.host()
    if not pt.gpu_available():
        raise SystemExit("render.py needs an MI355X: libptx_hip.so sees no HIP device (there is no CPU path)")
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    H.pth_set_device(device)
    H.pth_set_seed(args.seed)
    desc, cam = pt.load_scene(args.scene, width=args.width, height=args.height, textures=args.textures, bounces=args.bounces)
    ctx, scene = H.pth_context(), H.pth_scene_device()
    n_pix = cam.width * cam.height
    # the accumulation buffer is torch's, so the collective runs on it in place (gloo: staged through the host)
    accum = torch.zeros((n_pix, 4), dtype=torch.float32, device=f"cuda:{device}")
    torch.cuda.synchronize()
    pt.check(L.ptx_render_init(ctx, C.byref(cam), C.c_void_p(accum.data_ptr())), "ptx_render_init")

    def render_fn(first, count):
        pt.check(L.ptx_render_samples(ctx, scene, first, count, args.seed, 0), "ptx_render_samples")

    def sync_fn():
        pt.check(L.ptx_render_sync(ctx), "ptx_render_sync")

    def reduce_tensor(t):
        # gloo has no device path: stage through the host (rehearsal only; RCCL reduces in place)
        if world > 1 and args.backend == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
            torch.cuda.synchronize()
        elif world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()

    def write_png(path, rgba8):
        buf = np.ascontiguousarray(rgba8)
        rc = H.pth_write_png(path.encode(), cam.width, cam.height, 4, buf.ctypes.data_as(C.POINTER(C.c_ubyte)), 1)
        if rc != 0:
            raise RuntimeError(H.pth_last_error().decode(errors="replace"))

    t0 = time.perf_counter()
    if args.every > 0:
        image = accum.clone()
        done = 0
        while done < args.spp:
            n = min(args.every, args.spp - done)
            first, count = sp.sample_range(rank, world, n, done)
            if count > 0:
                render_fn(first, count)
            sync_fn()
            image.copy_(accum)
            torch.cuda.synchronize()
            reduce_tensor(image)
            done += n
            if rank == 0:
                print(f"[render] {done}/{args.spp} spp reduced over {world} rank(s), {time.perf_counter() - t0:.2f} s", flush=True)
                if args.progress_dir:
                    os.makedirs(args.progress_dir, exist_ok=True)
                    write_png(os.path.join(args.progress_dir, f"{done}spp.png"), tonemap_host(image.cpu().numpy(), done, cam.gamma_corr))
        accum.copy_(image)          # every rank now holds the whole image in the buffer the context tone-maps
        torch.cuda.synchronize()
    else:
        first, count = sp.sample_range(rank, world, args.spp)
        if count > 0:
            render_fn(first, count)
        sync_fn()                   # the kernels add into accum on the context's stream: finish before the collective
        reduce_tensor(accum)
    elapsed = time.perf_counter() - t0

    if rank == 0:
        st = pt.PtxStats()
        pt.check(L.ptx_stats_get(ctx, C.byref(st)))
        print(f"[render] {args.spp} spp of {cam.width}x{cam.height} on {world} rank(s): {elapsed:.2f} s "
              f"(rank 0 traced {st.rays / 1e6:.1f} Mrays)", flush=True)
        if args.out_accum:
            np.save(args.out_accum, accum.cpu().numpy())
        if args.out:
            rgba8 = np.zeros((n_pix, 4), np.uint8)
            pt.check(L.ptx_tonemap(ctx, max(1, args.spp), cam.gamma_corr, rgba8.ctypes.data_as(C.POINTER(C.c_ubyte))), "ptx_tonemap")
            write_png(args.out, rgba8)
            print(f"[render] wrote {args.out}", flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    pt.check(L.ptx_render_cleanup(ctx))
    return 0


if __name__ == "__main__":
    sys.exit(main())
