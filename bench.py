#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (whole job) on the Lamborghini scene at 1920x1080.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: SPP_PER_STEP samples of every pixel
(jittered primary ray -> KD traversal / intersection -> BSDF bounce loop -> accumulation), i.e.
what SPP_PER_STEP calls of the reference's runCUDA() do. Inputs (scene, textures, accumulation
buffer) are resident in HBM before the timed region. With N > 1 (torchrun, one rank per GPU)
every rank renders its own disjoint sample range per step (weak scaling, no data-path
collective) and ONE all-reduce (RCCL) of the float accumulation buffers closes the timed region.

The timed region ends when the accumulated buffer is resident on rank 0's HOST.

    python bench.py --gpus N --total-spp S      STRONG scaling: a step is one image of S samples per pixel, split
over the N ranks by sample index (rank g renders spp_split.sample_range(g, N, S)), all-reduced and brought to rank 0's
host inside the step; "scaling": "strong". Without the flag the run is the weak-scaling contract above, and a MEASURED
strong leg of 1000 spp (north_star's target: 125 per GPU at N = 8) is appended as config.time_to_1k_spp_s.

Prints one JSON line (rank 0) with the contract fields plus
  "roofline"      for the dominant kernel class of this run: wf_traverse_wide_kernel (traversal + ray/triangle
                  tests of the rays queued for a multi-node mesh) or wf_segment_kernel (ray generation, shading,
                  cheap objects); since the wide traversal they take about the same time, the other one is reported
                  under "other_kernel". Neither HBM nor MFMA bounds them (the trees are cache-resident, nothing
                  is a dense contraction), so `bound` names the unit that does: the larger of the vector-memory front end (texture-addresser busy) and VALU
                  issue, from the committed PMC passes (profiles/r03_pmc.json, recomputable from them
                  and profiles/r03_kernel_stats.csv alone; stamped with the commit and a hash of the kernel
                  sources: if that hash is not the tree's, or the live launch time is off by more than 10 %,
                  the view says "pmc": "stale" and carries no bound / frac) rescaled to the launch time measured live
                  here with HIP event pairs on the library's stream. `executed` = bytes and flops the
                  kernel really requests / computes, counted live on the same samples by its counting
                  twin; `logical` = the reference algorithm's figures of SURVEY.md 8(d), a side field;
                  `algorithmic` (in each kernel's view) = the HBM bytes the kernel cannot avoid (path state in /
                  out, queue entries, shading records, texels, parked radiance) over its launch time against 8 TB/s.
  config.reference_loop   (N = 1) the reference's OWN loop timed after everything else: runCUDA(pbo, scene, i) once per sample with a
                  tone-mapped image on the host after every call (ptx_render_sample_ahead behind it: DESIGN.md 3.6); not part of `value`.
  "cpu_baseline"  the CPU oracle (a straight multithreaded port, oracle/) timed on this box's host
                  cores on a bounded sample of the same workload (+ a 1-thread figure).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KNOWN_MISSING_TEXTURES = {"env.tga", "rim2.tga"}   # absent from the reference tree as well (.MISSING_LARGE_BLOBS)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector


L2_PEAK_GBS = 34500.0       # MI355X_MICROARCH.md: L2 (per XCD, aggregate) ~34.5 TB/s
PEAK_CLOCK_GHZ = 2.4        # MI355X_MICROARCH.md: max clock
PMC_FILE = os.path.join("profiles", "r03_pmc.json")


def csrc_sha1():
    """sha1 over the kernel sources and the build flags (the same function stamps the PMC file: tools/summarize_round.py)"""
    import glob
    import hashlib
    h = hashlib.sha1()
    base = os.path.join(ROOT, "path-tracer_amd")
    for f in sorted(glob.glob(os.path.join(base, "csrc", "*"))) + [os.path.join(base, "Makefile")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def logical_bytes_and_flops(st):
    """The REFERENCE algorithm's traversal terms of SURVEY.md 8(d) (a side field since round 2: the kernel does
    not execute them): 56 B per inner step, 40 B per triangle test of its leaves, 64 B per queued ray;
    22 flop per box test, 48 per triangle test."""
    b = 56 * st["trav_inner"] + 40 * st["trav_tri"] + 64 * st["trav_rays"]
    f = 22 * st["trav_box"] + 48 * st["trav_tri"]
    return b, f


def executed_bytes_and_flops(st):
    """What the wide traversal kernel itself requests and computes (its counting twin, variant 14, PtxStats.x_*):
    a 64-B wide-node record and four slab tests (22 flop each) per node visit; 320 B and eight triangle tests
    (48 flop each) per work unit; per traversal the queue entry, ray state in and the closest-hit record out (64 B)."""
    b = 64 * st["x_item_lanes"] + 320 * st["x_units"] + 64 * st["trav_rays"]
    f = 22 * 4 * st["x_item_lanes"] + 48 * 8 * st["x_units"]
    return b, f


def load_pmc(kernel_prefix):
    """(entry, file header) of the committed PMC passes (tools/profile_round.sh) for one kernel, matched on the kernel
    name up to its first template arguments; (None, header) if the file has no such kernel."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None, None
    head = {k: d.get(k) for k in ("round", "commit", "csrc_sha1", "compute_units")}
    for name, e in d.get("kernels", {}).items():
        if name.replace(" ", "").startswith(kernel_prefix.replace(" ", "")):
            e = dict(e)
            e["full_name"] = name
            return e, head
    return None, head


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="lamborghini")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp-per-step", type=int, default=128)
    ap.add_argument("--textures", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: every rank uses cuda:0 and the collective runs over gloo, to exercise the N>1 "
                         "code path on a one-GPU box (the reported number is then meaningless)")
    ap.add_argument("--cpu-spp", type=int, default=24)
    ap.add_argument("--total-spp", type=int, default=0,
                    help="strong scaling: a step is ONE image of this many samples per pixel split over the ranks (all-reduce and "
                         "download to rank 0 inside the step)")
    ap.add_argument("--no-strong-leg", action="store_true", help="weak mode: skip the measured 1000-spp strong leg")
    ap.add_argument("--no-loop-leg", action="store_true", help="skip the timing of the reference's own loop (runCUDA once per sample)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # stdout carries exactly one JSON line: the loader's progress chatter (C stdio) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    pt = importlib.import_module("path-tracer_amd")
    L, H = pt.lib(), pt.host()
    if not pt.gpu_available():
        raise SystemExit("bench.py needs an MI355X: libptx_hip.so sees no HIP device (there is no CPU path)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    H.pth_set_device(local_rank)
    H.pth_set_seed(args.seed)
    desc, cam = pt.load_scene(args.scene, width=args.width, height=args.height, textures=args.textures)
    # Texture files the scene names but the loader could not read. The reference tree itself lacks
    # textures/lp560/env.tga and rim2.tga (.MISSING_LARGE_BLOBS); anything beyond those two means the
    # workload is not the one config.workload names, and the run is refused.
    H.pth_texture_failures.argtypes = [C.c_char_p, C.c_int]
    tbuf = C.create_string_buffer(8192)
    n_tex_failed = H.pth_texture_failures(tbuf, 8192)
    tex_failed = sorted(os.path.basename(t.replace("\\", "/")) for t in tbuf.value.decode(errors="replace").split("\n") if t)
    if args.scene == "lamborghini" and not set(tex_failed) <= KNOWN_MISSING_TEXTURES:
        raise SystemExit(f"bench.py: texture files missing beyond the known set {sorted(KNOWN_MISSING_TEXTURES)}: {tex_failed}")
    ctx = H.pth_context()
    scene = H.pth_scene_device()
    cu_count = C.c_int(0)
    pt.check(L.ptx_device_info(local_rank, None, 0, C.byref(cu_count), None, None))
    CUS = cu_count.value or 256
    SIMDS = 4 * CUS
    n_pix = cam.width * cam.height
    n_tris = desc.contents.num_triangles

    # accumulation buffer owned by torch so the collective can run on it in place
    accum = torch.zeros((n_pix, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    pt.check(L.ptx_render_init(ctx, C.byref(cam), C.c_void_p(accum.data_ptr())), "ptx_render_init")

    spp = args.spp_per_step
    stats = pt.PtxStats()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    from spp_split import sample_range
    strong = args.total_spp > 0
    host_accum = np.empty((n_pix, 4), np.float32) if rank == 0 else None

    def strong_image(i, total):
        """One image of `total` samples per pixel: this rank's share of the sample range, ONE all-reduce, rank 0's host copy."""
        first, count = sample_range(rank, world, total, i * total)
        if count > 0:
            pt.check(L.ptx_render_samples(ctx, scene, first, count, args.seed, 0), "ptx_render_samples")
        pt.check(L.ptx_render_sync(ctx))
        if world > 1:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()
        if rank == 0:
            pt.check(L.ptx_accum_download(ctx, host_accum.ctypes.data_as(C.POINTER(C.c_float))), "ptx_accum_download")

    def step(i):
        if strong:
            accum.zero_()
            torch.cuda.synchronize()
            strong_image(i, args.total_spp)
            return
        # global sample indices: step-major, rank-minor, disjoint across ranks and steps
        first = (i * world + rank) * spp
        pt.check(L.ptx_render_samples(ctx, scene, first, spp, args.seed, 0), "ptx_render_samples")

    for i in range(args.warmup):
        step(i)
    pt.check(L.ptx_render_sync(ctx))
    accum.zero_()
    pt.check(L.ptx_stats_reset(ctx))
    barrier()

    kernel_ms = C.c_float(0)
    pt.check(L.ptx_profile_enable(ctx, 1))
    t0 = time.perf_counter()
    pt.check(L.ptx_timer_begin(ctx))
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    pt.check(L.ptx_timer_end(ctx, C.byref(kernel_ms)))     # also waits for the launches
    if not strong:
        if world > 1:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM)      # the one exchange step of the job
            torch.cuda.synchronize()
        if rank == 0:
            # "... to accumulated buffer resident on rank 0 host" (33 MB over PCIe, once per image)
            pt.check(L.ptx_accum_download(ctx, host_accum.ctypes.data_as(C.POINTER(C.c_float))), "ptx_accum_download")
    barrier()
    elapsed = time.perf_counter() - t0

    trav_ms, seg_ms = C.c_float(0), C.c_float(0)
    trav_n, seg_n = C.c_int32(0), C.c_int32(0)
    pt.check(L.ptx_profile_get(ctx, C.byref(trav_ms), C.byref(trav_n), C.byref(seg_ms), C.byref(seg_n)))
    pt.check(L.ptx_profile_enable(ctx, 0))
    pt.check(L.ptx_stats_get(ctx, C.byref(stats)))
    rays_local = torch.tensor([float(stats.rays), elapsed, float(kernel_ms.value)], dtype=torch.float64,
                              device=f"cuda:{local_rank}")
    if world > 1:
        rays_sum = rays_local.clone()
        dist.all_reduce(rays_sum, op=dist.ReduceOp.SUM)
        tmax = rays_local.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        total_rays, max_elapsed, max_kernel_ms = rays_sum[0].item(), tmax[1].item(), tmax[2].item()
    else:
        total_rays, max_elapsed, max_kernel_ms = float(stats.rays), elapsed, float(kernel_ms.value)

    # ---- measured strong-scaling leg (weak mode): north_star's 1000 spp split over the ranks, timed to rank 0's host ----
    strong_1k = None
    if not strong and not args.no_strong_leg:
        accum.zero_()
        barrier()
        ts = time.perf_counter()
        strong_image(1 << 10, 1000)      # sample indices far from the timed steps'
        barrier()
        strong_1k = time.perf_counter() - ts
        if world > 1:
            tmx = torch.tensor([strong_1k], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
            strong_1k = tmx[0].item()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline leg (rank 0): counters of the SAME samples ---------------------------------------------------
    # two counting passes: the default kernels' exact twin gives the REFERENCE algorithm's counts (logical), the wide
    # traversal's own counting twin (variant 14) what that kernel executes (PtxStats.x_*)
    def counting_pass(variant):
        pt.check(L.ptx_ctx_tune(ctx, b"trav_variant", variant))
        pt.check(L.ptx_stats_reset(ctx))
        for i in range(args.warmup, args.warmup + args.steps):
            first, count = sample_range(rank, world, args.total_spp, i * args.total_spp) if strong else ((i * world + rank) * spp, spp)
            if count > 0:
                pt.check(L.ptx_render_samples(ctx, scene, first, count, args.seed, 1))
        pt.check(L.ptx_render_sync(ctx))
        st = pt.PtxStats()
        pt.check(L.ptx_stats_get(ctx, C.byref(st)))
        pt.check(L.ptx_ctx_tune(ctx, b"trav_variant", -1))
        return st.as_dict()

    cd = counting_pass(-1)
    xd = counting_pass(14)
    tr = max(1, cd["trav_rays"])
    xb, xf = executed_bytes_and_flops(xd)
    lb, lf = logical_bytes_and_flops(cd)

    src_hash = csrc_sha1()

    def kernel_view(kernel, class_ms, class_n, algo_bytes_per_step, algo_formula):
        """bound / achieved / peak / frac of one kernel class: the larger of texture-addresser busy and VALU issue at
        2 cycles per wave64 instruction, from the committed PMC passes rescaled to the launch time measured live.
        The PMC file is used only if it was made from THIS tree's kernel sources and its launch time is within 10 %
        of the live one; else the view says "pmc": "stale" and carries no bound / frac."""
        n_launch = max(1, class_n)
        launch_ms = float(class_ms) / n_launch
        launch_s = max(launch_ms * 1e-3, 1e-12)
        algo_launch = algo_bytes_per_step * args.steps / n_launch
        view = {"kernel": kernel, "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "launch_ms": round(launch_ms, 4), "launches": class_n,
                "share_of_step": round(float(class_ms) / max(1e-9, float(kernel_ms.value)), 4), "pmc": None,
                # what the kernel cannot avoid moving to / from HBM, over its launch time, against the HBM rate
                "algorithmic": {"bytes_per_launch": round(algo_launch), "achieved": round(algo_launch / launch_s / 1e9, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(algo_launch / launch_s / 1e9 / HBM_PEAK_GBS, 4), "bytes": algo_formula}}
        pmc, head = load_pmc(kernel)
        if not (pmc and pmc.get("derived", {}).get("ta_busy_frac") and pmc.get("avg_ns_kernel_trace")):
            view["pmc"] = "absent" if head is None or pmc is None else "incomplete"
            return view
        prof_ms = pmc["avg_ns_kernel_trace"] * 1e-6
        if head.get("csrc_sha1") != src_hash or abs(prof_ms - launch_ms) > 0.10 * launch_ms:
            view["pmc"] = "stale"
            view["pmc_detail"] = (f"{PMC_FILE} (commit {head.get('commit')}): kernel sources {'differ from' if head.get('csrc_sha1') != src_hash else 'match'} "
                                  f"this tree's; launch time there {prof_ms:.3f} ms, here {launch_ms:.3f} ms")
            return view
        view["kernel"] = pmc["full_name"]
        d, c, nd = pmc["derived"], pmc["counters"], pmc["dispatches"]
        # (the committed passes are of the headline workload: for another scene the fractions are only indicative)
        scale = prof_ms / launch_ms if launch_ms > 0 else 1.0
        ta = d["ta_busy_frac"] * scale
        valu2 = d.get("valu_issue_frac_2cyc", 0.0) * scale
        # vector-instruction issue priced per instruction class at its MEASURED cost (tools/micro/valu_rate.hip; the class counters
        # and the one assumption about the unnamed rest are in tools/summarize_round.py); without the class pass: 2 cycles each
        valu_w = d.get("valu_issue_frac_weighted_est", 0.0) * scale
        valu = valu_w if valu_w > 0.0 else valu2
        clock = d.get("clock_ghz", PEAK_CLOCK_GHZ)
        view["traffic"] = round(d["hbm_side_bytes_per_launch"]) if "hbm_side_bytes_per_launch" in d else None
        if ta >= valu:
            view["bound"], view["unit"] = "vmem", f"Gcycle/s busy, summed over the {CUS} texture addressers (one per CU)"
            achieved, peak = c["TA_TA_BUSY_sum"] / nd / launch_s / 1e9, CUS * clock
        elif valu_w > 0.0:
            view["bound"] = "valu-issue"
            view["unit"] = (f"Gcycle/s of vector-instruction issue, summed over the {SIMDS} SIMDs: instruction counts by class (PMC) x the measured "
                            "issue cost of the class (2.33 / 3.85 / 7.3 cycles; the unnamed rest priced slow: an estimate)")
            achieved, peak = d["valu_issue_cycles_weighted_est"] / nd / launch_s / 1e9, SIMDS * clock
        else:
            view["bound"], view["unit"] = "valu-issue", "G wave64 VALU instructions/s (2 cycles each per SIMD)"
            achieved, peak = c["SQ_INSTS_VALU"] / nd / launch_s / 1e9, SIMDS * clock / 2.0
        view["achieved"], view["peak"], view["frac"] = round(achieved, 3), round(peak, 3), round(achieved / peak, 4)
        l2_gbs = c.get("TCP_TCC_READ_REQ_sum", 0.0) * 64.0 / nd / launch_s / 1e9
        hbm_gbs = d.get("hbm_side_bytes_per_launch", 0.0) / launch_s / 1e9
        view["pmc"] = {
            "source": f"{PMC_FILE} + profiles/{head.get('round')}_kernel_stats.csv (tools/profile_round.sh, commit {head.get('commit')}, same kernel sources "
                      f"as this tree; launch time there {prof_ms:.3f} ms, here {launch_ms:.3f} ms)",
            "clock_ghz": round(clock, 3),
            "ta_busy_frac": round(ta, 4),
            "valu_issue_frac_2cyc_per_inst": round(valu2, 4),
            "valu_issue_frac_4cyc_per_inst": round(d.get("valu_issue_frac_4cyc", 0.0) * scale, 4),
            # tools/micro/valu_rate.hip: only f32 add / mul / fma, mov, integer add and and / or / xor issue at the 2-cycle rate; compares,
            # selects, min / max, conversions, lane moves and packed f32 take ~1.65x as long. Estimate from the class counters:
            "valu_issue_frac_weighted_est": round(d.get("valu_issue_frac_weighted_est", 0.0) * scale, 4) or None,
            "valu_mix": d.get("valu_mix"),
            "active_lanes_per_valu_inst": round(d.get("active_lanes_per_valu_inst", 0.0), 2),
            "valu_useful_frac": round(valu2 * d.get("active_lanes_per_valu_inst", 0.0) / 64.0, 4),
            "l1_to_l2_read": {"achieved": round(l2_gbs, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(l2_gbs / L2_PEAK_GBS, 4)},
            "hbm_side": {"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4)},
            "vmem_wave_instructions_per_launch": round((c.get("SQ_INSTS_VMEM_RD", 0.0) + c.get("SQ_INSTS_VMEM_WR", 0.0)) / nd),
            "valu_wave_instructions_per_launch": round(c.get("SQ_INSTS_VALU", 0.0) / nd),
            "l2_hit_frac": round(d.get("l2_hit_frac", 0.0), 4),
            "wave_time_waiting_frac": round(d.get("wave_time_waiting_frac", 0.0), 4),
        }
        return view

    # Algorithmic HBM bytes per step (SURVEY 8d's per-ray terms, restated for this pipeline's data layout; the trees, the
    # cheap objects and the material tables are cache-resident and not counted):
    #   segment kernel: a ray that needs a traversal is written as a 64-B state record and read back after the traversal
    #   (128 B), plus a 4-B queue entry per mesh it enters; a shaded mesh hit reads its 64-B tri_shade record, a textured one a
    #   16-B texel; a finished path parks 12 B of radiance. Rays that never enter a multi-node mesh stay in LDS (pt_segloop.h).
    #   traversal kernel: 4-B queue entry + origin, direction, current best (36 B read) + the closest-hit words (28 B) per ray.
    maxb = cam.max_ray_bounces
    live = (C.c_int32 * (maxb + 1))()
    queued = (C.c_int32 * (maxb + 1))()
    pt.check(L.ptx_wavefront_counts(ctx, maxb, live, queued))      # the LAST pass (<= 128 samples of every pixel)
    last_first, last_count = sample_range(rank, world, args.total_spp, 0) if strong else (0, spp)
    wf_batch = int(os.environ.get("PTX_WF_BATCH", "128"))
    last_pass_spp = (last_count % wf_batch) or min(wf_batch, last_count) or 1
    per_step = (last_count / last_pass_spp)
    state_records = sum(live) * per_step
    queue_entries = sum(queued) * per_step
    k_steps = float(args.steps)
    seg_bytes = 128.0 * state_records + 4.0 * queue_entries + (64.0 * cd["hit_mesh"] + 16.0 * cd["hit_textured"] + 12.0 * cd["paths"]) / k_steps
    trav_bytes = 68.0 * cd["trav_rays"] / k_steps
    seg_formula = "128 B x state records written and read back + 4 B x queue entries + 64 B x mesh hits shaded + 16 B x textured hits + 12 B x paths"
    trav_formula = "68 B x traversals (queue entry, ray and current best in, closest hit out); tree and triangles are cache-resident"

    # The two kernel classes of a step (HIP event pairs around every launch; a traversal pair spans the wide kernel
    # and the strict kernel's retry launch behind it). Since the wide traversal they take about the same time on the
    # headline scene, so the dominant one is whichever is larger in THIS run and the other is reported beside it.
    seg_name = "wf_segment_loop_kernel<0," if os.environ.get("PTX_SEG_MODE", "1") != "0" else "wf_segment_kernel<0,"
    trav = kernel_view("wf_traverse_wide_kernel<0,", trav_ms.value, trav_n.value, trav_bytes, trav_formula) if trav_n.value > 0 else None
    seg = kernel_view(seg_name, seg_ms.value, seg_n.value, seg_bytes, seg_formula)
    trav_dominant = trav is not None and trav_ms.value >= seg_ms.value
    dom, other = (trav, seg) if trav_dominant else (seg, trav)
    trav_s = max(float(trav_ms.value) * 1e-3, 1e-12)
    # (a view whose PMC passes are absent or stale keeps bound / frac = null: its "algorithmic" and "executed" figures are live)
    xtr = max(1, xd["trav_rays"])
    roofline = dict(dom)
    roofline.update({
        "timing": "HIP event pairs around every launch on the library's stream, inside the timed region (their cost is in value)",
        "other_kernel": other,
        "traversals_per_launch": round(cd["trav_rays"] / max(1, trav_n.value)),
        # what the traversal kernel executes, counted live on the same samples by its counting twin
        "executed": {"bytes_per_traversal": round(xb / xtr, 1), "flops_per_traversal": round(xf / xtr, 1),
                     "vmem": {"achieved": round(xb / trav_s / 1e9, 1), "peak": L2_PEAK_GBS, "unit": "GB/s requested of L1/L2",
                              "frac": round(xb / trav_s / 1e9 / L2_PEAK_GBS, 4)},
                     "valu": {"achieved": round(xf / trav_s / 1e12, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(xf / trav_s / 1e12 / VALU_PEAK_TFLOPS, 4)},
                     "per_traversal": {"wide_node_visits": round(xd["x_item_lanes"] / xtr, 2), "work_units": round(xd["x_units"] / xtr, 2),
                                       "triangle_slots_tested": round(8 * xd["x_units"] / xtr, 1),
                                       "stacked_children_dropped_at_the_pop": round(xd["x_wait_lanes"] / xtr, 2),
                                       "lanes_visiting_per_iteration": round(xd["x_item_lanes"] / max(1, xd["x_iters"]), 1),
                                       "units_per_step": round(xd["x_units"] / max(1, xd["x_unit_steps"]), 1)},
                     "rays_retried_by_the_exact_kernel": xd["x_half_lanes"], "of_them_for_a_full_stack": xd["diag3"]},
        # the reference algorithm's own traffic and work on the same rays, which the kernels mostly avoid
        "logical": {"bytes_per_traversal": round(lb / tr, 1), "flops_per_traversal": round(lf / tr, 1),
                    "gbs": round(lb / trav_s / 1e9, 1) if trav is not None else 0.0,
                    "tflops": round(lf / trav_s / 1e12, 2) if trav is not None else 0.0,
                    "per_traversal": {k: round(cd[k] / tr, 3) for k in ("trav_inner", "trav_box", "trav_tri", "trav_leaf")}},
        "segment_kernel": {"launch_ms": round(float(seg_ms.value) / max(1, seg_n.value), 4), "launches": seg_n.value},
        "step_kernel_ms": round(float(kernel_ms.value) / args.steps, 3),
        "note": "bound = the unit with the highest utilisation in the committed PMC passes, rescaled to this run's launch time: "
                "the vector-memory front end (texture-addresser busy) against vector-instruction issue priced per class at its "
                "measured cost (pmc.valu_issue_frac_weighted_est; the 2-cycle figure of MI355X_MICROARCH.md is printed beside it). Not HBM: the trees are cache-resident and the path state streams at a third of the HBM "
                "rate (pmc.hbm_side; traffic = measured L2-miss side per launch; `algorithmic` = unavoidable HBM bytes over the launch time "
                "against 8 TB/s). pmc.valu_issue_frac_weighted_est prices the instruction classes at their measured issue cost "
                "(tools/micro/valu_rate.hip: compares, selects, min / max, conversions take 1.65x an fma): by it the vector unit is "
                "the first limit of the traversal and ties with the vector-memory front end in the segment kernel (DESIGN.md 3.4)",
    })

    # ---- CPU baseline leg (rank 0): the oracle, bounded sample of the same workload ------------------
    cpu = None
    if not args.no_cpu_baseline and world == 1:   # on rank 0 at N = 1 only
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        # A one-GPU job's share of the host is 16 cores (the box's rule for worker pools), however many
        # the machine shows: threads = min(16, visible). Both numbers and a 1-thread figure are reported.
        visible = os.cpu_count() or 1
        try:
            visible = max(visible, len(os.sched_getaffinity(0)))
        except (AttributeError, OSError):
            pass
        threads = min(16, visible)

        def cpu_run(spp, nthreads):
            acc = np.zeros((n_pix, 4), np.float32)
            ost = pt.PtxStats()
            tc = time.perf_counter()
            rc = orc.lib().orc_render(desc, C.byref(cam), 0, spp, args.seed, orc.fptr(acc), nthreads, C.byref(ost))
            assert rc == 0
            return int(ost.rays), time.perf_counter() - tc

        rays_n, dt_n = cpu_run(args.cpu_spp, threads)
        rays_1, dt_1 = cpu_run(1, 1)
        cpu = {"value": round(rays_n / dt_n / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
               "host_cores_visible": visible,
               "one_thread": {"value": round(rays_1 / dt_1 / 1e6, 3), "unit": "Mrays/s",
                              "sample": f"1 spp of the full frame ({rays_1} rays, {dt_1:.1f} s)"},
               "sample": f"{args.cpu_spp} spp of the full {cam.width}x{cam.height} frame ({rays_n} rays, {dt_n:.1f} s) "
                         f"on {threads} threads (a one-GPU job's share of the {visible} cores visible)"}

    # ---- the reference's OWN loop (N = 1, after everything else): runCUDA(pbo, scene, i) once per sample, one tone-mapped image on
    # the host per call (qt/cudarunthread.cpp:21-41), through the host layer's default look-ahead and a page-locked display buffer
    ref_loop = None
    if world == 1 and not strong and not args.no_loop_leg:
        raw = C.c_void_p()
        pt.check(L.ptx_host_alloc(ctx, n_pix * 4, C.byref(raw)), "ptx_host_alloc")
        pbo = C.cast(raw, C.POINTER(C.c_ubyte))
        pt.check(H.pth_kernel_init(), "kernelInit")   # the render side again, with the library's own accumulation buffer
        calls = 128
        for i in range(1, 4):
            assert H.pth_run_iteration(i, pbo) == i, H.pth_last_error()
        pt.check(L.ptx_render_sync(ctx))
        pt.check(L.ptx_stats_reset(ctx))
        tl = time.perf_counter()
        for i in range(4, 4 + calls):
            assert H.pth_run_iteration(i, pbo) == i, H.pth_last_error()
        pt.check(L.ptx_render_sync(ctx))
        dl = time.perf_counter() - tl
        pt.check(L.ptx_stats_get(ctx, C.byref(stats)))
        ref_loop = {"ms_per_call": round(1e3 * dl / calls, 3), "calls": calls, "look_ahead_samples": int(H.pth_look_ahead()),
                    "mrays_per_s_traced": round(stats.rays / dl / 1e6, 1), "samples_per_s": round(n_pix * calls / dl, 1),
                    "what": "runCUDA(pbo, scene, i) once per sample, a tone-mapped image on the host after every call (page-locked "
                            "buffer from ptx_host_alloc); rays traced ahead for calls beyond the last one are counted"}
        pt.check(L.ptx_host_free(ctx, raw))

    out = {
        "metric": "Mrays/s (whole node), Lamborghini 1920x1080", "value": round(total_rays / max_elapsed / 1e6, 2),
        "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(max_elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"scenes/scene_{args.scene}.json at {cam.width}x{cam.height}, textures "
                               f"{'on' if args.textures else 'off'} ({desc.contents.num_textures} texture files loaded, {n_tex_failed} absent"
                               f"{': ' + ', '.join(tex_failed) if tex_failed else ''}), 5 bounces; lp560.obj absent -> procedural stand-in car",
                   "triangles": n_tris,
                   **({"total_spp_per_step": args.total_spp, "spp_per_step_per_gpu": round(args.total_spp / world, 2),
                       "samples_per_s": round(n_pix * args.total_spp * args.steps / max_elapsed, 1),
                       "time_to_total_spp_s": round(max_elapsed / args.steps, 4),
                       "time_to_1k_spp_s": round(max_elapsed / args.steps * 1000.0 / args.total_spp, 4),
                       "time_to_1k_spp_is": "scaled from the measured image of total_spp_per_step samples (all-reduce and download to rank 0 included)"}
                      if strong else
                      {"spp_per_step_per_gpu": spp, "samples_per_s": round(n_pix * spp * args.steps * world / max_elapsed, 1),
                       "time_to_1k_spp_s": round(strong_1k, 4) if strong_1k is not None else round(1000.0 / (spp * args.steps * world / max_elapsed), 2),
                       "time_to_1k_spp_is": ("MEASURED: one image of 1000 spp split over the ranks by sample index, one all-reduce, on rank 0's host"
                                             if strong_1k is not None else "extrapolated from the weak-scaling steps")}),
                   **({"reference_loop": ref_loop} if ref_loop is not None else {}),
                   "parallelism": f"spp-split x{world}"},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    sys.stdout.flush()
    C.CDLL(None).fflush(None)
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
