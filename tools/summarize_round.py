"""Condense a tools/profile_round.sh run into the two files bench.py and the reviewers read:
    <dst>/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (copied as written)
    <dst>/<tag>_pmc.json           per-kernel PMC sums over the dispatches of ONE bench step + derived fractions, stamped with
                                   the commit id and a hash of the kernel sources, so bench.py can tell a stale file
usage: python tools/summarize_round.py gpurun_out/prof_r03 r03 <commit id> [dst = profiles]"""
import collections, csv, glob, hashlib, json, os, shutil, sys

src, tag, commit = sys.argv[1], sys.argv[2], sys.argv[3]
dst = sys.argv[4] if len(sys.argv) > 4 else "profiles"
os.makedirs(dst, exist_ok=True)
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def csrc_sha1():
    """sha1 over the kernel sources and the build flags (the same function is in bench.py)"""
    h = hashlib.sha1()
    base = os.path.join(ROOT, "path-tracer_amd")
    for f in sorted(glob.glob(os.path.join(base, "csrc", "*"))) + [os.path.join(base, "Makefile")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def device_units():
    try:
        import importlib, ctypes as C
        sys.path.insert(0, ROOT)
        pt = importlib.import_module("path-tracer_amd")
        cu = C.c_int()
        pt.check(pt.lib().ptx_device_info(0, None, 0, C.byref(cu), None, None))
        return cu.value
    except Exception:
        return 256


CUS = device_units()
SIMDS = 4 * CUS
stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
avg_ns = {}
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        avg_ns[r["Name"].split("(")[0].replace("void ", "")] = (float(r["AverageNs"]), int(r["Calls"]))

sums = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(os.path.join(src, "*", "**", "*_counter_collection.csv"), recursive=True)):
    pass_name = os.path.relpath(f, src).split(os.sep)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("wf_"):
            continue
        sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][pass_name].add(r["Dispatch_Id"])
        sums[k]["_vgpr"] = float(r["VGPR_Count"]); sums[k]["_sgpr"] = float(r["SGPR_Count"]); sums[k]["_lds"] = float(r["LDS_Block_Size"])

out = {"round": tag, "commit": commit, "csrc_sha1": csrc_sha1(), "compute_units": CUS,
       "command": "tools/profile_round.sh: rocprofv3 --pmc <one block per pass> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-strong-leg --no-loop-leg; "
                  "kernel_stats from rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strong-leg --no-loop-leg",
       "note": "counter sums over the dispatches of one bench step (128 spp, 1920x1080 Lamborghini stand-in, real lp560 textures); keys are the "
               "FULL kernel names incl. template arguments; <1,...> kernels are the counting variants of bench.py's roofline leg. "
               "derived: see tools/summarize_round.py",
       "kernels": {}}
for k, c in sums.items():
    n = max(len(v) for v in disp[k].values())
    e = {"dispatches": n, "counters": dict(sorted(c.items()))}
    if k in avg_ns:
        e["avg_ns_kernel_trace"], e["calls_kernel_trace"] = avg_ns[k]
    d = {}
    g = c.get("GRBM_GUI_ACTIVE")
    if g:
        cyc = g / 8.0                      # GRBM_GUI_ACTIVE is summed over the 8 XCDs: chip cycles over all dispatches
        d["cycles_per_launch"] = cyc / n
        if k in avg_ns:
            d["clock_ghz"] = cyc / n / avg_ns[k][0]
        if "TA_TA_BUSY_sum" in c:
            d["ta_busy_frac"] = c["TA_TA_BUSY_sum"] / (CUS * cyc)          # one texture addresser per CU
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            d["l1_tag_lookups_per_cu_cycle"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (CUS * cyc)
        if "SQ_ACTIVE_INST_VALU" in c:
            # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles (SIMD-32), 4 for a wave alone on its SIMD.
            d["valu_issue_frac_2cyc"] = c["SQ_INSTS_VALU"] * 2.0 / (SIMDS * cyc)
            d["valu_issue_frac_4cyc"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * cyc)
    if g and c.get("SQ_INSTS_VALU") and "SQ_INSTS_VALU_FMA_F32" in c:
        # tools/micro/valu_rate.hip (2 ... 6 waves per SIMD, shader-clock ticks and wall clock agree): f32 add / mul / fma (and mov,
        # integer add, and / or / xor, right shifts) issue every 2.33 cycles, nearly everything else (min / max, compares, selects,
        # conversions, left shifts, bit-field, integer multiply-add, lane moves, packed f32) every 3.85, transcendentals every 7.3.
        # The class counters name only part of the stream: add + mul + fma are priced fast, transcendentals as such, integer work
        # half and half, and the UNNAMED rest (mov, compare, select, min / max, conversions ...) slow. An estimate built from measured
        # counts and measured costs; the split of the unnamed rest is the assumption (all of it fast would give the lower figure).
        CY_FAST, CY_SLOW, CY_TRANS = 2.33, 3.85, 7.3
        n_all = c["SQ_INSTS_VALU"]
        fast = c.get("SQ_INSTS_VALU_ADD_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) + c["SQ_INSTS_VALU_FMA_F32"]
        trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0)
        i32 = c.get("SQ_INSTS_VALU_INT32", 0)
        rest = max(0.0, n_all - fast - trans - i32)
        cyc_est = fast * CY_FAST + trans * CY_TRANS + i32 * 0.5 * (CY_FAST + CY_SLOW) + rest * CY_SLOW
        d["valu_mix"] = {"f32_add_mul_fma": fast / n_all, "transcendental": trans / n_all, "int32": i32 / n_all, "cvt": c.get("SQ_INSTS_VALU_CVT", 0) / n_all}
        d["valu_issue_cycles_weighted_est"] = cyc_est
        d["valu_issue_frac_weighted_est"] = cyc_est / (SIMDS * g / 8.0)
        d["valu_issue_frac_all_fast"] = n_all * CY_FAST / (SIMDS * g / 8.0)
    if c.get("SQ_ACTIVE_INST_VALU"):
        d["active_lanes_per_valu_inst"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
    if c.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in c:
        d["wave_time_waiting_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_WAVE_CYCLES") and g:
        d["waves_per_simd"] = 4.0 * c["SQ_WAVE_CYCLES"] / (SIMDS * g / 8.0)   # SQ_WAVE_CYCLES counts wave-resident QUAD-cycles (MI355X_MICROARCH.md); average over the launch incl. ramp and tail
    if "TCC_HIT_sum" in c:
        d["l2_hit_frac"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE reads half the bytes of a wide read on gfx950 -> doubled; units of 1 KiB
        d["hbm_side_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / n
    if "TCP_TCC_READ_REQ_sum" in c and k in avg_ns:
        d["l1_to_l2_read_gbs"] = c["TCP_TCC_READ_REQ_sum"] * 64.0 / n / avg_ns[k][0]      # 64 B per request; B/ns = GB/s
    e["derived"] = d
    out["kernels"][k] = e
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
for k, e in out["kernels"].items():
    print(k, json.dumps({a: (round(b, 4) if isinstance(b, float) else b) for a, b in e["derived"].items()}))
