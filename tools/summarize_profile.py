"""Condenses gpurun_out/prof_r01 (tools/profile_r01.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof_r01")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_wavefront"
latest = lambda d, suf: sorted(glob.glob(f"{P}/{d}/vm/*_{suf}.csv"), key=os.path.getmtime)[-1]
shutil.copy(latest("trace", "kernel_stats"), f"{ROOT}/profiles/{tag}_kernel_stats.csv")
out = {}
for d in ("fetch", "write", "l2", "sq", "ta"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(latest(d, "counter_collection"))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
        agg[k]["_vgpr"] = float(r["VGPR_Count"]); agg[k]["_lds"] = float(r["LDS_Block_Size"])
    for k, v in agg.items():
        if "wf_" in k:
            out.setdefault(k, {"dispatches": len(seen[k])}).update(v)
K = next(k for k in out if k.startswith("wf_traverse_coop_kernel<0"))
e = out[K]; n = e["dispatches"]
fetch = e["FETCH_SIZE"] * 1024 * 2 / n   # KB -> B; x2: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
write = e["WRITE_SIZE"] * 1024 / n
derived = {"hbm_side_fetch_bytes_per_launch": fetch, "hbm_side_write_bytes_per_launch": write,
           "l2_hit_rate": e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]),
           "valu_threads_per_inst": e["SQ_THREAD_CYCLES_VALU"] / e["SQ_ACTIVE_INST_VALU"],
           "valu_insts_per_launch": e["SQ_INSTS_VALU"] / n,
           "ta_busy_frac": e["TA_TA_BUSY_sum"] / 256 / (e["GRBM_GUI_ACTIVE"] / 8)}
json.dump({"command": "rocprofv3 --pmc <one counter block per pass> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (tools/profile_r01.sh)",
           "note": "per-kernel sums over the dispatches of one bench step (128 spp, 1920x1080 Lamborghini stand-in); <1,...> kernels = counting variant of the roofline leg",
           "kernels": out, "derived": {K: derived}}, open(f"{ROOT}/profiles/{tag}_pmc.json", "w"), indent=1)
json.dump({"workload": "lamborghini:1920x1080:spp128:tex1", "kernel": K, "traverse_hbm_bytes_per_launch": round(fetch + write),
           "how": "2 x FETCH_SIZE + WRITE_SIZE (KB -> B) per dispatch, separate --pmc passes; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                  "(gfx950 tallies 128-B requests at 64 B); counts L2-miss traffic incl. Infinity-Cache hits",
           "source": f"profiles/{tag}_pmc.json"}, open(f"{ROOT}/profiles/{tag}_hbm_traffic.json", "w"), indent=1)
print(K, json.dumps(derived, indent=1))
print(open(f"{ROOT}/profiles/{tag}_kernel_stats.csv").read()[:900])
