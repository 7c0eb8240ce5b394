"""Sum rocprofv3 --pmc CSV output per kernel over the dispatches of a run.
usage: python tools/pmc_sum.py <dir with pass subdirs> [kernel substring]"""
import csv, glob, collections, json, os, sys
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "wf_traverse"
out = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if want not in k:
            continue
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add((f, r["Dispatch_Id"]))
        for extra in ("VGPR_Count", "LDS_Block_Size", "SGPR_Count"):
            if extra in r:
                out[k]["_" + extra] = float(r[extra])
res = {k: dict(v) for k, v in out.items()}
print(json.dumps(res, indent=1, sort_keys=True))
