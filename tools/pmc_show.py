import csv, glob, collections, os, sys
for tag in sys.argv[1:]:
    for d in "abc":
        fs = sorted(glob.glob(f"gpurun_out/pmc_mix_{tag}/{d}/vm/*_counter_collection.csv"), key=os.path.getmtime)
        if not fs: print(tag, d, "no output"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(fs[-1])):
            k = r["Kernel_Name"].split("(")[0][:48]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            if "coop_kernel<0" in k: print(tag, k, len(n[k]), {c: f"{x:.4g}" for c, x in v.items()})
