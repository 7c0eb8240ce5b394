"""Per-launch durations of the render kernels from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): the launches of the
LAST step, in order. Usage: python tools/launch_times.py <dir>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = [i for i, r in enumerate(rows) if "wf_accumulate" in r["Kernel_Name"]]
steps = [(acc[j - 1] + 1 if j else 0, acc[j]) for j in range(len(acc))]
# the last step whose segment kernel is not the counting variant (template argument FLAGS = 0)
steps = [st for st in steps if any("wf_segment" in r["Kernel_Name"] and "<0," in r["Kernel_Name"].replace(" ", "") for r in rows[st[0]:st[1]])]
lo, hi = steps[-1]
out = []
for r in rows[lo:hi + 1]:
    if "rocclr" in r["Kernel_Name"]:
        continue
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("wf_", "")[:14]
    out.append("%s %.2f" % (n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
print(" | ".join(out))
