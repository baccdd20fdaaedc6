"""Print the kernel timeline of the last step of a `rocprofv3 --kernel-trace` run of bench.py.
usage: python tools/timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id")) for r in rows)
zs = [i for i, k in enumerate(ks) if k[2] == "k_fwd_z16"]
t0 = ks[zs[-1]][0]
start = min(k[0] for k in ks if t0 - 3e6 < k[0] <= t0)
for k in (k for k in ks if k[0] >= start):
    if k[1] - k[0] > float(os.environ.get("MIN_NS", "400e3")) and k[2]:
        print(f"{(k[0] - start) / 1e6:7.2f} -> {(k[1] - start) / 1e6:7.2f}  q{k[3]}  {k[2]}")
