"""Print the kernel timeline of the last full step of a `rocprofv3 --kernel-trace` run of bench.py.
usage: python tools/timeline.py <dir with *_kernel_trace.csv>   (MIN_NS: shortest kernel shown, default 300e3)"""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""), r.get("Queue_Id")) for r in rows)
start = [k for k in ks if k[2] == "k_init_execute"][-1][0]              # every vpr_execute begins with this launch
for k in ks:
    if k[0] >= start and k[1] - k[0] > float(os.environ.get("MIN_NS", "300e3")):
        print(f"{(k[0] - start) / 1e6:7.2f} -> {(k[1] - start) / 1e6:7.2f}  q{k[3]}  {k[2] or 'copy/fill'}")
