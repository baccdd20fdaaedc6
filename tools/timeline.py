"""Print the kernel timeline of the last full step of a `rocprofv3 --kernel-trace` run of bench.py.
usage: python tools/timeline.py <dir with *_kernel_trace.csv>   (MIN_NS: shortest kernel shown, default 300e3)"""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id")) for r in rows)
big = [k for k in ks if k[2] == "k_fwd_z16" and k[1] - k[0] > 3e6]     # the round-0 sweep (tie rounds re-run it on a few alignments)
start = big[-1][0] - 3e6
for k in ks:
    if k[0] >= start and k[1] - k[0] > float(os.environ.get("MIN_NS", "300e3")):
        print(f"{(k[0] - start) / 1e6:7.2f} -> {(k[1] - start) / 1e6:7.2f}  q{k[3]}  {k[2] or 'copy/fill'}")
