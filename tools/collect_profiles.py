"""Turn the raw rocprofv3 output of tools/make_profiles.sh into the compact files kept under profiles/."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

out, tag = sys.argv[1], sys.argv[2]


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    if not g:
        raise SystemExit("missing " + pattern)
    return g[0]


# bench.py's names for the kernels (vpr_launch_stat.kernel)
def bench_name(k):
    k = k.split("(")[0].replace("void ", "")
    return {"k_bwd_q16<true>": "k_bwd_q16<zero>", "k_bwd_q16<false>": "k_bwd_q16", "k_credit<false>": "k_credit<lane>",
            "k_credit<true>": "k_credit<wave>", "k_walk<false>": "k_walk<lane>", "k_walk<true>": "k_walk<wave>"}.get(k, k)


# 1. kernel stats
src = one("stats/**/*kernel_stats.csv")
rows = list(csv.reader(open(src)))
with open(os.path.join(out, f"{tag}_kernel_stats_wgs1M.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:120]
        w.writerow(r)

# 2. HBM traffic
def kb(kind):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(one(f"{kind}/**/*counter_collection.csv"))):
        a = acc[bench_name(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc
fe, wr = kb("fetch"), kb("write")
kern = {}
with open(os.path.join(out, f"{tag}_counters_wgs1M.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "FETCH_SIZE_KB_per_call", "WRITE_SIZE_KB_per_call"])
    for k in sorted(fe, key=lambda k: -(fe[k][1] + wr[k][1])):
        n = fe[k][0]
        if (fe[k][1] + wr[k][1]) / max(n, 1) < 1024:
            continue
        kern[k] = {"calls": n, "fetch_kb": round(fe[k][1] / n), "write_kb": round(wr[k][1] / max(wr[k][0], 1))}
        w.writerow([k, n, kern[k]["fetch_kb"], kern[k]["write_kb"]])
json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "workload": "wgs_synth", "superclusters_per_gpu": 1000000,
           "note": "KB per kernel launch as reported (launches of one kernel with different lists are averaged); per MI355X_MICROARCH.md "
                   "FETCH_SIZE on gfx950 counts half the bytes of wide (16 B/lane) coalesced reads",
           "kernels": kern}, open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1)

# 3. SQ counters
subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), "sq_summary.py"),
                       os.path.join(out, f"{tag}_sq_counters_wgs1M.csv"),
                       one("sq1/**/*counter_collection.csv"), one("sq2/**/*counter_collection.csv")])
