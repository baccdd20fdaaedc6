"""Turn the raw rocprofv3 output of tools/make_profiles.sh into the compact files kept under profiles/.
usage: python tools/collect_profiles.py <raw dir> <tag> <workload> <superclusters per GPU>

A kernel runs in several roles per step (round 0 over a whole part; retry and tie rounds over a few alignments).  What is
kept per kernel is its MAIN launch: the dispatches whose grid is at least half of the kernel's largest grid."""
import collections
import csv
import glob
import json
import os
import sys

out, tag, workload, n_sc = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])


def one(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    if not g:
        raise SystemExit("missing " + pattern)
    return max(g, key=os.path.getmtime)      # (a directory may hold the files of an earlier, failed run)


def bench_name(k):      # bench.py's names for the kernels (vpr_launch_stat.kernel)
    k = k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace(", ", ",")
    if k.startswith("k_bwd<") and k.endswith(",false>"):      # dense classes: bench.py says k_bwd<NT,C> / k_bwd<NT,C,s16>
        k = k[:-len(",false>")] + ">"
    elif k.startswith("k_bwd<") and k.endswith(",true>") and k.count(",") == 2:
        k = k[:-len(",true>")] + ",s16>"
    return {"k_bwd_q16<true>": "k_bwd_q16<zero>", "k_bwd_q16<false>": "k_bwd_q16", "k_credit<false>": "k_credit<lane>",
            "k_credit<true>": "k_credit<wave>", "k_walk<false>": "k_walk<lane>", "k_walk<true>": "k_walk<wave>"}.get(k, k)


# 1. kernel stats of the --stats pass: calls / total / average per kernel name, main launches separately (from the trace)
trace = list(csv.DictReader(open(one("stats/**/*kernel_trace.csv"))))
grid = collections.defaultdict(int)
for r in trace:
    grid[bench_name(r["Kernel_Name"])] = max(grid[bench_name(r["Kernel_Name"])], int(r.get("Grid_Size") or r["Grid_Size_X"]))
acc = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for r in trace:
    k = bench_name(r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = acc[k]
    if int(r.get("Grid_Size") or r["Grid_Size_X"]) * 2 >= grid[k]:
        a[0] += 1; a[1] += d
    else:
        a[2] += 1; a[3] += d
with open(os.path.join(out, f"{tag}_kernel_stats_{workload}.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "main_launches", "main_total_ms", "main_avg_ms", "other_launches", "other_total_ms"])
    for k in sorted(acc, key=lambda k: -(acc[k][1] + acc[k][3])):
        a = acc[k]
        if a[1] + a[3] < 0.05:
            continue
        w.writerow([k, a[0], round(a[1], 3), round(a[1] / max(a[0], 1), 4), a[2], round(a[3], 3)])


# 2. counters: per kernel, main launches only
def counters(sub):
    path = one(f"{sub}/**/*counter_collection.csv")
    rows = list(csv.DictReader(open(path)))
    g = collections.defaultdict(int)
    for r in rows:
        g[bench_name(r["Kernel_Name"])] = max(g[bench_name(r["Kernel_Name"])], int(r.get("Grid_Size") or r["Grid_Size_X"]))
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in rows:
        k = bench_name(r["Kernel_Name"])
        if int(r["Grid_Size"]) * 2 < g[k]:
            continue
        vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return vals, disp


kern = collections.defaultdict(dict)
for sub in ("fetch", "write", "sq1", "sq2"):
    vals, disp = counters(sub)
    for k, cv in vals.items():
        n = max(len(disp[k]), 1)
        for c, v in cv.items():
            kern[k][c] = v / n            # per main launch
kept = {}
for k, c in kern.items():
    if c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0) < 1024 and c.get("SQ_BUSY_CYCLES", 0) < 2e6:
        continue
    waves = c.get("SQ_WAVES", 0)
    e = {"fetch_kb": round(c.get("FETCH_SIZE", 0)), "write_kb": round(c.get("WRITE_SIZE", 0)), "waves": round(waves),
         "busy_cycles_8xcd": round(c.get("SQ_BUSY_CYCLES", 0))}
    if waves:
        for name, key in (("SQ_WAVE_CYCLES", "wave_cycles_per_wave"), ("SQ_INSTS_VALU", "valu_insts_per_wave"),
                          ("SQ_INSTS_SALU", "salu_insts_per_wave"), ("SQ_INSTS_LDS", "lds_insts_per_wave"),
                          ("SQ_INSTS_VMEM_RD", "vmem_rd_per_wave"), ("SQ_INSTS_VMEM_WR", "vmem_wr_per_wave"),
                          ("SQ_ACTIVE_INST_VALU", "valu_active_per_wave"), ("SQ_ACTIVE_INST_ANY", "any_active_per_wave"),
                          ("SQ_WAIT_INST_ANY", "wait_inst_per_wave"), ("SQ_WAIT_ANY", "wait_any_per_wave"),
                          ("SQ_ACTIVE_INST_LDS", "lds_active_per_wave"), ("SQ_WAIT_INST_LDS", "lds_wait_per_wave"),
                          ("SQ_ACTIVE_INST_SCA", "salu_active_per_wave")):
            if name in c:
                e[key] = round(c[name] / waves, 1)
    kept[k] = e
json.dump({"command": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                      f"--workload {workload} --n-sc {n_sc}",
           "passes": ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR",
                      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"],
           "workload": workload, "superclusters_per_gpu": n_sc,
           "note": "values per MAIN launch of each kernel (dispatches with at least half of the kernel's largest grid); fetch_kb / write_kb as "
                   "reported (per MI355X_MICROARCH.md FETCH_SIZE on gfx950 counts half the bytes of wide coalesced reads: bench.py doubles it); "
                   "*_per_wave = counter / SQ_WAVES; the *_active / wait counters are in quad-cycles; busy_cycles_8xcd is summed over the 8 XCDs",
           "kernels": dict(sorted(kept.items(), key=lambda kv: -kv[1].get("busy_cycles_8xcd", 0)))},
          open(os.path.join(out, f"{tag}_counters_{workload}.json"), "w"), indent=1)
print(open(os.path.join(out, f"{tag}_kernel_stats_{workload}.csv")).read())
