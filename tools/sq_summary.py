"""Aggregate rocprofv3 --pmc SQ_* counter collections (one or more passes) into one row per kernel.
usage: python tools/sq_summary.py out.csv pass1_counter_collection.csv [pass2_counter_collection.csv ...]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[(k, path)].add(r["Dispatch_Id"])
cols = sorted({c for v in agg.values() for c in v})
with open(sys.argv[1], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches"] + [c + "_per_wave" if c not in ("SQ_WAVES", "SQ_BUSY_CYCLES") else c + "_per_dispatch" for c in cols])
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0)):
        n = max(len(disp[(k, sys.argv[2])]), 1)
        waves = max(agg[k].get("SQ_WAVES", 0), 1)
        if agg[k].get("SQ_BUSY_CYCLES", 0) / n < 2e6:
            continue
        w.writerow([k, n] + [round(agg[k][c] / (n if c in ("SQ_WAVES", "SQ_BUSY_CYCLES") else waves), 1) for c in cols])
