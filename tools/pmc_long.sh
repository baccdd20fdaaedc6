#!/bin/bash
# PMC counters of the long-alignment latency-chain kernels on a handful of long superclusters (tools/long_latency.py)
R=$(pwd); O=$R/gpurun_out/pmc_long; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
export LEN=${LEN:-8000} NSC=${NSC:-4}
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/a --output-format csv -- python tools/long_latency.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH -d $O/b --output-format csv -- python tools/long_latency.py > $O/b.log 2>&1
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("O", "gpurun_out/pmc_long")
for sub in ("a", "b"):
    f = max(glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
    for k in acc:
        if any(x in k for x in ("stripe", "walk_rows", "k_credit")):
            print(sub, k, "dispatches", len(nd[k]), {c: round(v / len(nd[k])) for c, v in acc[k].items()})
PY
