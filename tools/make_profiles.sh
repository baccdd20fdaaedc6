#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ (run on the GPU box from the repo root, e.g. through gpurun):
#   bash tools/make_profiles.sh r01        -> gpurun_out/profiles_r01/{kernel_stats,counters,hbm_traffic,sq_counters}
# Counter passes use --kernel-trace + --pmc only (never combined with other trace domains).
set -u
TAG=${1:-r01}
R=$(pwd)
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$R"
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- $CMD > "$OUT/stats.log" 2>&1
CMD2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $CMD2 > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $CMD2 > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d "$OUT/sq1" --output-format csv -- $CMD2 > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS \
    -d "$OUT/sq2" --output-format csv -- $CMD2 > "$OUT/sq2.log" 2>&1
python tools/collect_profiles.py "$OUT" "$TAG"
ls -la "$OUT"/*.csv "$OUT"/*.json
