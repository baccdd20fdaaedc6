#!/bin/bash
# Regenerates the rocprofv3 evidence under profiles/ (run on the GPU box from the repo root, e.g. through gpurun):
#   bash tools/make_profiles.sh r02 wgs_synth 1000000   -> gpurun_out/profiles_r02_wgs_synth/{r02_kernel_stats_*.csv, r02_counters_*.json}
# Counter passes use --kernel-trace + --pmc only (never combined with other trace domains).
set -u
TAG=${1:-r05}; WL=${2:-wgs_synth}; NSC=${3:-1000000}
R=$(pwd)
OUT=$R/gpurun_out/profiles_${TAG}_${WL}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$R"
CMD="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --workload $WL --n-sc $NSC"     # (warm-up >= 2: two batches in flight, as in the driver's command)
timeout ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- $CMD > "$OUT/stats.log" 2>&1
CMD2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 --one-pass-batches 0 --no-secondary --workload $WL --n-sc $NSC"
timeout ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- $CMD2 > "$OUT/fetch.log" 2>&1
timeout ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- $CMD2 > "$OUT/write.log" 2>&1
timeout ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d "$OUT/sq1" --output-format csv -- $CMD2 > "$OUT/sq1.log" 2>&1
timeout ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS \
    -d "$OUT/sq2" --output-format csv -- $CMD2 > "$OUT/sq2.log" 2>&1
python tools/collect_profiles.py "$OUT" "$TAG" "$WL" "$NSC"
grep '^{' "$OUT/stats.log" | tail -1 > "$OUT/${TAG}_bench_line_under_rocprof_${WL}.json"
# the timeline of ONE step alone (with batches in flight the kernels of three steps interleave)
timeout ${PASS_TIMEOUT:-400} rocprofv3 --kernel-trace -d "$OUT/tl" --output-format csv -- $CMD2 > "$OUT/tl.log" 2>&1
python tools/timeline.py "$OUT/tl" > "$OUT/${TAG}_timeline_${WL}.txt" 2>&1
ls -la "$OUT"/*.csv "$OUT"/*.json "$OUT"/*.txt
