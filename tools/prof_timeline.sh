#!/bin/bash
# kernel-trace timeline of one bench step (run on the GPU box through gpurun): tools/prof_timeline.sh <name> [bench args]
N=${1:-tl}; shift
R=$(pwd); O=$R/gpurun_out/prof_$N; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $O/stats.log 2>&1
python tools/timeline.py $O/stats > $O/timeline.txt 2>&1

grep '^{' $O/stats.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['kernel_ms_per_step'], 'value', d['value'])"
cat $O/timeline.txt
