#!/bin/bash
# One rocprofv3 kernel trace of the default bench step; leaves gpurun_out/<tag>/{timeline.txt,c_kernel_stats.csv} and the
# captured graph's DOT + simulated executor streams.  usage: profile_step.sh <tag> [bench args...]
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python3 $R/bench.py --graph-dot $OUT/step.dot --steps 3 --warmup 2 --no-cpu-baseline --no-other "$@" > /dev/null 2>&1 || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o c --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other "$@" > $OUT/bench.log 2>&1
cd $R
python3 scripts/step_timeline.py $OUT/c_kernel_trace.csv > $OUT/timeline.txt
[ -f $OUT/step.dot ] && python3 scripts/graph_streams.py $OUT/step.dot > $OUT/streams.txt || true
rm -f $OUT/c_kernel_trace.csv
grep -o '"ms_per_step": [0-9.]*' $OUT/bench.log | head -1
head -1 $OUT/timeline.txt
