#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args]
# kernel-trace + stats of the bench command; prints the per-kernel summary and leaves
# gpurun_out/<tag>_kernel_stats.csv (copy into profiles/ to keep it)
tag=$1; shift
ROOTDIR=$(pwd)
mkdir -p $ROOTDIR/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $ROOTDIR/bench.py --steps ${PROF_STEPS:-20} --warmup ${PROF_WARMUP:-5} --no-cpu-baseline "$@" > $ROOTDIR/gpurun_out/${tag}_bench.log 2>&1
db=$(find /tmp/prof_$tag -name "*_results.db" | head -1)
cd $ROOTDIR
python tools/rocpd_summary.py $db gpurun_out/${tag}_kernel_stats.csv | head -${PROF_LINES:-24}
tail -1 gpurun_out/${tag}_bench.log | cut -c1-400
