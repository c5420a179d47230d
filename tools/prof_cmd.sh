#!/bin/bash
# usage (GPU box, repo root): tools/prof_cmd.sh <tag> <python script + args...> -- rocprofv3 --kernel-trace --stats of any
# command of this repo; prints the per-kernel summary, leaves gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
ROOTDIR=$(pwd)
mkdir -p $ROOTDIR/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
( cd $ROOTDIR && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python "$@" > $ROOTDIR/gpurun_out/${tag}.log 2>&1 )
db=$(find /tmp/prof_$tag -name "*_results.db" | head -1)
cd $ROOTDIR
python tools/rocpd_summary.py $db gpurun_out/${tag}_kernel_stats.csv | head -${PROF_LINES:-40}
