ROOTDIR=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -- python $ROOTDIR/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /tmp/pd.log 2>&1
db=$(find /tmp/pd -name "*_results.db" | head -1)
cd $ROOTDIR; python tools/kernel_durations.py $db render_bwd_kernel; python tools/kernel_durations.py $db render_fwd_kernel; python tools/kernel_durations.py $db accumulate_views
