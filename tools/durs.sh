ROOTDIR=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -- python $ROOTDIR/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /tmp/pd.log 2>&1
db=$(find /tmp/pd -name "*_results.db" | head -1)
cd $ROOTDIR; python tools/kernel_durations.py $db radix_hist | cut -c1-400; python tools/kernel_durations.py $db radix_scatter | cut -c1-400; python tools/kernel_durations.py $db radix_rowscan | cut -c1-300
