# usage (GPU box): bash tools/durs_small.sh $PWD -- per-launch durations (us, launch order) of the small dependent kernels of an iteration
ROOTDIR=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -- python $ROOTDIR/bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-extras --no-pmc > /tmp/pd.log 2>&1
db=$(find /tmp/pd -name "*_results.db" | head -1)
cd $ROOTDIR; for k in repair_kernel render_fwd_repair blend_order scan_chunk_offsets scan_chunk_sums emit_instances accumulate_scan accumulate_chain adam_kernel; do echo $k; python tools/kernel_durations.py $db $k | cut -c1-300; done
