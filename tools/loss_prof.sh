cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lp; rocprofv3 --kernel-trace --stats -d /tmp/lp -- python $1/tools/loss_time.py fused > /tmp/lp.log 2>&1
db=$(find /tmp/lp -name "*_results.db" | head -1)
cd $1; python tools/rocpd_summary.py $db /tmp/lp.csv | head -14 | cut -c1-90
