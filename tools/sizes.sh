for P in 100000 500000 1000000 2000000; do
python bench.py --gaussians $P --no-extras --no-pmc --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); c=d['config']; print('P=$P 800x600 6 views:', d['value'], 'iters/s', d['ms_per_step'], 'ms', 'N_binned/view', c.get('instances_N_binned'), 'rounds', c.get('binning_rounds'), 'frac', d['roofline']['frac'])"
done
