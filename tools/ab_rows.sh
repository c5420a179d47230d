for i in 1 2; do
for f in "" "--dense-grad-rows"; do
python bench.py --no-extras --no-pmc --no-cpu-baseline --steps 40 --warmup 10 $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$f', d['value'], d['ms_per_step'], {k:round(v,4) for k,v in d.get('kernel_ms',{}).items()} if 'kernel_ms' in d else '')"
done; done
