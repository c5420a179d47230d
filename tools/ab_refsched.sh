#!/bin/bash
# usage (GPU box, repo root): tools/ab_refsched.sh [sizes...] -- the two-view schedule (fused_graph surface: device-bound, immune
# to host jitter) once per tools/ab/*.so and twice with the in-tree library; extra env through AB_ENV="K=V ..."
sizes=${@:-"500000,800,600 100000,504,378"}
run() { env B3GS_LIB=$1 $AB_ENV python bench_ref_schedule.py $sizes --surfaces=fused_graph 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', '$AB_ENV', {k:v['fused_graph']['iters_per_s'] for k,v in d.items() if k!='what'})"; }
run ""
for f in tools/ab/*.so; do run $PWD/$f; done
run ""
