#!/bin/bash
# usage (GPU box, repo root): tools/final_prof.sh <tag> -- the evidence set kept under profiles/ for one version:
# kernel stats + JSON line of the profiled driver command, the full default JSON line, three PMC passes
tag=$1
tools/prof.sh ${tag}_default --no-extras --no-pmc > gpurun_out/${tag}_prof_stdout.txt 2>&1
grep '^{"metric"' gpurun_out/${tag}_default_bench.log | tail -1 > gpurun_out/${tag}_profiled_run_bench_line.json
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' | tail -1 > gpurun_out/${tag}_default_bench_line.json
# (same warm-up as the driver's command: settled open-tile prediction; these CSVs average ALL launches of the pass, warm-up
# included -- bench.py's own passes keep the launches of the timed steps only: `hbm_measured` / `roofline.traffic` in the JSON)
export PMC_TARGET="bench.py --inner --no-extras --no-pmc --no-cpu-baseline --steps 3 --warmup 5"
tools/pmc.sh ${tag}_pmc_fetch_size "FETCH_SIZE" "render|preprocess|radix|emit|scan|accumulate|adam|repair" > gpurun_out/${tag}_pmc_f.txt 2>&1
tools/pmc.sh ${tag}_pmc_write_size "WRITE_SIZE" "render|preprocess|radix|emit|scan|accumulate|adam|repair" > gpurun_out/${tag}_pmc_w.txt 2>&1
tools/pmc.sh ${tag}_pmc_valu "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS" "render|preprocess|radix|emit|scan|accumulate|adam|repair" > gpurun_out/${tag}_pmc_v.txt 2>&1
# round 3: static instruction mix of the blend loops, per-wave trace of the batched backward (candidates, live lanes,
# residency), kernel stats of the reference-shaped surface (render() per view)
python tools/isa_count.py > gpurun_out/${tag}_isa_counts.txt 2>&1
python tools/bwd_trace_batched.py bwd 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${tag}_bwd_trace.txt
python tools/bwd_trace_batched.py fwd 2>/dev/null | grep -v amdgpu.ids > gpurun_out/${tag}_fwd_trace.txt
PROF_STEPS=10 PROF_WARMUP=3 tools/prof.sh ${tag}_dropin --path dropin --optimizer b3gs --graph 0 --no-extras --no-pmc > gpurun_out/${tag}_dropin_stdout.txt 2>&1
# round 4: the reference's own two-view iteration (fused surface), 500k Gaussians
tools/prof_cmd.sh ${tag}_refsched_fused_500k bench_ref_schedule.py 500000,800,600 --surfaces=fused > gpurun_out/${tag}_refsched_stdout.txt 2>&1
head -16 gpurun_out/${tag}_default_kernel_stats.csv
python -c "
import json
d=json.load(open('gpurun_out/${tag}_default_bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('valu',{}).get('frac'), d.get('hbm_measured',{}).get('frac_of_peak'), {k:(v.get('iters_per_s') if isinstance(v,dict) else v) for k,v in d['extras'].items() if k!='dropin_what'})
p=json.load(open('gpurun_out/${tag}_profiled_run_bench_line.json')); print('profiled run avg_launch_ms', p['roofline']['avg_launch_ms'], p['value'])"
cat gpurun_out/${tag}_pmc_v.txt | tail -12
# round 5: the reference's own loop with swapped imports only (bench_ref_schedule.py "unchanged"): device timeline + host profile;
# the bin-first estimate
python tools/unchanged_profile.py 500000 800 600 unchanged 2>&1 | grep -v -i warn > gpurun_out/${tag}_unchanged_500k_800x600_device_profile.txt
python tools/unchanged_host_profile.py unchanged 2>&1 | grep -v -i warn | head -70 > gpurun_out/${tag}_unchanged_host_profile.txt
python tools/binfirst_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_binfirst_probe.txt
tools/prof_cmd.sh ${tag}_refsched_unchanged_500k bench_ref_schedule.py 500000,800,600 --surfaces=unchanged > gpurun_out/${tag}_refsched_unchanged_stdout.txt 2>&1
# round 6: the surface north_star names literally (device timeline), and the reference loop's call sequence under the three forward
# policies, interleaved in one process (the protocol the verdict's bars are stated in)
python tools/module_surface_profile.py 2>&1 | grep -v -i warn > gpurun_out/${tag}_module_surface_profile.txt
(python tools/ab_interleaved.py lazy 500000 504 378; python tools/ab_interleaved.py lazy 500000 800 600) 2>&1 | grep -v -i warn > gpurun_out/${tag}_ab_interleaved.txt
