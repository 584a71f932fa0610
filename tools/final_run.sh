#!/bin/bash
# end-of-round evidence run (GPU box): plain default bench (timed), rocprofv3 kernel stats + digest of the same command, whole-step HBM traffic
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
t0=$(date +%s)
timeout -k 10 600 python bench.py > gpurun_out/final_bench_plain.json 2> gpurun_out/final_bench_plain.err || { echo "bench failed"; tail -5 gpurun_out/final_bench_plain.err; exit 1; }
echo "default bench wall: $(( $(date +%s) - t0 )) s"
python3 -c "
import json
d=json.loads(open('gpurun_out/final_bench_plain.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'roof',d['roofline']['frac'],'GB',d['config']['hbm_in_use_gb'])
print('single',d['single_image'])
print('legs',{k:(v.get('value'),v.get('hbm_in_use_gb')) for k,v in d.get('legs',{}).items()})
print('config4',d.get('config4'))
print('cpu',d.get('cpu_baseline'))"
bash tools/prof_stats.sh final_prof --no-legs --no-config4 || exit 1
grep '^{' gpurun_out/final_prof.log | tail -1 > gpurun_out/final_prof_line.json
python3 tools/profile_digest.py gpurun_out/final_prof/p_kernel_trace.csv gpurun_out/final_prof_line.json > gpurun_out/final_prof_digest.json && echo digest ok
bash tools/prof_pmc_mem.sh 64 2> gpurun_out/final_pmc_step_mem.err | sed -n '/^{/,$p' > gpurun_out/final_pmc_step_mem.json || exit 1
python3 -c "
import json
d=json.load(open('gpurun_out/final_pmc_step_mem.json'))
print('step traffic B/px', d and d.get('bytes_per_px'))"
