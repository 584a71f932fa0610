#!/bin/bash
# tests + rocprofv3 kernel stats of the headline leg only: per-kernel averages of the library's kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/pq_tests.log 2>&1; rc=$?; tail -2 gpurun_out/pq_tests.log; [ $rc -eq 0 ] || exit 1
bash tools/prof_stats.sh pq_prof --no-legs --no-config4 "$@" || exit 1
python3 - <<'P'
import csv,re
for r in list(csv.DictReader(open('gpurun_out/pq_prof/p_kernel_stats.csv')))[:40]:
    n=r['Name']
    if 'at::' in n or 'rocclr' in n: continue
    n=re.sub(r'\(.*','',n).replace('void xpng::','').replace('xpng::','')
    print(f"{n:44s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms  tot {float(r['TotalDurationNs'])/1e6:9.1f}")
P
timeout -k 10 200 bash tools/quick_bench.sh 3 "$@"
