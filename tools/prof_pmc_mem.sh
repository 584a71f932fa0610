#!/bin/bash
# HBM traffic of EVERY kernel of one batched encode+decode step: rocprofv3 counters-only passes (FETCH_SIZE and WRITE_SIZE cannot
# share a pass) over bench.py with one pipeline slot and one timed step.  Output: gpurun_out/pmcmem_<counter>/, digest on stdout.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}; shift || true
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-include-regex "xpng" --pmc $c --output-format csv -d $R/gpurun_out/pmcmem_$c -o pmc -- \
    python3 $R/bench.py --no-cpu --no-legs --no-config4 --batch $B --pipeline 1 --steps 1 --warmup 1 --roofline-reps 1 "$@" > $R/gpurun_out/pmcmem_$c.log 2>&1
  echo "pass $c done"
done
python3 $R/tools/pmc_mem_digest.py $R/gpurun_out/pmcmem_FETCH_SIZE $R/gpurun_out/pmcmem_WRITE_SIZE $B "$@"
