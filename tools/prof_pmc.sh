#!/bin/bash
# PMC passes over one batched bench step (run on the GPU box through gpurun).  Each pass is its own rocprofv3 run
# with counters only (never combined with --sys-trace / hip / hsa tracing).  Output: gpurun_out/pmc_<tag>/
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}
run() { # tag counters...
  tag=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-include-regex "xpng" --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- \
    python3 $R/bench.py --no-cpu --no-legs --no-config4 --batch $B --pipeline 1 --steps 1 --warmup 1 --roofline-reps 1 > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "pass $tag done"
}
run insts SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run active SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
