#!/bin/bash
# PMC passes over the roofline kernels only (k_chooser + k_m1_transform), B rasters per launch.  Counters only per pass.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-64}; MODE=${2:-rgba}
run() { tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-include-regex "xpng" --pmc "$@" --output-format csv -d $R/gpurun_out/pmct_${MODE}_$tag -o pmc -- python3 $R/tools/gpu_transform_only.py 4096 $B 3 $MODE > $R/gpurun_out/pmct_${MODE}_$tag.log 2>&1
  echo "pass $tag done"; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT
run d FETCH_SIZE
run e WRITE_SIZE
