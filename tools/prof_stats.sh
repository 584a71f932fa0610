#!/bin/bash
# rocprofv3 kernel trace of one batched bench run (on the GPU box through gpurun); output: gpurun_out/<tag>/
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-prof}; shift || true
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o p -- python3 $R/bench.py --no-cpu "$@" > $R/gpurun_out/$TAG.log 2>&1
grep '^{' $R/gpurun_out/$TAG.log | cut -c1-160
