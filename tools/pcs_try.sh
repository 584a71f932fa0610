#!/bin/bash
# PC sampling (rocprofv3 beta) of the pipelined step: does it run on this box, and what do the samples look like?
OUT=$GRAFT_REPO_ROOT/gpurun_out/pcs_try; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export KNOCKOUT_STEPS=${KSTEPS:-4}
METHOD=${METHOD:-stochastic}; UNIT=${UNIT:-cycles}; INTERVAL=${INTERVAL:-1048576}
timeout -k 10 ${LIMIT:-300} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL \
  --kernel-trace --output-format csv -d /tmp/pcs_raw -- python3 $GRAFT_REPO_ROOT/tools/knockout.py --child ${P:-2} ${B:-8} both > $OUT/run.log 2>&1
echo "rc $?" >> $OUT/run.log
find /tmp/pcs_raw -type f | while read f; do echo "$f $(stat -c %s $f)"; done > $OUT/files.txt
for f in $(find /tmp/pcs_raw -type f -name '*.csv'); do head -c 6000 $f > $OUT/head_$(basename $f); done
exit 0
