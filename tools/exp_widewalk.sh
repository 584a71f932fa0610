#!/bin/bash
# wide walk for the big tiles too, with more slots
cd $GRAFT_REPO_ROOT

for cfg in "4 0" "5 0" "6 0" "4 1" "5 1" "6 1" "7 1"; do
  set -- $cfg
  echo "== pipeline $1 wide_big_walk $2"
  if [ "$2" = "1" ]; then export XPNG_WIDE_BIG_WALK=1; else unset XPNG_WIDE_BIG_WALK; fi
  timeout -k 10 200 bash tools/quick_bench.sh 2 --probe-run --pipeline $1 || exit 1
done
