#!/bin/bash
# usage: pad_sweep.sh OUT "ENV1=.. ENV2=.." ...   (each arg one configuration; runs bench headline only at P=3 and P=5)
out=$1; shift
: > $out
for cfg in "$@"; do
  for P in 3 5; do
    v=$(env $cfg timeout -k 10 200 python bench.py --probe-run --no-legs --no-config4 --no-cpu --pipeline $P --steps 20 --warmup 3 --roofline-reps 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "P=$P [$cfg] $v" | tee -a $out
  done
done
