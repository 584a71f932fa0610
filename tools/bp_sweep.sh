#!/bin/bash
# usage: bp_sweep.sh OUT "B P" "B P" ...   headline-only bench at the given batch / pipeline-slot pairs
out=$1; shift
: > $out
for cfg in "$@"; do
  set -- $cfg
  v=$(timeout -k 10 300 python bench.py --no-legs --no-config4 --no-cpu --batch $1 --pipeline $2 --steps 16 --warmup 3 --roofline-reps 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('hbm_in_use_gb'))")
  echo "B=$1 P=$2 $v" | tee -a $out
done
