#!/bin/bash
# headline leg (release library) under (environment, bench args) pairs: exp_cfg.sh "VAR=val|-" "<bench args>" ["VAR=val|-" "<bench args>" ...]
while [ $# -ge 2 ]; do
  e=$1; a=$2; shift 2
  echo "== $e :: $a"
  for i in 1 2; do
  ( [ "$e" != "-" ] && export $e; timeout -k 10 400 python bench.py --no-legs --no-config4 --no-cpu --steps 24 --warmup 4 --roofline-reps 10 $a 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'GB', d['config']['hbm_in_use_gb'])" )
  done
done
