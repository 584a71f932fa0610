#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for st in "$@"; do
  echo "== stagger $st ms (200 timed steps)"
  for i in 1 2; do
  python bench.py --no-legs --no-config4 --no-cpu --steps 200 --warmup 4 --roofline-reps 5 --stagger-ms $st 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'])"
  done
done
