#!/bin/bash
# RGB legs at several (batch, slots): value, ms per step, GB in use
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  set -- $cfg
  echo "== level $1 --batch $2 --pipeline $3"
  timeout -k 10 250 python bench.py --rgb --level $1 --batch $2 --pipeline $3 --steps 8 --warmup 2 --roofline-reps 5 --no-legs --no-config4 --no-cpu 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['hbm_in_use_gb'], 'enc1', d['single_image_encode_ms'], 'dec1', d['single_image_decode_ms'])" || exit 1
done
