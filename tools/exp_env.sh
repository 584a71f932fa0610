#!/bin/bash
# headline leg under probe-library environment settings: exp_env.sh "<bench args>" "VAR=val VAR2=val" ...   ("-" = no setting)
args=$1; shift
for e in "$@"; do
  echo "== $e"
  for i in 1 2; do
  ( [ "$e" != "-" ] && export $e; timeout -k 10 400 python bench.py --probe-run --no-legs --no-config4 --no-cpu $args 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], 'tr_ms', d['roofline']['ms_per_launch'], 'enc1', d['single_image_encode_ms'], 'dec1', d['single_image_decode_ms'])" )
  done
done
