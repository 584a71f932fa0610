#!/bin/bash
# headline leg only (default batch x slots), N runs; prints value / ms_per_step / single-image times.  usage: quick_bench.sh [N] [extra bench args]
n=${1:-2}; shift
for i in $(seq $n); do
  python bench.py --no-legs --no-config4 --no-cpu --steps 20 --warmup 5 --roofline-reps 20 "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], 'enc1', d['single_image_encode_ms'], 'dec1', d['single_image_decode_ms'], 'store', d['single_image'].get('store_ms'), 'load', d['single_image'].get('load_ms'), 'rawenc', d['single_image'].get('raw_entry_encode_ms'), 'step_frac', d.get('step_roofline', {}).get('frac'), 'GB', d['config']['hbm_in_use_gb'])"
done
