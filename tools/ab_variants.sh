#!/bin/bash
# A/B of library variants built into xpng_amd/lib/variants/*.so (probe flavour): each is copied over libxpng_hip_probes.so and the
# headline leg is run with --probe-run.  usage: ab_variants.sh "<bench args>" variant.so ...
args=$1; shift
cp xpng_amd/lib/libxpng_hip_probes.so /tmp/probes_keep.so
for v in "$@"; do
  cp xpng_amd/lib/variants/$v xpng_amd/lib/libxpng_hip_probes.so
  echo "== $v"
  for i in 1 2; do
  timeout -k 10 400 python bench.py --probe-run --no-legs --no-config4 --no-cpu $args 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['frac'], 'tr_ms', d['roofline']['ms_per_launch'], 'enc1', d['single_image_encode_ms'], 'dec1', d['single_image_decode_ms'], 'store', d['single_image'].get('store_ms'), 'load', d['single_image'].get('load_ms'))"
  done
done
cp /tmp/probes_keep.so xpng_amd/lib/libxpng_hip_probes.so
