#!/bin/bash
# A/B of RELEASE library variants (xpng_amd/lib/variants/*.so copied over libxpng_hip.so): ab_release_variants.sh "<bench args>" variant.so ...
args=$1; shift
cp xpng_amd/lib/libxpng_hip.so /tmp/release_keep.so
for v in "$@"; do
  cp xpng_amd/lib/variants/$v xpng_amd/lib/libxpng_hip.so
  echo "== $v"
  for i in 1 2; do
  timeout -k 10 400 python bench.py --no-legs --no-config4 --no-cpu $args 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'GB', d['config']['hbm_in_use_gb'], 'enc1', d['single_image_encode_ms'], 'dec1', d['single_image_decode_ms'], d['verified'])"
  done
done
cp /tmp/release_keep.so xpng_amd/lib/libxpng_hip.so
