#!/bin/bash
# headline bench (value, ms/step, roofline frac of the transform pair) under ROC_GLOBAL_CU_MASK patterns
for pat in "$@"; do
  M=0x$(python3 -c "print('$pat'*(64//len('$pat')))")
  v=$(ROC_GLOBAL_CU_MASK=$M timeout -k 10 300 python bench.py --no-legs --no-config4 --no-cpu --steps 12 --warmup 3 --roofline-reps 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['single_image_encode_ms'], d['single_image_decode_ms'])")
  echo "mask $pat: $v"
done
