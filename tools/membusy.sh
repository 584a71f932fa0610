#!/bin/bash
# sample the driver's busy percentages (gfx / memory controller) while the pipelined headline leg runs
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --no-legs --no-config4 --no-cpu --steps ${1:-500} --warmup 5 --roofline-reps 2 > gpurun_out/membusy_bench.json 2> gpurun_out/membusy_bench.err &
pid=$!
for i in $(seq 80); do
  if ! kill -0 $pid 2>/dev/null; then break; fi
  g=$(cat /sys/class/drm/card*/device/gpu_busy_percent 2>/dev/null | tr '\n' ' ')
  m=$(cat /sys/class/drm/card*/device/mem_busy_percent 2>/dev/null | tr '\n' ' ')
  echo "t=$i gpu_busy=[$g] mem_busy=[$m]"
  sleep 0.5
done
wait $pid
python3 -c "
import json
d=json.loads(open('gpurun_out/membusy_bench.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms', d['ms_per_step'])"
