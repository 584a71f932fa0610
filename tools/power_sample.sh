#!/bin/bash
# power, shader clock and busy figures of the driver while the pipelined leg runs (sampled twice a second)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_power_sample.txt
( python bench.py --no-cpu --no-legs --no-config4 --steps 200 --warmup 4 --roofline-reps 4 > $GRAFT_REPO_ROOT/gpurun_out/r4_power_bench.log 2>&1 ) &
BP=$!
: > $OUT
for i in $(seq 1 400); do
  kill -0 $BP 2>/dev/null || break
  line="t=$i"
  for H in /sys/class/drm/card*/device/hwmon/hwmon*; do
    D=$(dirname $(dirname $H))
    b=$(cat $D/gpu_busy_percent 2>/dev/null)
    [ "${b:-0}" -gt 0 ] && line="$line | $(basename $(dirname $D)) W=$(( $(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null || echo 0) / 1000000 )) cap=$(( $(cat $H/power1_cap 2>/dev/null || echo 0) / 1000000 )) sclk_MHz=$(( $(cat $H/freq1_input 2>/dev/null || echo 0) / 1000000 )) busy=$b mem_busy=$(cat $D/mem_busy_percent 2>/dev/null)"
  done
  echo "$line" >> $OUT
  sleep 0.5
done
wait $BP
rocm-smi --showpower --showclocks --showperflevel 2>&1 | head -40 >> $OUT
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r4_power_bench.log | cut -c1-200 >> $OUT
