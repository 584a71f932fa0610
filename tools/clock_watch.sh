#!/bin/bash
# sample the GPU's shader / memory clocks and power while bench.py runs (is the chip throttling under the pipelined load?)
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/bench.py --no-cpu --steps 120 > $R/gpurun_out/clk_bench.log 2>&1 &
BP=$!
: > $R/gpurun_out/clk_samples.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' ' >> $R/gpurun_out/clk_samples.txt
  echo >> $R/gpurun_out/clk_samples.txt
  sleep 0.5
done
wait $BP
grep -o '"value": [0-9.]*' $R/gpurun_out/clk_bench.log
