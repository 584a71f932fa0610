#!/bin/bash
# shader clock / power while ONE chain kernel runs alone at P slots (tools/chain_capacity.sh's setting, many steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
ALL="chooser,transform,streams,prep_a,chain_a,prep_c,chain_c,finish,gather,dec_prep,dec_chain_a,dec_alpha,dec_chain_c,dec_odd,walk_small,walk_big,resid_small,recon_small,resid_big,recon_big"
k=${1:-dec_chain_a}
skip=$(echo $ALL | tr ',' '\n' | grep -vx $k | paste -sd, -)
for P in ${2:-1 6}; do
  XPNG_SKIP=$skip XPNG_SKIP_AFTER=$((4 * P)) KNOCKOUT_STEPS=$((60 * P)) python $R/tools/knockout.py --child $P 64 both > $R/gpurun_out/clk_chain_$P.log 2>&1 &
  BP=$!
  : > $R/gpurun_out/clk_chain_$P.txt
  while kill -0 $BP 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $R/gpurun_out/clk_chain_$P.txt
    echo >> $R/gpurun_out/clk_chain_$P.txt
    sleep 0.3
  done
  wait $BP
  echo "P=$P ms/step $(tail -1 $R/gpurun_out/clk_chain_$P.log)"; tail -6 $R/gpurun_out/clk_chain_$P.txt | cut -c1-200
done
