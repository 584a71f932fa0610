#!/bin/bash
# chains alone / chip-filling kernels alone / both, against the number of slots (tools/knockout.py groups)
for P in ${SLOTS:-4 6 8 10}; do
  echo "== slots $P"
  KNOCKOUT_STEPS=$((P*3)) timeout -k 10 600 python tools/knockout.py $P 64 both none,all_bw,all_chains || exit 1
done
echo "== chains alone at 8 slots, one chain class less"
KNOCKOUT_STEPS=24 timeout -k 10 600 python tools/knockout.py 8 64 both all_bw,all_bw+chain_a,all_bw+chain_c,all_bw+dec_chain_a,all_bw+dec_chain_c,all_bw+walk_small,all_bw+walk_big
