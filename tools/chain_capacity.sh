#!/bin/bash
# Throughput capacity of each chain kernel ALONE: every other kernel knocked out (XPNG_SKIP, tools/knockout.py --child), P slots.
# ms/step flat in P = the kernel's waves run into each other (a resource is full); ms/step ~ 1/P = they do not.
# usage: chain_capacity.sh "kernel names" "P values"
ALL="chooser,transform,streams,prep_a,chain_a,prep_c,chain_c,finish,gather,dec_prep,dec_chain_a,dec_alpha,dec_chain_c,dec_odd,walk_small,walk_big,resid_small,recon_small,resid_big,recon_big"
for k in ${1:-chain_a chain_c dec_chain_a dec_chain_c walk_small walk_big}; do
  skip=$(echo $ALL | tr ',' '\n' | grep -vx $k | paste -sd, -)
  for P in ${2:-1 4}; do
    r=$(XPNG_SKIP=$skip XPNG_SKIP_AFTER=$((4 * P)) timeout -k 10 120 python tools/knockout.py --child $P 64 both 2>/dev/null | tail -1)
    echo "$k P=$P $r ms/step"
  done
done
