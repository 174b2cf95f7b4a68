#!/bin/bash
# round 5, call r: which part of the product build's arithmetic makes the Allegro loop lose the ball?  256 seeds per variant
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05r
for v in ieee_contract fast_nocontract; do
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/ab_$v.so python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:256 --ticks 40 --json gpurun_out/r05r/allegro_$v.json > gpurun_out/r05r/allegro_$v.txt 2>&1
  echo "$v: $(tail -1 gpurun_out/r05r/allegro_$v.txt)"
done
