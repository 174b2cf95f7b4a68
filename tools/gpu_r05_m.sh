#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05m
for seed in 0 11 31 36 1 2; do
  python tools/allegro_closed_loop_parity.py --seed $seed --nsample 512 --ticks 40 --rollouts 24 > gpurun_out/r05m/parity_seed$seed.txt 2>&1
  tail -1 gpurun_out/r05m/parity_seed$seed.txt
done
grep -h "UNWITNESSED [1-9]" gpurun_out/r05m/parity_seed*.txt | head -20
