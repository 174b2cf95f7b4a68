#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05q
for seed in 0 1 2 3 4 5 6 7; do
  python tools/allegro_closed_loop_parity.py --seed $seed --nsample 512 --ticks 40 --rollouts 32 --dump gpurun_out/r05q/bad > gpurun_out/r05q/parity_seed$seed.txt 2>&1
  tail -1 gpurun_out/r05q/parity_seed$seed.txt
done
grep -h "UNWITNESSED [1-9]\|saved" gpurun_out/r05q/parity_seed*.txt | head -30
