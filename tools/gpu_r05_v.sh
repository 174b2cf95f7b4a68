#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05v
C=$PWD/dial_mpc_amd/csrc
python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:4 --ticks 40 --bitcheck > gpurun_out/r05v/bitcheck_product.txt 2>&1
DIAL_HIP_LIB=$C/libdialhip_ieee.so python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:4 --ticks 40 --bitcheck > gpurun_out/r05v/bitcheck_ieee.txt 2>&1
grep -h bitcheck gpurun_out/r05v/bitcheck_*.txt
DIAL_HIP_LIB=$C/libdialhip_ieee.so python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:192 --ticks 40 --plant-jitter 1 --json gpurun_out/r05v/ieee_plant_jitter1.json > gpurun_out/r05v/ieee_plant_jitter1.txt 2>&1
tail -1 gpurun_out/r05v/ieee_plant_jitter1.txt
