#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05u
C=$PWD/dial_mpc_amd/csrc
# hybrid A: product planner, strict-IEEE plant
python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:128 --ticks 40 --plant-lib $C/libdialhip_ieee.so --json gpurun_out/r05u/planner_product_plant_ieee.json > gpurun_out/r05u/planner_product_plant_ieee.txt 2>&1
tail -1 gpurun_out/r05u/planner_product_plant_ieee.txt
# hybrid B: strict-IEEE planner, product plant
DIAL_HIP_LIB=$C/libdialhip_ieee.so python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:128 --ticks 40 --plant-lib $C/libdialhip.so --json gpurun_out/r05u/planner_ieee_plant_product.json > gpurun_out/r05u/planner_ieee_plant_product.txt 2>&1
tail -1 gpurun_out/r05u/planner_ieee_plant_product.txt
