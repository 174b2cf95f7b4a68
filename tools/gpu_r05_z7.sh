#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z7
C=$PWD/dial_mpc_amd/csrc
python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:128 --ticks 40 --json gpurun_out/r05z7/drift_product.json > gpurun_out/r05z7/drift_product.txt 2>&1
DIAL_HIP_LIB=$C/libdialhip_ieee.so python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:128 --ticks 40 --json gpurun_out/r05z7/drift_strict.json > gpurun_out/r05z7/drift_strict.txt 2>&1
tail -1 gpurun_out/r05z7/drift_product.txt; tail -1 gpurun_out/r05z7/drift_strict.txt
