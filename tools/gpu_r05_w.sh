#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05w
C=$PWD/dial_mpc_amd/csrc
DIAL_HIP_LIB=$C/libdialhip_ieee.so python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:192 --ticks 40 --plant-jitter 1 --jitter-ticks 1 --json gpurun_out/r05w/ieee_oneoff_jitter.json > gpurun_out/r05w/ieee_oneoff_jitter.txt 2>&1
tail -1 gpurun_out/r05w/ieee_oneoff_jitter.txt
