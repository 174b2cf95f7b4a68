#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05o
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_ieee.so python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 64:256 --ticks 40 --json gpurun_out/r05o/allegro_ieee_N512_more.json > gpurun_out/r05o/allegro_ieee_N512_more.txt 2>&1
tail -1 gpurun_out/r05o/allegro_ieee_N512_more.txt
