#!/bin/bash
# round 5, call n: the Allegro closed loop on the build WITHOUT fast-math and contraction (is the drop rate a fast-math effect?), and
# more seeds on the product build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05n
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_ieee.so python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:64 --ticks 40 --json gpurun_out/r05n/allegro_ieee_N512.json > gpurun_out/r05n/allegro_ieee_N512.txt 2>&1
tail -1 gpurun_out/r05n/allegro_ieee_N512.txt
python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 64:256 --ticks 40 --json gpurun_out/r05n/allegro_gpu_N512_more.json > gpurun_out/r05n/allegro_gpu_N512_more.txt 2>&1
tail -1 gpurun_out/r05n/allegro_gpu_N512_more.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -s -k "closed_loop and allegro" > gpurun_out/r05n/test.log 2>&1; grep -E "passed|failed|stayed" gpurun_out/r05n/test.log | tail -3
