#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
python tools/pair_diff.py unitree_go2_trot 2048 16 > gpurun_out/r05b/diff_default.txt 2>&1
python tools/pair_diff.py unitree_go2_trot 2048 16 swap > gpurun_out/r05b/diff_default_swap.txt 2>&1
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/ab_nocontract.so python tools/pair_diff.py unitree_go2_trot 2048 16 > gpurun_out/r05b/diff_nocontract.txt 2>&1
tail -20 gpurun_out/r05b/*.txt
