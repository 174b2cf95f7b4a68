#!/bin/bash
# GPU call L (round 4): H1 / H1 loco on the row layout (smooth_rows.h) -- parity tests, A/B against the phase version on one box
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04l; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "h1" > $O/tests_h1.txt 2>&1; tail -5 $O/tests_h1.txt
tools/ab_bench.sh dial_mpc_amd/csrc/libdialhip_base.so dial_mpc_amd/csrc/libdialhip.so unitree_h1_jog unitree_h1_loco unitree_go2_trot > $O/ab_rows.txt 2>&1
cat $O/ab_rows.txt
