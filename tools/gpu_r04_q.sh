#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04q; mkdir -p $O
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_base.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "full_size_oracle_parity and allegro" > $O/tests_base.txt 2>&1; tail -4 $O/tests_base.txt
grep "one-step\|knife" $O/tests_base.txt
