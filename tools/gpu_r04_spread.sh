#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04s; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "spread_launch" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 100 python tools/ab_time.py tools/gpu_r04_spread_cases.txt 2 > $O/ab.txt 2>/dev/null; cat $O/ab.txt
