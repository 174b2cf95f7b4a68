#!/bin/bash
# GPU call (round 4): second broad phases of the box narrow phases (separating face axis / slab / lowest vertex)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04bp; mkdir -p $O
timeout 60 python tools/ab_time.py tools/gpu_r04_bp_cases.txt 3 > $O/ab.txt 2>/dev/null; cat $O/ab.txt
timeout 200 python -m pytest tests/test_gpu_crate.py tests/test_gpu_push_crate.py -m gpu -q -x -k "not distribution" > $O/tests.txt 2>&1; tail -4 $O/tests.txt
