#!/bin/bash
# GPU call Y (round 4): factor re-use in the generic Newton solver (crate scenes)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04y; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_crate.py tests/test_gpu_push_crate.py -m gpu -q -x -k "not closed_loop and not distribution" > $O/tests.txt 2>&1; tail -8 $O/tests.txt
timeout 240 python tools/ab_time.py tools/gpu_r04_y_cases.txt 3 > $O/ab.txt 2> $O/ab.err; cat $O/ab.txt; tail -3 $O/ab.err
