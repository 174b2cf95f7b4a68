#!/bin/bash
# GPU call Y2 (round 4): push crate on the row layout (the H1's tree in registers, the crate as a one-lane phase)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04y2; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_push_crate.py -m gpu -q -x -k "not closed_loop and not distribution" > $O/tests.txt 2>&1; tail -6 $O/tests.txt
timeout 100 python tools/ab_time.py tools/gpu_r04_y2_cases.txt 3 > $O/ab.txt 2> $O/ab.err; cat $O/ab.txt; tail -3 $O/ab.err
