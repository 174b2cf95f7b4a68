#!/bin/bash
# GPU call W (round 4): the mean trajectory interleaved with the first T wavefronts' own steps (Go2 large batches)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04w; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "time_sliced or rollout_queue or config5" > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 240 python tools/ab_time.py tools/gpu_r04_w_cases.txt 2 > $O/ab.txt 2> $O/ab.err; cat $O/ab.txt; tail -3 $O/ab.err
