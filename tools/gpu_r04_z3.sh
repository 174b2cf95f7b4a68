#!/bin/bash
# GPU call Z3 (round 4): the crate scenes' distribution-level and closed-loop tests on the final build (their register stages changed last)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04z3; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_push_crate.py tests/test_gpu_crate.py -m gpu -q -s -k "closed_loop or distribution or default_rule or transition" > $O/tests.txt 2>&1; tail -5 $O/tests.txt; grep -n "closed loop\|unwitnessed\|direct" $O/tests.txt | head
