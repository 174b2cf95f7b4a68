#!/bin/bash
# GPU call X (round 4): the second Newton iteration on the first one's factor when the active set did not change
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04x; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rollout_matches_oracle or reverse_once_matches_oracle_stagewise or golden_fixtures or full_size_properties_go2 or stress_parity or (full_size_oracle_parity and go2_trot)" > $O/tests.txt 2>&1; tail -8 $O/tests.txt
timeout 240 python tools/ab_time.py tools/gpu_r04_x_cases.txt 3 > $O/ab.txt 2> $O/ab.err; cat $O/ab.txt; tail -3 $O/ab.err
timeout 120 python bench.py --force-sharded --nsample-per-gpu 8192 --steps 30 --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 > $O/bench_sharded_8192.json 2>$O/bench_sharded.err; python -c "
import json; d=json.loads(open('$O/bench_sharded_8192.json').read().strip().splitlines()[-1]); print('force-sharded N=8192:', d['value'], d['ms_per_step'])"
