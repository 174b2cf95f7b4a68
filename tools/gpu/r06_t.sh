#!/bin/bash
# round 6, call t: the whole GPU suite (planner replay test with the oracle's own 1-ulp envelope) + default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06t; mkdir -p $OUT; cd $ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "planner_replay" -s > $OUT/pytest_replay.txt 2>&1; grep -E "^tick|passed|failed" $OUT/pytest_replay.txt | tail -8
timeout 900 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.txt | tail -4
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; python -c "
import json; d=json.loads(open('$OUT/bench_n1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])"
