#!/bin/bash
# round 6, call q: the whole GPU suite with durations + the default bench line on the same box
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06q; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu --durations=30 > $OUT/pytest_gpu.txt 2>&1; grep -E "passed|failed|error|Elapsed" $OUT/pytest_gpu.txt | tail -4
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; python -c "
import json; d=json.loads(open('$OUT/bench_n1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d.get('plan_ms'))"
