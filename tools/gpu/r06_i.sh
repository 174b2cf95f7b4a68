#!/bin/bash
# round 6, call i: fused K4 (weights + gather map, one-launch weighted sums / mean action) -- full GPU suite + sharded overhead + headline
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06i; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shard or lean or reverse or k4 or K4 or weights or sigma or std" > $OUT/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_gpu.txt | tail -3
for N in 2048 8192; do for mode in "" "--force-sharded"; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --nsample-per-gpu $N $mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['iteration_modes']; print('N=$N', '$mode' or 'fused', 'full', round(m['ms_per_step_full'],4), 'lean', round(m['ms_per_step_lean'],4), 'plan', round(m['ms_per_step_plan_pattern'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'kernel_lean', round(m['avg_rollout_kernel_ms_lean'],4))"
done; done | tee $OUT/sharded_overhead.txt
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_n1.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_n1.json')); print('headline', round(d['value']), d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['plan_latency_ms'], 'cfg5', d['strong_cfg5']['value'], d['strong_cfg5']['ms_per_step'])"
