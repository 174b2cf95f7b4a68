#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z9
( time python -m pytest tests/ -x -q -m gpu -k "allegro or smoke" ) > gpurun_out/r05z9/gputest_allegro.txt 2>&1; tail -6 gpurun_out/r05z9/gputest_allegro.txt
python bench.py --example allegro_reorient --steps 30 --warmup 3 --ticks 40 --no-cpu-baseline --no-strong-cfg5 > gpurun_out/r05z9/bench_n1_allegro_reorient_example.json 2>/dev/null
python bench.py --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 20 --warmup 3 --ticks 10 --no-cpu-baseline --no-strong-cfg5 > gpurun_out/r05z9/bench_n1_allegro_reorient_N4096_H24.json 2>/dev/null
python -c "
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r05z9/bench_n1*.json')):
    d=json.load(open(f)); print(os.path.basename(f), 'rollouts/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
