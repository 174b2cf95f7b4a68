#!/bin/bash
# GPU call P (round 4): lag-based issue priority for the Allegro (rollouts of data-dependent length) -- tests, example A/B
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "allegro or time_sliced" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2 3; do
  for opt in "" "--option no_lag_priority=1"; do
    timeout 300 python bench.py --example allegro_reorient --steps 30 --warmup 5 --no-cpu-baseline --ticks 20 --no-strong-cfg5 $opt 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro example [$opt]', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
  done
done > $O/ab_lag.txt 2>&1
cat $O/ab_lag.txt
