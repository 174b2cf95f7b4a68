#!/bin/bash
# GPU call O (round 4): time-sliced rollout queue for Allegro batches beyond the resident set -- bit-identity test, cfg 4 A/B
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "time_sliced or rollout_queue or relay" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for rep in 1 2; do
  for opt in "" "--option no_slice=1" "--option slice_steps=2" "--option slice_steps=5"; do
    timeout 300 python bench.py --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12 --warmup 3 --no-cpu-baseline --ticks 2 --no-strong-cfg5 $opt 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro cfg4 [$opt]', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4))"
  done
done > $O/ab_slice.txt 2>&1
cat $O/ab_slice.txt
