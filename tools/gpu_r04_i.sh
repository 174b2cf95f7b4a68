#!/bin/bash
# GPU call I (round 4): Allegro J^T f with a fixed trip count against the previous build; then the round's profile collection
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04i; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "crate_overflow or (allegro and (stagewise or rollout_matches))" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2; do
  for lib in libdialhip_base.so libdialhip.so; do
    DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib python bench.py --example allegro_reorient --steps 25 --warmup 5 --no-cpu-baseline --ticks 10 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro example', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
    DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib python bench.py --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12 --warmup 3 --no-cpu-baseline --ticks 2 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro cfg4', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
  done
done > $O/ab_allegro.txt 2>&1
cat $O/ab_allegro.txt
