#!/bin/bash
# GPU call U (round 4): Allegro on the row layout (static root + free object) -- parity tests, A/B against the phase version
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "allegro and not full_size" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2; do
  for lib in libdialhip_pre.so libdialhip.so; do
    DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib timeout 300 python bench.py --example allegro_reorient --steps 30 --warmup 5 --no-cpu-baseline --ticks 20 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro example $lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
    DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib timeout 300 python bench.py --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12 --warmup 3 --no-cpu-baseline --ticks 2 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro cfg4 $lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
  done
done > $O/ab.txt 2>&1
cat $O/ab.txt
