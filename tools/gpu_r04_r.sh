#!/bin/bash
# GPU call R (round 4): thresholds of the lag-based priority (a: 0.85 / 1.05 / 1.25, b: 1.0 / 1.2 / 1.4, c: 0.9 / 1.0 / 1.1), Allegro full-size test
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04r; mkdir -p $O
for rep in 1 2; do
  for lib in libdialhip.so libdialhip_lagb.so libdialhip_lagc.so; do
    DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib timeout 300 python bench.py --example allegro_reorient --steps 30 --warmup 5 --no-cpu-baseline --ticks 20 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro example $lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
  done
done > $O/ab_lag.txt 2>&1
cat $O/ab_lag.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "full_size_oracle_parity and allegro" > $O/tests.txt 2>&1; tail -3 $O/tests.txt; grep "per transition\|knife" $O/tests.txt
