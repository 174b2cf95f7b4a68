#!/bin/bash
# GPU call S (round 4): more thresholds of the lag-based priority (b: 1.0 / 1.2 / 1.4, d: 1.1 / 1.3 / 1.5, e: 1.0 / 1.3 / 1.6); the Allegro
# full-size transition that sits at 1.22 x the gate, on the IEEE build (no fast-math flags)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04s; mkdir -p $O
for rep in 1 2; do
  for lib in libdialhip_lagb.so libdialhip_lagd.so libdialhip_lage.so; do
    DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib timeout 300 python bench.py --example allegro_reorient --steps 30 --warmup 5 --no-cpu-baseline --ticks 20 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro example $lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
  done
done > $O/ab_lag.txt 2>&1
cat $O/ab_lag.txt
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_ieee.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "full_size_oracle_parity and allegro" > $O/tests_ieee.txt 2>&1; tail -3 $O/tests_ieee.txt; grep "per transition\|knife\|where 1" $O/tests_ieee.txt
