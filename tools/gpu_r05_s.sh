#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05s
run() {  # label, lib, extra args
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$2 python bench.py --steps 25 --warmup 3 --no-cpu-baseline --ticks 20 --no-strong-cfg5 "${@:3}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
}
for rep in 1 2; do
  for lib in libdialhip.so libdialhip_ieee.so ab_ieee_contract.so ab_fast_nocontract.so; do
    run "allegro example $lib" $lib --example allegro_reorient
  done
done 2>&1 | tee gpurun_out/r05s/allegro_flags_perf.txt
for lib in libdialhip.so libdialhip_ieee.so; do run "allegro cfg4 $lib" $lib --example allegro_reorient --nsample-per-gpu 4096 --hsample 24; done 2>&1 | tee -a gpurun_out/r05s/allegro_flags_perf.txt
