#!/bin/bash
# A/B: the > 64-row line search's loop conditions and bracket update on the scalar forms (libdialhip.so) against HEAD (ab_pcold.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05y
run() {  # label, lib, extra args
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$2 python bench.py --warmup 3 --no-cpu-baseline --ticks 20 --no-strong-cfg5 "${@:3}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
}
{
for rep in 1 2 3; do for lib in ab_pcold.so libdialhip.so; do run "push crate" $lib --steps 60 --example unitree_h1_push_crate; done; done
for rep in 1 2; do for lib in ab_pcold.so libdialhip.so; do run "crate climb" $lib --steps 60 --example unitree_go2_crate_climb; done; done
for rep in 1; do for lib in ab_pcold.so libdialhip.so; do run "go2 headline" $lib --steps 200; done; done
} 2>&1 | tee gpurun_out/r05y/ab_rows_gt64_scalar.txt
