#!/bin/bash
# A/B: LLVM's max-ilp scheduling strategy (ab_maxilp.so) against the default (libdialhip.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05y
run() {  # label, lib, extra args
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$2 python bench.py --warmup 3 --no-cpu-baseline --ticks 10 --no-strong-cfg5 "${@:3}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
}
{
for rep in 1 2; do for lib in libdialhip.so ab_maxilp.so; do run "go2 headline" $lib --steps 200; done; done
for rep in 1; do for lib in libdialhip.so ab_maxilp.so; do run "go2 N=65536" $lib --steps 10 --nsample-per-gpu 65536; done; done
for rep in 1 2; do for lib in libdialhip.so ab_maxilp.so; do run "h1 jog" $lib --steps 100 --example unitree_h1_jog; done; done
for rep in 1 2; do for lib in libdialhip.so ab_maxilp.so; do run "allegro example" $lib --steps 25 --example allegro_reorient; done; done
for rep in 1; do for lib in libdialhip.so ab_maxilp.so; do run "push crate" $lib --steps 50 --example unitree_h1_push_crate; done; done
} 2>&1 | tee gpurun_out/r05y/ab_sched_max_ilp.txt
