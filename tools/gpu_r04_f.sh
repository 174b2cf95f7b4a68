#!/bin/bash
# GPU call F (round 4): crate kernels after the GEMM write-back fix; Go2 large-batch build with 16 wavefronts per CU
# (8 per workgroup, 128-VGPR budget, opaque lane id per step) against the shipped 12-per-CU build, N sweep
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04f; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "crate_overflow or crate_env_step or push_crate_env_reset" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for ex in unitree_go2_crate_climb unitree_h1_push_crate; do
  python bench.py --example $ex --steps 60 --warmup 8 --ticks 20 --no-cpu-baseline --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
done > $O/crate.txt 2>&1
cat $O/crate.txt
for n in 4096 8192 16384 65536; do
  steps=40; [ $n -ge 16384 ] && steps=15
  for rep in 1 2; do
    for lib in libdialhip.so libdialhip_w8.so; do
      DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib python bench.py --nsample-per-gpu $n --steps $steps --warmup 4 --ticks 2 --no-cpu-baseline --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('go2 N=$n', '$lib', 'rollouts/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4))"
    done
  done
done > $O/go2_large.txt 2>&1
cat $O/go2_large.txt
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_go2_crate_climb 2048 25 > $O/sections_unitree_go2_crate_climb_cycles.txt 2>&1
head -30 $O/sections_unitree_go2_crate_climb_cycles.txt
