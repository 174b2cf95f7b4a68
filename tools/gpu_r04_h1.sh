#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04h1; mkdir -p $O
for rep in 1 2; do for lib in ab_base.so ab_noreuse.so libdialhip.so; do
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib timeout 100 python bench.py --example unitree_h1_jog --steps 100 --warmup 10 --no-cpu-baseline --ticks 5 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('h1_jog bench.py $lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4))"
done; done > $O/ab.txt 2>&1; cat $O/ab.txt
