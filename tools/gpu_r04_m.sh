#!/bin/bash
# GPU call M (round 4): Go2 quad stage with collision / contact Jacobian / constraint rows fused in (q3) against the stage alone (q2)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04m; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "go2" > $O/tests_go2.txt 2>&1; tail -3 $O/tests_go2.txt
tools/ab_bench.sh dial_mpc_amd/csrc/libdialhip_q2.so dial_mpc_amd/csrc/libdialhip.so unitree_go2_trot unitree_go2_seq_jump > $O/ab.txt 2>&1
cat $O/ab.txt
for n in 8192 65536; do
  steps=40; [ $n -ge 16384 ] && steps=15
  for lib in libdialhip_q2.so libdialhip.so; do
    DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib python bench.py --nsample-per-gpu $n --steps $steps --warmup 4 --ticks 2 --no-cpu-baseline --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('go2 N=$n', '$lib', 'rollouts/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4))"
  done
done > $O/go2_large.txt 2>&1
cat $O/go2_large.txt
