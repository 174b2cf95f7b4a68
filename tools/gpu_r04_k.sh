#!/bin/bash
# GPU call K (round 4): quad stage variants (q1 = first version; q2 = opaque lane id + dump-word stores + branch-free actuator +
# constant contact frames + fixed-trip node2u; q2b = q2 with branchy stores) on one box
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04k; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "go2" > $O/tests_go2.txt 2>&1; tail -3 $O/tests_go2.txt
for ex in unitree_go2_trot unitree_go2_seq_jump unitree_h1_jog; do
  for rep in 1 2 3; do
    for lib in libdialhip_q1.so libdialhip.so libdialhip_q2b.so; do
      DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib python bench.py --example $ex --steps 100 --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
    done
  done
done > $O/ab.txt 2>&1
cat $O/ab.txt
