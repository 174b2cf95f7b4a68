#!/bin/bash
# A/B of two builds of the HIP library on the SAME box in one gpurun call (box-to-box variation is +-1-2 %):
# usage: tools/ab_bench.sh <libA.so> <libB.so> [examples...]; alternates A B A B A B and prints the kernel ms per run.
A=$1; B=$2; shift 2
EX=${@:-unitree_go2_trot unitree_h1_jog unitree_h1_loco allegro_reorient}
for ex in $EX; do
  steps=100; [ "$ex" = "allegro_reorient" ] && steps=25
  for rep in 1 2 3; do
    for lib in $A $B; do
      DIAL_HIP_LIB=$PWD/$lib python bench.py --example $ex --steps $steps --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
    done
  done
done
