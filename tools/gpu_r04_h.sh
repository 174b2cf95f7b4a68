#!/bin/bash
# GPU call H (round 4): fixed-trip-count subtree sums (bit-identical) with / without the opaque lane id per step, against the
# previous build, on one box; crate bit-identity test
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04h; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "crate_overflow" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for ex in unitree_go2_trot unitree_go2_seq_jump unitree_h1_jog unitree_h1_loco allegro_reorient; do
  steps=100; [ "$ex" = "allegro_reorient" ] && steps=25
  for rep in 1 2; do
    for lib in libdialhip_base.so libdialhip.so libdialhip_laund.so; do
      DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$lib python bench.py --example $ex --steps $steps --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
    done
  done
done > $O/ab_subtree.txt 2>&1
cat $O/ab_subtree.txt
python tools/cpu_scaling.py > $O/cpu_scaling.txt 2>&1; OMP_PROC_BIND=close OMP_PLACES=cores python tools/cpu_scaling.py >> $O/cpu_scaling.txt 2>&1
cat $O/cpu_scaling.txt | grep -v "^$" | head -40
