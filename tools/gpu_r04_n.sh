#!/bin/bash
# GPU call N (round 4): per-rollout timeline of the Allegro launches (example and BASELINE config 4), Go2 N sweep of the new build
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04n; mkdir -p $O
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so python tools/wave_times.py allegro_reorient 2048 20 9 > $O/wave_times.txt 2>&1
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so WAVE_TIMES_OUT=$O/wt_cfg4.npz python tools/wave_times.py allegro_reorient 4096 24 9 >> $O/wave_times.txt 2>&1
cat $O/wave_times.txt
for n in 256 1024 2047 2048 2304 4096 8192; do
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --nsample-per-gpu $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$n rollouts/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4), 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))"
done > $O/n_sweep.txt
cat $O/n_sweep.txt
