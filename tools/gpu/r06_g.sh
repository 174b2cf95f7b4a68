#!/bin/bash
# round 6, call g: constant workspace entries once per wavefront in the queue kernels, lane scopes -- full parity file + Allegro / pair-queue A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06g; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_parity.txt 2>&1; tail -3 $OUT/pytest_parity.txt
ab() { ex=$1; shift
  for rep in 1 2 3; do for lib in libdialhip_base.so libdialhip.so; do
    DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --example $ex --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex $*', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
  done; done
}
ab allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12 | tee $OUT/ab_allegro_cfg4.txt
ab allegro_reorient --steps 20 | tee $OUT/ab_allegro_example.txt
for N in 4096 5120 6144 8192 16384 65536; do for lib in libdialhip_base.so libdialhip.so; do
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only --nsample-per-gpu $N 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$N', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'Mroll/s', round(d['value']/1e6,3))"
done; done | tee $OUT/ab_nsweep.txt
