#!/bin/bash
# round 6, call f: phase-local lane scopes (zero scratch on the cfg 3 / cfg 4 kernels) -- parity of H1 / Allegro / crate + A/B on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06f; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "h1 or H1 or allegro or Allegro" > $OUT/pytest_h1_allegro.txt 2>&1; tail -3 $OUT/pytest_h1_allegro.txt
ab() { # example, extra args...
  ex=$1; shift
  for rep in 1 2 3; do for lib in libdialhip_base.so libdialhip.so; do
    DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --example $ex --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex $*', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
  done; done
}
ab unitree_h1_jog --nsample-per-gpu 2048 --hsample 16 --steps 100 | tee $OUT/ab_h1_cfg3.txt
ab unitree_h1_jog --steps 100 | tee $OUT/ab_h1_example.txt
ab allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12 | tee $OUT/ab_allegro_cfg4.txt
ab allegro_reorient --steps 20 | tee $OUT/ab_allegro_example.txt
