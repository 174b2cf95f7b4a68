#!/bin/bash
# round 6, call n: Allegro -- J^T f / H zones from the row lanes' registers, diagonal adds folded into 16-byte copies: parity + A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06n; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "allegro or Allegro" > $OUT/pytest_allegro.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_allegro.txt | tail -3
ab() { ex=$1; shift
  for rep in 1 2 3; do for lib in libdialhip_base.so libdialhip.so; do
    DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --example $ex --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex $*', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
  done; done
}
ab allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12 | tee $OUT/ab_allegro_cfg4.txt
ab allegro_reorient --steps 20 | tee $OUT/ab_allegro_example.txt
