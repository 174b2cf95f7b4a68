#!/bin/bash
# round 6, call w: the root's cdof_dot without signed-zero arithmetic in the row-layout stage (H1, H1 loco, Allegro, push crate): A/B + bit comparison + parity
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06w; mkdir -p $OUT; cd $ROOT
L=dial_mpc_amd/csrc
python tools/bit_compare.py $L/libdialhip_base.so $L/libdialhip.so unitree_h1_jog 2048 16 2>&1 | grep unitree | tee $OUT/bit_compare.txt
ab() { ex=$1; shift
  for rep in 1 2 3; do for lib in libdialhip_base.so libdialhip.so; do
    DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --example $ex --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex $*', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
  done; done
}
{
ab unitree_h1_jog --nsample-per-gpu 2048 --hsample 16 --steps 100
ab unitree_h1_loco --steps 100
ab unitree_h1_push_crate --steps 50
ab allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12
} | tee $OUT/ab_rows.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "h1 or H1 or allegro or Allegro or push" > $OUT/pytest_rows.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_rows.txt | tail -3
