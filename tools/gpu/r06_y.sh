#!/bin/bash
# round 6, call y: constant lane masks (s_mov + v_cndmask, no v_cmp) in the register factorisations: bit comparison + A/B every robot + parity
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06y; mkdir -p $OUT; cd $ROOT
L=dial_mpc_amd/csrc
{ python tools/bit_compare.py $L/libdialhip_base.so $L/libdialhip.so 2>&1 | grep unitree; python tools/bit_compare.py $L/libdialhip_base.so $L/libdialhip.so unitree_h1_jog 2048 16 2>&1 | grep unitree
  python tools/bit_compare.py $L/libdialhip_base.so $L/libdialhip.so unitree_go2_trot 8192 16 2>&1 | grep unitree; python tools/bit_compare.py $L/libdialhip_base.so $L/libdialhip.so allegro_reorient 2048 20 2>&1 | grep allegro; } | tee $OUT/bit_compare.txt
ab() { ex=$1; shift
  for rep in 1 2 3; do for lib in libdialhip_base.so libdialhip.so; do
    DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --example $ex --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex $*', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'value', round(d['value']))"
  done; done
}
{
ab unitree_go2_trot --steps 200
ab unitree_go2_seq_jump --nsample-per-gpu 1024 --steps 200
ab unitree_h1_jog --nsample-per-gpu 2048 --hsample 16 --steps 100
ab unitree_h1_loco --steps 100
ab unitree_go2_crate_climb --steps 50
ab unitree_h1_push_crate --steps 50
ab allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12
ab unitree_go2_trot --nsample-per-gpu 8192 --steps 100
ab unitree_go2_trot --nsample-per-gpu 65536 --steps 20
} | tee $OUT/ab_all.txt
