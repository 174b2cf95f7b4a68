#!/bin/bash
# round 6, call s: Go2 changes -- parity tests + A/B against the build before (libdialhip_base.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06s; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "go2 or Go2 or pair or sharded or planner or closed_loop or crate" > $OUT/pytest_go2.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_go2.txt | tail -4
ab() { ex=$1; shift
  for rep in 1 2 3; do for lib in libdialhip_base.so libdialhip.so; do
    DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --example $ex --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex $*', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'value', round(d['value']))"
  done; done
}
{
ab unitree_go2_trot --steps 200
ab unitree_go2_seq_jump --nsample-per-gpu 1024 --steps 200
ab unitree_go2_trot --nsample-per-gpu 8192 --steps 100
ab unitree_go2_trot --nsample-per-gpu 65536 --steps 20
} | tee $OUT/ab_go2.txt
