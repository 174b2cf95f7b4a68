#!/bin/bash
# round 6, call o: Allegro -- the three changes of call n one at a time (variant libraries built with DIAL_HIPCC_EXTRA)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06o; mkdir -p $OUT; cd $ROOT
ab() { ex=$1; shift
  for rep in 1 2; do for lib in ${LIBS}; do
    DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --example $ex --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex $*', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
  done; done
}
LIBS=${LIBS:-"libdialhip_base.so libdialhip_v_none.so libdialhip_v_jtf.so libdialhip_v_fold.so libdialhip_v_euler.so libdialhip.so"}
ab allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 12 | tee $OUT/ab_allegro_cfg4.txt
