#!/bin/bash
# round 5, call j: register / SGPR pressure of the pair kernels -- A (previous commit: ab_nofuse is older; use libs) vs B (stage-local
# lane scope in smooth_quad2: the working tree's libdialhip.so) vs C (+ lane scope in the solver) vs D (B + per-step launder)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "two_samples or pair_kernel or config5" > gpurun_out/r05j/test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05j/test.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r05j/test.log | tail -4
run() {  # label, lib, extra args
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$2 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:3}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"
}
for rep in 1 2; do
  for N in 2560 4096 8192 65536; do
    run "N=$N B smooth-scope   " libdialhip.so --nsample-per-gpu $N
    run "N=$N C +solver-scope  " ab_scopeC.so --nsample-per-gpu $N
    run "N=$N D +step launder  " ab_launderD.so --nsample-per-gpu $N
  done
done 2>&1 | tee gpurun_out/r05j/ab_scopes.txt
for N in 256 2048; do run "N=$N B pair_mode=2" libdialhip.so --nsample-per-gpu $N --option pair_mode=2; run "N=$N C pair_mode=2" ab_scopeC.so --nsample-per-gpu $N --option pair_mode=2; done 2>&1 | tee -a gpurun_out/r05j/ab_scopes.txt
