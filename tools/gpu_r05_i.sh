#!/bin/bash
# round 5, call i: fused v_fmac_f32_dpp in the pair kernels -- gates, then the A/B against the build of the previous commit (ab_nofuse.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "two_samples or pair_kernel or rollout_queue or config5 or time_sliced" > gpurun_out/r05i/test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05i/test.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r05i/test.log | tail -6
run() {  # label, lib, extra args
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/$2 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:3}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"
}
for rep in 1 2; do
  for N in 2560 4096 8192 16384 65536; do
    run "N=$N fused  " libdialhip.so --nsample-per-gpu $N
    run "N=$N unfused" ab_nofuse.so --nsample-per-gpu $N
  done
done 2>&1 | tee gpurun_out/r05i/ab_fused.txt
