#!/bin/bash
# round 5, call f: pair kernels with the interleaved mean trajectory -- gates, then the N sweep against the one-sample kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --durations=8 -k "two_samples or pair_kernel or rollout_queue or time_sliced or ieee_build or go2" > gpurun_out/r05f/test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05f/test.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r05f/test.log | tail -12
run() {  # label, extra args
  python bench.py --steps 60 --warmup 8 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"
}
for N in 256 1024 2048 2560 3072 4096 6144 8192 16384 65536; do
  run "N=$N pair" --nsample-per-gpu $N --option pair_mode=2
  run "N=$N one " --nsample-per-gpu $N --option pair_mode=1
done 2>&1 | tee gpurun_out/r05f/sweep.txt
