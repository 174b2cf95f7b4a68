#!/bin/bash
# round 5, call c: pair kernel bit-identity on the no-contraction build, the Go2 oracle-parity gates on the product build (now the
# pair kernel), A/B bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x --durations=15 -k "two_samples or ieee_build or go2" > gpurun_out/r05c/test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05c/test.log
tail -40 gpurun_out/r05c/test.log
run() {  # label, extra args
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"
}
for rep in 1 2; do
  for N in 256 1024 2048 4096 8192 65536; do
    run "N=$N pair" --nsample-per-gpu $N
    run "N=$N one " --nsample-per-gpu $N --option pair_mode=1
  done
done 2>&1 | tee gpurun_out/r05c/ab.txt
