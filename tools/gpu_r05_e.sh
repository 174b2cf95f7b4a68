#!/bin/bash
# round 5, call e: where does the pair kernel start to win (N sweep, pair_mode 2 vs 1), then the whole GPU suite with timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
run() {  # label, extra args
  python bench.py --steps 60 --warmup 8 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"
}
for N in 2560 3072 4096 5120 6144 8192 12288 16384 32768; do
  run "N=$N pair" --nsample-per-gpu $N --option pair_mode=2
  run "N=$N one " --nsample-per-gpu $N --option pair_mode=1
done 2>&1 | tee gpurun_out/r05e/sweep.txt
timeout 1500 python -m pytest tests -m gpu -q -x --durations=40 > gpurun_out/r05e/suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05e/suite.log
tail -60 gpurun_out/r05e/suite.log
