#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05y
run() {  # label, extra args
  python bench.py --warmup 5 --no-cpu-baseline --ticks 5 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4))"
}
{
for rep in 1 2; do
  run "N=2048 one rollout per wavefront (default)" --steps 200
  run "N=2048 two rollouts per wavefront (pair_mode=2)" --steps 200 --option pair_mode=2
  run "N=2304 default" --steps 100 --nsample-per-gpu 2304
  run "N=2304 pair_mode=2" --steps 100 --nsample-per-gpu 2304 --option pair_mode=2
  run "N=1024 default" --steps 100 --nsample-per-gpu 1024
  run "N=1024 pair_mode=2" --steps 100 --nsample-per-gpu 1024 --option pair_mode=2
done
} 2>&1 | tee gpurun_out/r05y/ab_pair_small_batches_final.txt
