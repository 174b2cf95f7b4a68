#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "two_samples or pair_kernel" > gpurun_out/r05k/test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05k/test.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r05k/test.log | tail -4
run() {  # label, extra args
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --ticks 20 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3), 'plan p50', round(d['plan_latency_ms']['p50'],3))"
}
for rep in 1 2 3; do
  run "N=2048 pair+inline" --option pair_mode=2
  run "N=2048 pair plain " --option pair_mode=2 --option no_mean_inline=1
  run "N=2048 one        " --option pair_mode=1
done 2>&1 | tee gpurun_out/r05k/ab_n2048.txt
run "seq_jump N=1024 default" --example unitree_go2_seq_jump | tee -a gpurun_out/r05k/ab_n2048.txt
run "seq_jump N=1024 pair   " --example unitree_go2_seq_jump --option pair_mode=2 | tee -a gpurun_out/r05k/ab_n2048.txt
