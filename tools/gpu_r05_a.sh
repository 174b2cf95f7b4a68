#!/bin/bash
# round 5, call a: two samples per wavefront -- bit-identity test, then the A/B against the one-sample kernels (dial_options.pair_mode)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "two_samples_per_wavefront" > gpurun_out/r05a/test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05a/test.log
tail -15 gpurun_out/r05a/test.log
run() {  # label, extra args
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"
}
for rep in 1 2; do
  for N in 256 1024 2048 4096 8192 65536; do
    run "N=$N pair" --nsample-per-gpu $N
    run "N=$N one " --nsample-per-gpu $N --option pair_mode=1
  done
done 2>&1 | tee gpurun_out/r05a/ab.txt
run "seqjump pair" --example unitree_go2_seq_jump | tee -a gpurun_out/r05a/ab.txt
run "seqjump one " --example unitree_go2_seq_jump --option pair_mode=1 | tee -a gpurun_out/r05a/ab.txt
