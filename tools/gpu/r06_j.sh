#!/bin/bash
# round 6, call j: velocity sweep restructured + 16-byte staging -- Go2 parity, A/B, sections
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06j; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "go2 or Go2 or pair or two_samples or shard or lean or bit" > $OUT/pytest_go2.txt 2>&1; grep -E "passed|failed|error" $OUT/pytest_go2.txt | tail -3
bash tools/ab_bench.sh dial_mpc_amd/csrc/libdialhip_base.so dial_mpc_amd/csrc/libdialhip.so unitree_go2_trot unitree_go2_seq_jump 2>&1 | grep -v "^unitree_.*ab_" | tee $OUT/ab.txt
for N in 256 8192 65536; do for lib in libdialhip_base.so libdialhip.so; do
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only --nsample-per-gpu $N 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$N', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))"
done; done | tee $OUT/ab_nsweep.txt
