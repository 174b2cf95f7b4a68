#!/bin/bash
# round 6, call c: act2tau in the MO lanes + q/qd stores moved -- parity, A/B, sections, PMC passes (all six) of the headline
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06c; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "go2 or Go2 or pair or two_samples or shard or lean or bit" > $OUT/pytest_go2.txt 2>&1; tail -3 $OUT/pytest_go2.txt
bash tools/ab_bench.sh dial_mpc_amd/csrc/libdialhip_base.so dial_mpc_amd/csrc/libdialhip.so unitree_go2_trot unitree_go2_seq_jump 2>&1 | grep -v "^unitree_.*ab_" | tee $OUT/ab_pre_ctrl.txt
for N in 256 1024 4096 8192 65536; do for lib in libdialhip_base.so libdialhip.so; do
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only --nsample-per-gpu $N 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$N', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))"
done; done | tee $OUT/ab_pre_ctrl_nsweep.txt
DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_go2_trot > $OUT/sections_go2.txt 2>&1
head -27 $OUT/sections_go2.txt
bash tools/pmc_passes.sh r06c/pmc_go2_n2048 > $OUT/pmc_passes_go2_n2048.log 2>&1
python tools/pmc_to_json.py $OUT/pmc_go2_n2048 $OUT/pmc_unitree_go2_trot.json unitree_go2_trot 2048 16 | head -60
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -path "*pass*" -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*agent_info.csv" -delete 2>/dev/null
