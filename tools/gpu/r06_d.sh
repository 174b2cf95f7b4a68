#!/bin/bash
# round 6, call d: DPP-operand sweeps (solver_reg), min/max bracket in the gen kernels, sink-common flag -- FULL GPU suite + A/B of every robot
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06d; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
bash tools/ab_bench.sh dial_mpc_amd/csrc/libdialhip_base.so dial_mpc_amd/csrc/libdialhip.so unitree_go2_trot unitree_h1_jog unitree_h1_loco unitree_go2_crate_climb unitree_h1_push_crate 2>&1 | grep -v "^unitree_.*ab_" | tee $OUT/ab_all.txt
DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_go2_trot > $OUT/sections_go2.txt 2>&1
head -27 $OUT/sections_go2.txt
