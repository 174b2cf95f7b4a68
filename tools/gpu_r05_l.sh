#!/bin/bash
# round 5, call l: the DPP-operand L D L^T in EVERY register solver (reg_chol -> reg_chol_solve2) vs the v_readlane formulation (-DDIAL_CHOL_DPP=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05l
bash tools/ab_bench.sh dial_mpc_amd/csrc/libdialhip.so dial_mpc_amd/csrc/ab_cholreadlane.so unitree_go2_trot unitree_h1_jog unitree_h1_loco allegro_reorient unitree_go2_crate_climb unitree_h1_push_crate 2>&1 | tee gpurun_out/r05l/ab_chol_dpp.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "stagewise or rollout_matches or two_samples" > gpurun_out/r05l/test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05l/test.log
grep -E "passed|failed|FAILED|rc=|Error" gpurun_out/r05l/test.log | tail -4
