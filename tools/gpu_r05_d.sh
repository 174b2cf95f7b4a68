#!/bin/bash
# round 5, call d: the Go2 gates on the pair kernel vs on the one-sample kernels (no -x: count), section profile of the lone pair wavefront
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
K="two_samples or ieee_build or (go2 and not relay and not time_sliced)"
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s --durations=10 -k "$K" > gpurun_out/r05d/test_pair.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05d/test_pair.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r05d/test_pair.log | tail -20
DIAL_TEST_OPTIONS="pair_mode=1" timeout 1500 python -m pytest tests/test_gpu_parity.py -q -s -k "go2 and not relay and not time_sliced and not two_samples" > gpurun_out/r05d/test_one.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05d/test_one.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/r05d/test_one.log | tail -20
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_go2_trot 2048 16 > gpurun_out/r05d/sections_pair.txt 2>&1
tail -60 gpurun_out/r05d/sections_pair.txt
