#!/bin/bash
# GPU call V (round 4): time-sliced queue for the Go2's large batches, K2 with a compile-time node count, crate climb on the
# quadruped register stage -- A/B timings in one process, the parity tests the three changes touch, per-rollout timelines
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04v; mkdir -p $O
timeout 240 python tools/ab_time.py tools/gpu_r04_v_cases.txt 2 > $O/ab.txt 2> $O/ab.err; tail -45 $O/ab.txt; tail -3 $O/ab.err
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --durations=12 \
  -k "time_sliced or rollout_queue or reverse_once_matches_oracle_stagewise or golden_fixtures or rollout_matches_oracle or config5" > $O/tests_parity.txt 2>&1; tail -20 $O/tests_parity.txt
timeout 240 python -m pytest tests/test_gpu_crate.py -m gpu -q --durations=8 \
  -k "generic_instantiation or env_step_and_rollouts or (full_size_oracle_parity and 0) or overflow" > $O/tests_crate.txt 2>&1; tail -12 $O/tests_crate.txt
for c in "unitree_go2_crate_climb 2048 25" "unitree_h1_push_crate 2048 24" "unitree_go2_trot 8192 16"; do
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so timeout 120 python tools/wave_times.py $c 2>/dev/null
done > $O/wave_times.txt 2>&1; cat $O/wave_times.txt
