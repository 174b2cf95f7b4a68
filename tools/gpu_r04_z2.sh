#!/bin/bash
# GPU call Z2 (round 4): the tests the last changes touch (spread launch, crate scenes' register stages), then the final measurement set
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04f; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_crate.py -m gpu -q -k "spread_launch or time_sliced or rollout_queue or env_step_and_rollouts or generic_instantiation or (crate_full_size_oracle_parity and 0) or overflow" > $O/tests_last_changes.txt 2>&1; tail -4 $O/tests_last_changes.txt
bash tools/collect_profiles_r04_final.sh > $O/collect.log 2>&1; tail -40 $O/collect.log
