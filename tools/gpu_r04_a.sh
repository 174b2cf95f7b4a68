#!/bin/bash
# GPU call A (round 4): changed-switch tests, transition survey, trace-hook A/B, headline bench
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04a; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "relay or overflow or queue or split or recovers or timeout or crate_context" > $O/tests_switches.txt 2>&1; tail -3 $O/tests_switches.txt
python tools/transition_survey.py --ieee --only allegro --seeds 1 > $O/survey_allegro.txt 2>&1
python tools/transition_survey.py > $O/survey_all.txt 2>&1
tools/ab_bench.sh dial_mpc_amd/csrc/libdialhip.so dial_mpc_amd/csrc/libdialhip_notrace.so unitree_go2_trot unitree_h1_jog > $O/ab_trace.txt 2>&1
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -2 $O/ab_trace.txt; tail -c 600 $O/bench_n1.json
