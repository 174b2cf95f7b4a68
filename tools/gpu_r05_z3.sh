#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z3
( time python -m pytest tests/ -x -q -m gpu --durations=12 ) > gpurun_out/r05z3/gputest_full.txt 2>&1
tail -25 gpurun_out/r05z3/gputest_full.txt
