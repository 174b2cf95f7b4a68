#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z5
( time python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r05z5/smoke.txt 2>&1; tail -4 gpurun_out/r05z5/smoke.txt
( time python bench.py ) > gpurun_out/r05z5/bench_default.txt 2>&1; tail -5 gpurun_out/r05z5/bench_default.txt | cut -c1-600
