#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; export GRAFT_REPO_ROOT=/root/repo
mkdir -p gpurun_out/r04f; PMC_PASSES="1 2 3 4 5 6 7" bash tools/pmc_passes.sh r04f/pmc_go2_n2048 > gpurun_out/r04f/pmc_passes_go2_n2048.log 2>&1
python tools/pmc_summary.py gpurun_out/r04f/pmc_go2_n2048 > gpurun_out/r04f/pmc_unitree_go2_trot.txt 2>&1
python tools/pmc_to_json.py gpurun_out/r04f/pmc_go2_n2048 gpurun_out/r04f/pmc_unitree_go2_trot.json unitree_go2_trot 2048 16 > /dev/null 2>&1
find gpurun_out/r04f -name "*.db" -delete 2>/dev/null; find gpurun_out/r04f -path "*pass*" -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out/r04f -name "*agent_info.csv" -delete 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r04f/pmc_unitree_go2_trot.json')); print(json.dumps({k:v for k,v in d.items() if k!='counters' and k!='kernels'},indent=1))"
