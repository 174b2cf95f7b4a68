#!/bin/bash
# PMC passes (instruction counts, wave-time split, HBM bytes) of the rollout kernel for the other BASELINE robots -> bench.py's roofline.valu_issue_frac
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $GRAFT_REPO_ROOT
for ex in unitree_h1_jog allegro_reorient unitree_go2_seq_jump; do
  st=20; [ $ex = allegro_reorient ] && st=6
  PMC_PASSES="1 3 4" PMC_BENCH_ARGS="--example $ex --steps $st" bash tools/pmc_passes.sh r05p2/pmc_$ex > gpurun_out/r05p2_$ex.log 2>&1
  H=$(python -c "import yaml; from dial_mpc_amd.utils.io_utils import get_example_path; print(yaml.safe_load(open(get_example_path('$ex.yaml')))['Hsample'])")
  python tools/pmc_to_json.py gpurun_out/r05p2/pmc_$ex gpurun_out/r05p2/pmc_$ex.json $ex 2048 $H > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
done
find gpurun_out/r05p2 -name "*.db" -delete 2>/dev/null; find gpurun_out/r05p2 -name "*kernel_trace.csv" -delete 2>/dev/null; find gpurun_out/r05p2 -name "*agent_info.csv" -delete 2>/dev/null
ls -la gpurun_out/r05p2/*.json; du -sh gpurun_out/r05p2
