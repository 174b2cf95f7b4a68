#!/bin/bash
# GPU call B (round 4): crate scenes on their own instantiations -- parity tests, A/B against the capacity-dimension kernel on
# the same box, HBM traffic (PMC), per-section cycles
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "crate or stagewise or env_reset_and_step or rollout_matches" > $O/tests_crate.txt 2>&1; tail -3 $O/tests_crate.txt
for ex in unitree_go2_crate_climb unitree_h1_push_crate; do
  for rep in 1 2; do
    for opt in "" "--option force_generic=1"; do
      python bench.py --example $ex --steps 60 --warmup 8 --ticks 20 --no-cpu-baseline --no-strong-cfg5 $opt 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex', '[$opt]', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
    done
  done
done > $O/ab_crate.txt 2>&1
cat $O/ab_crate.txt
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
PMC_PASSES="1 3 4" PMC_BENCH_ARGS="--example unitree_go2_crate_climb" bash tools/pmc_passes.sh r04b/pmc_crate_climb > $O/pmc_crate_climb.log 2>&1
PMC_PASSES="1 3 4" PMC_BENCH_ARGS="--example unitree_h1_push_crate" bash tools/pmc_passes.sh r04b/pmc_push_crate > $O/pmc_push_crate.log 2>&1
cd /root/repo
python tools/pmc_to_json.py $O/pmc_crate_climb $O/pmc_unitree_go2_crate_climb.json unitree_go2_crate_climb 2048 25 > /dev/null 2>&1
python tools/pmc_to_json.py $O/pmc_push_crate $O/pmc_unitree_h1_push_crate.json unitree_h1_push_crate 2048 24 > /dev/null 2>&1
python -c "
import json
for e in ('unitree_go2_crate_climb','unitree_h1_push_crate'):
    d=json.load(open('$O/pmc_%s.json'%e)); print(e, 'HBM MB/launch', d.get('hbm_bytes_per_launch',0)/1e6, d.get('wave_time_breakdown'), 'VALU/step', d.get('valu_insts_per_wave_env_step'))"
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_go2_crate_climb 2048 25 > $O/sections_unitree_go2_crate_climb_cycles.txt 2>&1
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_h1_push_crate 2048 24 > $O/sections_unitree_h1_push_crate_cycles.txt 2>&1
rm -rf $O/pmc_crate_climb/pass*/*/ 2>/dev/null; find $O -name "*.db" -delete 2>/dev/null
tail -30 $O/sections_unitree_go2_crate_climb_cycles.txt
