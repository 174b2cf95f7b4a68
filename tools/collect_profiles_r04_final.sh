#!/bin/bash
# Round-4 FINAL measurement set (one box, the build of the last commit; a trimmed tools/collect_profiles_r04.sh: the GPU budget of the
# session's end): headline bench line, the other envs, N sweep in one process, rocprofv3 kernel stats and PMC passes of the headline
# and of crate climb, per-section cycles.   usage: tools/collect_profiles_r04_final.sh  -> gpurun_out/r04f/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $ROOT
python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
for ex in unitree_go2_seq_jump unitree_h1_jog unitree_h1_loco; do
  python bench.py --example $ex --steps 100 --warmup 10 --no-cpu-baseline --ticks 30 --no-strong-cfg5 > $OUT/bench_n1_$ex.json 2>/dev/null
done
python bench.py --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 20 --warmup 3 --ticks 10 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_n1_allegro_reorient_N4096_H24.json 2>/dev/null
python bench.py --example allegro_reorient --steps 30 --warmup 3 --ticks 40 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_n1_allegro_reorient_example.json 2>/dev/null
for ex in unitree_go2_crate_climb unitree_h1_push_crate; do
  python bench.py --example $ex --steps 100 --warmup 10 --ticks 40 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_n1_${ex}_example.json 2>/dev/null
done
python -c "
import json,glob,os
for f in sorted(glob.glob('$OUT/bench_n1*.json')):
    d=json.load(open(f)); print(os.path.basename(f), 'rollouts/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))" > $OUT/bench_all_envs.txt
python tools/ab_time.py tools/n_sweep_cases.txt 2 > $OUT/n_sweep.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o k -- python $ROOT/bench.py --steps 100 --warmup 10 --ticks 5 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_crate -o k -- python $ROOT/bench.py --example unitree_go2_crate_climb --steps 40 --warmup 5 --ticks 3 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats_crate.log 2>&1
cd $ROOT
find $OUT/kstats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_kernel_stats.csv \;
find $OUT/kstats_crate -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_unitree_go2_crate_climb.csv \;
PMC_PASSES="1 2 3 4" bash tools/pmc_passes.sh r04f/pmc_go2_n2048 > $OUT/pmc_passes_go2_n2048.log 2>&1
python tools/pmc_summary.py $OUT/pmc_go2_n2048 > $OUT/pmc_unitree_go2_trot.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_go2_n2048 $OUT/pmc_unitree_go2_trot.json unitree_go2_trot 2048 16 > /dev/null 2>&1
PMC_PASSES="1 3 4" PMC_BENCH_ARGS="--example unitree_go2_crate_climb --steps 8" bash tools/pmc_passes.sh r04f/pmc_crate > $OUT/pmc_passes_crate.log 2>&1
python tools/pmc_summary.py $OUT/pmc_crate > $OUT/pmc_unitree_go2_crate_climb.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_crate $OUT/pmc_unitree_go2_crate_climb.json unitree_go2_crate_climb 2048 25 > /dev/null 2>&1
for a in "unitree_go2_trot 2048 16" "unitree_go2_crate_climb 2048 25" "unitree_h1_jog 2048 25"; do
  set -- $a
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py $1 $2 $3 > $OUT/sections_$1_cycles.txt 2>/dev/null
done
rm -rf $OUT/kstats $OUT/kstats_crate
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -path "*pass*" -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*agent_info.csv" -delete 2>/dev/null
du -sh $OUT; cat $OUT/bench_all_envs.txt; cat $OUT/n_sweep.txt
