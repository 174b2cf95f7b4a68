#!/bin/bash
# Round-4 measurement set (run on the GPU box through gpurun): the headline bench line, the other envs, the crate examples,
# N sweep with the 16-wavefronts-per-CU large-batch build, rocprofv3 kernel stats, PMC passes (HBM traffic, instruction counts,
# stall breakdown) for the headline and the two crate kernels, per-section cycles, the transition survey.
# usage: tools/collect_profiles_r04.sh  -> gpurun_out/r04/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/r04
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
for n in 256 1024 2047 2048 4096 8192 16384 65536; do
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --nsample-per-gpu $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$n rollouts/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4), 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))"
done > $OUT/n_sweep.txt
python -c "
import json,glob,os
for f in sorted(glob.glob('$OUT/bench_n1*.json')):
    d=json.load(open(f)); print(os.path.basename(f), 'rollouts/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'lean', round(d['iteration_modes']['ms_per_step_lean'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))" > $OUT/bench_all_envs.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o k -- python $ROOT/bench.py --steps 100 --warmup 10 --ticks 5 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_crate -o k -- python $ROOT/bench.py --example unitree_go2_crate_climb --steps 40 --warmup 5 --ticks 3 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats_crate.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_push -o k -- python $ROOT/bench.py --example unitree_h1_push_crate --steps 40 --warmup 5 --ticks 3 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats_push.log 2>&1
cd $ROOT
find $OUT/kstats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_kernel_stats.csv \;
find $OUT/kstats_crate -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_unitree_go2_crate_climb.csv \;
find $OUT/kstats_push -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_unitree_h1_push_crate.csv \;
bash tools/pmc_passes.sh r04/pmc_go2_n2048 > $OUT/pmc_passes_go2_n2048.log 2>&1
python tools/pmc_summary.py $OUT/pmc_go2_n2048 > $OUT/pmc_unitree_go2_trot.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_go2_n2048 $OUT/pmc_unitree_go2_trot.json unitree_go2_trot 2048 16 > /dev/null 2>&1
PMC_PASSES="1 2 3 4" PMC_BENCH_ARGS="--nsample-per-gpu 65536 --steps 6" bash tools/pmc_passes.sh r04/pmc_go2_n65536 > $OUT/pmc_passes_go2_n65536.log 2>&1
python tools/pmc_summary.py $OUT/pmc_go2_n65536 > $OUT/pmc_unitree_go2_trot_N65536.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_go2_n65536 $OUT/pmc_unitree_go2_trot_N65536.json unitree_go2_trot 65536 16 > /dev/null 2>&1
PMC_PASSES="1 2 3 4 6" PMC_BENCH_ARGS="--example unitree_go2_crate_climb --steps 8" bash tools/pmc_passes.sh r04/pmc_crate > $OUT/pmc_passes_crate.log 2>&1
python tools/pmc_summary.py $OUT/pmc_crate > $OUT/pmc_unitree_go2_crate_climb.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_crate $OUT/pmc_unitree_go2_crate_climb.json unitree_go2_crate_climb 2048 25 > /dev/null 2>&1
PMC_PASSES="1 2 3 4 6" PMC_BENCH_ARGS="--example unitree_h1_push_crate --steps 8" bash tools/pmc_passes.sh r04/pmc_push > $OUT/pmc_passes_push.log 2>&1
python tools/pmc_summary.py $OUT/pmc_push > $OUT/pmc_unitree_h1_push_crate.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_push $OUT/pmc_unitree_h1_push_crate.json unitree_h1_push_crate 2048 24 > /dev/null 2>&1
for a in "unitree_go2_trot 2048 16" "unitree_h1_jog 2048 25" "unitree_go2_crate_climb 2048 25" "unitree_h1_push_crate 2048 24" "allegro_reorient 2048 20"; do
  set -- $a
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py $1 $2 $3 > $OUT/sections_$1_cycles.txt 2>/dev/null
done
python tools/transition_survey.py --no-dist > $OUT/transition_parity.txt 2>&1
DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/wave_times.py allegro_reorient 2048 20 9 > $OUT/wave_times.txt 2>&1
DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/wave_times.py allegro_reorient 4096 24 9 >> $OUT/wave_times.txt 2>&1
for ex in unitree_h1_jog allegro_reorient; do
  PMC_PASSES="1 2 3 4" PMC_BENCH_ARGS="--example $ex --steps 8" bash tools/pmc_passes.sh r04/pmc_$ex > $OUT/pmc_passes_$ex.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_$ex > $OUT/pmc_$ex.txt 2>&1
done
python tools/pmc_to_json.py $OUT/pmc_unitree_h1_jog $OUT/pmc_unitree_h1_jog.json unitree_h1_jog 2048 25 > /dev/null 2>&1
python tools/pmc_to_json.py $OUT/pmc_allegro_reorient $OUT/pmc_allegro_reorient.json allegro_reorient 2048 20 > /dev/null 2>&1
rm -rf $OUT/kstats $OUT/kstats_crate $OUT/kstats_push
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -path "*pass*" -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*agent_info.csv" -delete 2>/dev/null
du -sh $OUT; cat $OUT/bench_all_envs.txt; cat $OUT/n_sweep.txt
