#!/bin/bash
# Round-5 measurement set (one box, the build of the last commit): headline bench line, the other envs, default-policy N sweep, rocprofv3
# kernel stats of the headline and of N = 65536, PMC passes of N = 65536 and N = 8192 (the pair kernels), per-section cycles of the pair
# kernel.   usage: tools/collect_profiles_r05.sh  -> gpurun_out/r05p/
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/r05p
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
cat $OUT/bench_all_envs.txt
run() {  # label, extra args
  python bench.py --steps 60 --warmup 8 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"
}
for N in 256 1024 2048 2304 2560 3072 4096 5120 8192 16384 32768 65536; do run "N=$N" --nsample-per-gpu $N; done > $OUT/n_sweep.txt 2>&1
cat $OUT/n_sweep.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o k -- python $ROOT/bench.py --steps 100 --warmup 10 --ticks 5 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats65536 -o k -- python $ROOT/bench.py --nsample-per-gpu 65536 --steps 30 --warmup 5 --ticks 2 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats65536.log 2>&1
cd $ROOT
find $OUT/kstats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_kernel_stats.csv \;
find $OUT/kstats65536 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_go2_N65536.csv \;
PMC_PASSES="1 2 3 4 5 6" PMC_BENCH_ARGS="--nsample-per-gpu 65536 --steps 8" bash tools/pmc_passes.sh r05p/pmc_go2_n65536 > $OUT/pmc_passes_n65536.log 2>&1
python tools/pmc_summary.py $OUT/pmc_go2_n65536 > $OUT/pmc_unitree_go2_trot_N65536.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_go2_n65536 $OUT/pmc_unitree_go2_trot_N65536.json unitree_go2_trot 65536 16 > /dev/null 2>&1
PMC_PASSES="1 3 4" PMC_BENCH_ARGS="--nsample-per-gpu 8192 --steps 20" bash tools/pmc_passes.sh r05p/pmc_go2_n8192 > $OUT/pmc_passes_n8192.log 2>&1
python tools/pmc_to_json.py $OUT/pmc_go2_n8192 $OUT/pmc_unitree_go2_trot_N8192.json unitree_go2_trot 8192 16 > /dev/null 2>&1
PMC_PASSES="1 3 4" bash tools/pmc_passes.sh r05p/pmc_go2_n2048 > $OUT/pmc_passes_go2_n2048.log 2>&1
python tools/pmc_to_json.py $OUT/pmc_go2_n2048 $OUT/pmc_unitree_go2_trot.json unitree_go2_trot 2048 16 > /dev/null 2>&1
rm -rf $OUT/kstats $OUT/kstats65536
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -path "*pass*" -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*agent_info.csv" -delete 2>/dev/null
du -sh $OUT; head -3 $OUT/kernel_stats_go2_N65536.csv | cut -c1-60,330-440
