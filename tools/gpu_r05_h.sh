#!/bin/bash
# round 5, call h: H1 launder A/B, Allegro closed-loop study on the GPU, the GPU suite again (time), the default bench line + kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $ROOT
bash tools/ab_bench.sh dial_mpc_amd/csrc/libdialhip.so dial_mpc_amd/csrc/ab_h1launder.so unitree_h1_jog > $OUT/ab_h1launder.txt 2>&1
cat $OUT/ab_h1launder.txt
python tools/allegro_closed_loop_study.py --mode philox-check --nsample 512 > $OUT/philox_check.txt 2>&1
python tools/allegro_closed_loop_study.py --mode gpu --nsample 512 --seeds 0:64 --ticks 40 --json $OUT/allegro_gpu_N512.json > $OUT/allegro_gpu_N512.txt 2>&1
python tools/allegro_closed_loop_study.py --mode gpu --nsample 2048 --seeds 0:64 --ticks 40 --json $OUT/allegro_gpu_N2048.json > $OUT/allegro_gpu_N2048.txt 2>&1
tail -2 $OUT/philox_check.txt; tail -1 $OUT/allegro_gpu_N512.txt; tail -1 $OUT/allegro_gpu_N2048.txt
( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > $OUT/suite.log 2>&1
echo "pytest rc=$?" >> $OUT/suite.log
grep -E "passed|failed|FAILED|rc=|^real" $OUT/suite.log | tail -8
python bench.py --steps 200 --warmup 20 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python -c "
import json; d=json.load(open('$OUT/bench_n1.json')); print('headline', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], 'issue', d['roofline']['valu_issue_frac'], 'cfg5', d['strong_cfg5']['value'], d['strong_cfg5']['ms_per_step'], d['strong_cfg5'].get('valu_issue_frac'), 'plan', d['plan_latency_ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o k -- python $ROOT/bench.py --steps 100 --warmup 10 --ticks 5 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats.log 2>&1
cd $ROOT
find $OUT/kstats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_kernel_stats.csv \;
rm -rf $OUT/kstats
head -4 $OUT/bench_n1_kernel_stats.csv
