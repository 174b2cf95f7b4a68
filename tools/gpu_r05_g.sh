#!/bin/bash
# round 5, call g: the whole GPU suite (time budget), the issue-rate microbenchmark, default-policy N sweep, kernel stats + PMC of the
# pair kernel at N = 65536 and of the headline
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/r05g
mkdir -p $OUT
cd $ROOT
( time timeout 1500 python -m pytest tests -m gpu -q --durations=25 ) > $OUT/suite.log 2>&1
echo "pytest rc=$?" >> $OUT/suite.log
grep -E "passed|failed|FAILED|rc=|^real" $OUT/suite.log | tail -12
./tools/ubench/issue > $OUT/ubench_issue.txt 2>&1
cat $OUT/ubench_issue.txt
run() {  # label, extra args
  python bench.py --steps 60 --warmup 8 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"
}
for N in 256 1024 2048 2304 2560 3072 3500 4096 5120 8192 16384 32768 65536; do
  run "N=$N default" --nsample-per-gpu $N
done 2>&1 | tee $OUT/sweep_default.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats65536 -o k -- python $ROOT/bench.py --nsample-per-gpu 65536 --steps 30 --warmup 5 --ticks 2 --full-only --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats65536.log 2>&1
cd $ROOT
find $OUT/kstats65536 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_go2_N65536.csv \;
PMC_PASSES="1 2 3 4 5 6" PMC_BENCH_ARGS="--nsample-per-gpu 65536 --steps 8" bash tools/pmc_passes.sh r05g/pmc_go2_n65536 > $OUT/pmc_passes_n65536.log 2>&1
python tools/pmc_summary.py $OUT/pmc_go2_n65536 > $OUT/pmc_unitree_go2_trot_N65536.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_go2_n65536 $OUT/pmc_unitree_go2_trot_N65536.json unitree_go2_trot 65536 16 > /dev/null 2>&1
rm -rf $OUT/kstats65536
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -path "*pass*" -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*agent_info.csv" -delete 2>/dev/null
du -sh $OUT; head -5 $OUT/kernel_stats_go2_N65536.csv; cat $OUT/pmc_unitree_go2_trot_N65536.txt | head -40
