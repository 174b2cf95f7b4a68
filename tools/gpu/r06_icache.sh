#!/bin/bash
# round 6: instruction-cache counters of the Allegro (16 k instructions) and the Go2 (8 k) rollout kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06ic; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "SQC_ICACHE|SQC_INST|ICACHE|SQ_IFETCH|SQ_WAIT_INST|SQC_" | head -40 > $OUT/counters.txt
for cfg in "allegro --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 6 --warmup 2" "go2 --steps 20 --warmup 3"; do
  tag=${cfg%% *}; args=${cfg#* }
  for CTRS in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
    n=$(echo $CTRS | cut -c1-12 | tr ' ' '_')
    rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/${tag}_$n -o p -- python $ROOT/bench.py --ticks 1 --full-only --no-cpu-baseline --no-strong-cfg5 $args > $OUT/${tag}_$n.log 2>&1
    python - <<PY
import csv,glob,collections
fs=glob.glob('$OUT/${tag}_$n/**/*counter_collection.csv', recursive=True)
acc=collections.defaultdict(float); cnt=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'rollout_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
print('$tag', {k:(v/max(cnt[k],1)) for k,v in acc.items()}, 'launches', dict(cnt))
PY
  done
done | tee $OUT/icache.txt
cat $OUT/counters.txt | head -20
