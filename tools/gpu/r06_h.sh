#!/bin/bash
# round 6, call h: (1) LDS bank skew of the pair kernels' workspaces: A/B + conflict counters; (2) sharded-path overhead on the current kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT; OUT=$ROOT/gpurun_out/r06h; mkdir -p $OUT; cd $ROOT
for rep in 1 2 3; do for lib in libdialhip.so libdialhip_skew0.so libdialhip_skew16.so; do for N in 8192 65536; do
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only --nsample-per-gpu $N 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$N', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))"
done; done; done | tee $OUT/ab_skew.txt
for lib in libdialhip_skew0.so libdialhip_skew16.so libdialhip.so; do
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/$lib PMC_PASSES="2 7" PMC_BENCH_ARGS="--nsample-per-gpu 65536 --steps 6" bash tools/pmc_passes.sh r06h/pmc_$lib > $OUT/pmc_$lib.log 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob('$OUT/pmc_$lib/pass*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'rollout_kernel' in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
d={k:v[0]/v[1] for k,v in acc.items()}
print('$lib', {k: round(d[k]/1e6,1) for k in ('SQ_LDS_BANK_CONFLICT','SQ_ACTIVE_INST_LDS','SQ_INSTS_LDS_LOAD' ,'SQ_INSTS_LDS_STORE','SQ_LDS_ADDR_CONFLICT','SQ_LDS_IDX_ACTIVE') if k in d}, 'conflict/active', round(d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_ACTIVE_INST_LDS',1),1),3))
PY
done | tee $OUT/lds_conflicts.txt
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -path "*pass*" -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT -name "*agent_info.csv" -delete 2>/dev/null
for N in 2048 8192; do for mode in "" "--force-sharded"; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --nsample-per-gpu $N $mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d['iteration_modes']; print('N=$N', '$mode' or 'fused', 'full', round(m['ms_per_step_full'],4), 'lean', round(m['ms_per_step_lean'],4), 'plan', round(m['ms_per_step_plan_pattern'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'kernel_lean', round(m['avg_rollout_kernel_ms_lean'],4))"
done; done | tee $OUT/sharded_overhead.txt
