#!/bin/bash
# round 6, call a: extended issue-rate microbenchmark, baseline bench line, BASELINE-size cfg 2 / cfg 3 lines, lone-wave section cycles
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/r06a; mkdir -p $OUT; cd $ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/issue tools/ubench/issue.hip 2>/dev/null && /tmp/issue > $OUT/ubench_issue.txt 2>&1
tail -45 $OUT/ubench_issue.txt
python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || tail -20 $OUT/bench_n1.err
python bench.py --example unitree_go2_seq_jump --nsample-per-gpu 1024 --hsample 16 --steps 200 --warmup 20 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_cfg2.json 2>/dev/null
python bench.py --example unitree_h1_jog --nsample-per-gpu 2048 --hsample 16 --steps 200 --warmup 20 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_cfg3.json 2>/dev/null
python - <<PY
import json,glob,os
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), 'rollouts/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_kernel_ms'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2), d['plan_latency_ms']['ticks'])
    except Exception as e: print(f, 'FAILED', e)
PY
DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_go2_trot > $OUT/sections_go2.txt 2>&1
head -30 $OUT/sections_go2.txt
