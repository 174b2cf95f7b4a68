#!/bin/bash
# all-env bench table on one box (second sample of the final build)
cd /root/repo; export TMPDIR=/tmp; OUT=gpurun_out/r04f2; mkdir -p $OUT
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err
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
python -c "
import json; d=json.load(open('$OUT/bench_n1.json')); print('strong_cfg5', json.dumps(d.get('strong_cfg5'))[:600])"
