#!/bin/bash
# GPU call G (round 4): the whole GPU suite, smoke, headline bench, crate benches
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/gputest_full.txt 2>&1; tail -15 $O/gputest_full.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 400 $O/bench_n1.json
for ex in unitree_go2_crate_climb unitree_h1_push_crate; do
  python bench.py --example $ex --steps 100 --warmup 10 --ticks 40 --no-cpu-baseline --no-strong-cfg5 > $O/bench_n1_${ex}_example.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/bench_n1_${ex}_example.json')); print('$ex', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2))"
done
