#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04fs; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 60 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-strong-cfg5 --ticks 10 > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']), round(d['ms_per_step'],4))"
