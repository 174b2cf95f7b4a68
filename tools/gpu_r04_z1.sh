#!/bin/bash
# GPU call Z1 (round 4): the full GPU suite + smoke() on the final build
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04z; mkdir -p $O
timeout 800 python -m pytest tests/ -m gpu -q -s > $O/gputest_full.txt 2>&1; tail -4 $O/gputest_full.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
