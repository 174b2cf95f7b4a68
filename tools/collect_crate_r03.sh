#!/bin/bash
# Round-3 measurements of the crate-climb and push-crate examples (generic kernel instantiation) and a refresh of the Allegro example line:
# bench lines, rocprofv3 kernel stats, per-section cycles.  Run on the GPU box through gpurun -> gpurun_out/r03c/
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03c
mkdir -p $OUT
cd $ROOT
python bench.py --example unitree_go2_crate_climb --steps 100 --warmup 10 --ticks 40 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_n1_unitree_go2_crate_climb_example.json 2>/dev/null
python bench.py --example unitree_h1_push_crate --steps 100 --warmup 10 --ticks 40 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_n1_unitree_h1_push_crate_example.json 2>/dev/null
python bench.py --example allegro_reorient --steps 30 --warmup 3 --ticks 40 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_n1_allegro_reorient_example.json 2>/dev/null
DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_go2_crate_climb 2048 25 > $OUT/sections_unitree_go2_crate_climb_cycles.txt 2>&1
DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py unitree_h1_push_crate 2048 24 > $OUT/sections_unitree_h1_push_crate_cycles.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_crate -o k -- python $ROOT/bench.py --example unitree_go2_crate_climb --steps 40 --warmup 5 --ticks 3 --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats_crate.log 2>&1
cd $ROOT
find $OUT/kstats_crate -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_unitree_go2_crate_climb.csv \;
rm -rf $OUT/kstats_crate
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_push -o k -- python $ROOT/bench.py --example unitree_h1_push_crate --steps 40 --warmup 5 --ticks 3 --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats_push.log 2>&1
cd $ROOT
find $OUT/kstats_push -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_unitree_h1_push_crate.csv \;
rm -rf $OUT/kstats_push
ls $OUT
