#!/bin/bash
# Round-3 measurement set (run on the GPU box through gpurun): bench lines of every env, N sweep, rocprofv3 kernel
# stats per robot, PMC passes (Go2 at N = 2048 and N = 65536 incl. the dynamic instruction mix; HBM traffic for H1 and
# Allegro), per-section cycles.  usage: tools/collect_profiles_r03.sh [quick]  -> gpurun_out/r03/
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03
mkdir -p $OUT
cd $ROOT
python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
for ex in unitree_go2_seq_jump unitree_h1_jog unitree_h1_loco; do
  python bench.py --example $ex --steps 100 --warmup 10 --no-cpu-baseline --ticks 30 --no-strong-cfg5 > $OUT/bench_n1_$ex.json 2>/dev/null
done
python bench.py --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 20 --warmup 3 --ticks 10 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_n1_allegro_reorient_N4096_H24.json 2>/dev/null
python bench.py --example allegro_reorient --steps 30 --warmup 3 --ticks 40 --no-cpu-baseline --no-strong-cfg5 > $OUT/bench_n1_allegro_reorient_example.json 2>/dev/null
for n in 256 1024 2047 2048 4096 8192 16384 65536; do
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --nsample-per-gpu $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$n rollouts/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))"
done > $OUT/n_sweep.txt
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --ticks 20 --no-strong-cfg5 --force-sharded > $OUT/bench_n1_force_sharded.json 2>/dev/null
# rocprofv3 kernel trace + stats: the default bench command, then H1 and Allegro
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o k -- python $ROOT/bench.py --steps 100 --warmup 10 --ticks 5 --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_h1 -o k -- python $ROOT/bench.py --example unitree_h1_jog --steps 50 --warmup 5 --ticks 3 --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats_h1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats_allegro -o k -- python $ROOT/bench.py --example allegro_reorient --steps 20 --warmup 3 --ticks 3 --no-cpu-baseline --no-strong-cfg5 > $OUT/kstats_allegro.log 2>&1
cd $ROOT
find $OUT/kstats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/kstats_h1 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_unitree_h1_jog.csv \;
find $OUT/kstats_allegro -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_allegro_reorient.csv \;
# PMC: Go2 at the headline batch and at the saturated batch, H1 / Allegro
bash tools/pmc_passes.sh r03/pmc_go2_n2048 > $OUT/pmc_passes_go2_n2048.log 2>&1
python tools/pmc_summary.py $OUT/pmc_go2_n2048 > $OUT/pmc_go2_n2048.txt 2>&1
PMC_BENCH_ARGS="--nsample-per-gpu 65536 --steps 6" bash tools/pmc_passes.sh r03/pmc_go2_n65536 > $OUT/pmc_passes_go2_n65536.log 2>&1
python tools/pmc_summary.py $OUT/pmc_go2_n65536 > $OUT/pmc_go2_n65536.txt 2>&1
if [ "${1:-}" != "quick" ]; then
  PMC_BENCH_ARGS="--example unitree_h1_jog" bash tools/pmc_passes.sh r03/pmc_h1 > $OUT/pmc_passes_h1.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_h1 > $OUT/pmc_unitree_h1_jog.txt 2>&1
  PMC_BENCH_ARGS="--example allegro_reorient --steps 6" bash tools/pmc_passes.sh r03/pmc_allegro > $OUT/pmc_passes_allegro.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_allegro > $OUT/pmc_allegro_reorient.txt 2>&1
fi
for a in "unitree_go2_trot 2048 16" "unitree_h1_jog 2048 25" "unitree_h1_loco 2048 20" "allegro_reorient 2048 20"; do
  set -- $a
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py $1 $2 $3 > $OUT/sections_$1.txt 2>/dev/null
done
(for a in "unitree_go2_trot 2048 16 1" "allegro_reorient 2048 20 9"; do
  set -- $a
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/wave_times.py $1 $2 $3 $4 2>/dev/null
done) > $OUT/wave_times.txt
rm -rf $OUT/kstats $OUT/kstats_h1 $OUT/kstats_allegro $OUT/pmc_*/pass*/*/*.db 2>/dev/null
du -sh $OUT; ls $OUT | head -50
