#!/bin/bash
# Round-2 measurement set (run on the GPU box through gpurun): bench lines of every env, N sweep, rocprofv3 kernel
# stats, PMC passes, per-section cycles.  usage: tools/collect_profiles.sh  -> gpurun_out/r02/
set -u
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r02
mkdir -p $OUT
cd $ROOT
python bench.py --steps 300 --warmup 30 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
for ex in unitree_go2_seq_jump unitree_h1_jog unitree_h1_loco; do
  python bench.py --example $ex --steps 100 --warmup 10 --no-cpu-baseline --ticks 30 > $OUT/bench_n1_$ex.json 2>/dev/null
done
python bench.py --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 20 --warmup 3 --ticks 10 --no-cpu-baseline > $OUT/bench_n1_allegro_reorient_N4096_H24.json 2>/dev/null
python bench.py --example allegro_reorient --steps 20 --warmup 3 --ticks 10 --no-cpu-baseline > $OUT/bench_n1_allegro_reorient_example.json 2>/dev/null
for n in 256 1024 2047 2048 4096 8192 16384; do
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --ticks 2 --nsample-per-gpu $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=$n rollouts/s', round(d['value']), 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))"
done > $OUT/n_sweep.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --ticks 3 --scaling strong --nsample-total 65536 > $OUT/bench_n1_strong_65536.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --ticks 20 --force-sharded > $OUT/bench_n1_force_sharded.json 2>/dev/null
# rocprofv3 kernel trace + stats of the default bench command
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -o k -- python $ROOT/bench.py --steps 100 --warmup 10 --ticks 5 --no-cpu-baseline > $OUT/kstats.log 2>&1
cd $ROOT
find $OUT/kstats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
bash tools/pmc_passes.sh r02/pmc > $OUT/pmc_passes.log 2>&1
python tools/pmc_summary.py $OUT/pmc > $OUT/pmc_rollout_kernel.txt 2>&1
for a in "unitree_go2_trot 2048 16" "unitree_h1_jog 2048 25" "unitree_h1_loco 2048 20" "allegro_reorient 2048 20"; do
  set -- $a
  DIAL_HIP_LIB=$ROOT/build/libdialhip_prof.so python tools/profile_sections.py $1 $2 $3 > $OUT/sections_$1.txt 2>/dev/null
done
# per-rollout start / end times and solver iteration counts (profile build): what sets the launch duration
(for a in "unitree_go2_trot 2048 16 1" "unitree_go2_trot 8192 16 4" "unitree_h1_jog 2048 25 3" "allegro_reorient 2048 20 9" "allegro_reorient 4096 24 9"; do
  set -- $a
  DIAL_HIP_LIB=$ROOT/build/libdialhip_prof.so python tools/wave_times.py $1 $2 $3 $4 2>/dev/null
done) > $OUT/wave_times.txt
ls -la $OUT | head -40
