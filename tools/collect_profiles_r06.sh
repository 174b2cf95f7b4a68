#!/bin/bash
# Round-6 measurement set (ONE box, the build of the last kernel commit) -> gpurun_out/r06p/, copied to profiles/r06_* afterwards.
#   1. bench lines: headline (300 steps), the five BASELINE.json configs AT THE SIZES BASELINE.md STATES (cfg 2: seq-jump N=1024 H=16,
#      cfg 3: H1 jog N=2048 H=16, cfg 4: Allegro N=4096 H=24, cfg 5's batch on one GPU: the headline line's strong_cfg5), the examples'
#      own sizes, the crate scenes, Go2 on the capacity-dimension kernel (force_generic), the sharded path forced (N = 2048 / 8192)
#   2. default-policy N sweep
#   3. rocprofv3 --kernel-trace --stats of the headline, cfg 2, cfg 3, cfg 4 and N = 65536
#   4. PMC passes (all seven) of the headline, cfg 2, cfg 3, cfg 4 (fewer steps), N = 8192, N = 65536 -> JSON + per-form issue price
#   5. lone-wavefront section cycles (profiling build) of Go2 / pair / H1 / Allegro; Allegro wave times; Allegro per-iteration kernel times
#   6. resource notes of every shipped kernel (tools/isa/disasm_lib.py), issue microbenchmark
# usage: tools/collect_profiles_r06.sh [fast]     (fast: skips 4's large batches and 5's Allegro)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; export GRAFT_REPO_ROOT=$ROOT
OUT=$ROOT/gpurun_out/r06p; mkdir -p $OUT; cd $ROOT
FAST=${1:-}
line() { python -c "
import json,sys
try:
    d=json.load(open('$1')); m=d['iteration_modes']; vi=d['roofline'].get('valu_issue') or {}
    print('$2'.ljust(46), 'rollouts/s', str(round(d['value'])).rjust(9), ' ms/iter full', round(d['ms_per_step'],4), 'lean', round(m['ms_per_step_lean'],4), ' kernel', round(d['roofline']['avg_kernel_ms'],4),
          ' plan p50/p95', round(d['plan_latency_ms']['p50'],2), round(d['plan_latency_ms']['p95'],2), '(%d ticks)' % d['plan_latency_ms']['ticks'],
          ' issue frac', None if not vi else round(vi['frac'],3), 'at 2 cyc', None if not vi else round(vi['frac_at_2_cycles'],3))
except Exception as e: print('$2', 'FAILED', e)"; }
b() { # label, file, args...
  local lab=$1 f=$2; shift 2
  python bench.py "$@" > $OUT/$f 2> $OUT/${f%.json}.err || tail -5 $OUT/${f%.json}.err
  line $OUT/$f "$lab" | tee -a $OUT/bench_all_envs.txt
}
: > $OUT/bench_all_envs.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/issue tools/ubench/issue.hip 2>/dev/null && /tmp/issue > $OUT/ubench_issue.txt 2>&1
python tools/isa/disasm_lib.py dial_mpc_amd/csrc/libdialhip.so /tmp/isa_lib > $OUT/isa_resources.txt 2>&1
# ---- 4. PMC passes -> JSON -> issue price
pmc() { # tag, example, N, H, kernel-substring-of-the-shipped-listing, listing, extra bench args...
  local tag=$1 ex=$2 N=$3 H=$4 ksub=$5 lst=$6; shift 6
  PMC_BENCH_ARGS="--example $ex --nsample-per-gpu $N --hsample $H $*" bash tools/pmc_passes.sh r06p/pmc_$tag > $OUT/pmc_passes_$tag.log 2>&1
  python tools/pmc_to_json.py $OUT/pmc_$tag $OUT/pmc_$tag.json $ex $N $H > /dev/null 2>&1
  python tools/isa/price_mix.py $OUT/ubench_issue.txt /tmp/isa_lib/$lst "$ksub" $OUT/pmc_$tag.json $OUT/issue_price_$tag.json > /dev/null 2>&1
  python -c "
import json; p=json.load(open('$OUT/pmc_$tag.json')); q=json.load(open('$OUT/issue_price_$tag.json'))
print('$tag', 'VALU/wave-step', round(p['valu_insts_per_wave_env_step']), 'mix', {k[:5]: round(v,3) for k,v in (p['valu_mix'] or {}).items()}, 'lanes/inst', p.get('valu_active_lanes_per_inst'), 'cycles per VALU W1..W4', [round(x,2) for x in q.get('cycles_per_valu_inst_pmc_weighted_W1_W4', q['cycles_per_valu_inst_static_W1_W4'])], 'wave time', {k[:12]: round(v,3) for k,v in p['wave_time_breakdown'].items()})" | tee -a $OUT/pmc_summary.txt
  find $OUT/pmc_$tag -name "*.db" -delete 2>/dev/null; find $OUT/pmc_$tag -name "*kernel_trace.csv" -delete 2>/dev/null; find $OUT/pmc_$tag -name "*agent_info.csv" -delete 2>/dev/null
}
: > $OUT/pmc_summary.txt
pmc unitree_go2_trot unitree_go2_trot 2048 16 "rollout_kernelI.*TopoGo2L.*EELi1ELi3ELb0ELb0EEv" co1.s
pmc unitree_go2_seq_jump_N1024 unitree_go2_seq_jump 1024 16 "rollout_kernelI.*TopoGo2L.*EELi1ELi3ELb0ELb0EEv" co1.s
pmc unitree_h1_jog unitree_h1_jog 2048 16 "rollout_kernelI.*TopoH1L.*EELi4ELi3ELb0ELb0EEv" co2.s
if [ "$FAST" != "fast" ]; then
  pmc allegro_reorient_N4096 allegro_reorient 4096 24 "rollout_kernelI.*TopoAllegroL.*EELi9ELi3ELb1ELb0EEv" co4.s --steps 4
  pmc unitree_go2_trot_N8192 unitree_go2_trot 8192 16 "rollout_kernel2I.*EELi4ELi2ELb1ELb1EEv" co8.s
  pmc unitree_go2_trot_N65536 unitree_go2_trot 65536 16 "rollout_kernel2I.*EELi4ELi2ELb1ELb1EEv" co8.s --steps 8
fi
# (the bench lines below read their PMC / price files from profiles/: install this run's)
for t in unitree_go2_trot unitree_h1_jog; do cp $OUT/pmc_$t.json profiles/r06_pmc_$t.json; cp $OUT/issue_price_$t.json profiles/r06_issue_price_$t.json; done
cp $OUT/pmc_unitree_go2_seq_jump_N1024.json profiles/r06_pmc_unitree_go2_seq_jump_N1024.json; cp $OUT/issue_price_unitree_go2_seq_jump_N1024.json profiles/r06_issue_price_unitree_go2_seq_jump_N1024.json
if [ "$FAST" != "fast" ]; then
  for t in allegro_reorient_N4096 unitree_go2_trot_N8192 unitree_go2_trot_N65536; do cp $OUT/pmc_$t.json profiles/r06_pmc_$t.json; cp $OUT/issue_price_$t.json profiles/r06_issue_price_$t.json; done
fi
b "headline go2_trot N=2048 H=16"            bench_n1.json --steps 300 --warmup 30
b "cfg2 go2_seq_jump N=1024 H=16 (BASELINE)" bench_cfg2.json --example unitree_go2_seq_jump --nsample-per-gpu 1024 --hsample 16 --steps 200 --warmup 20 --no-cpu-baseline --no-strong-cfg5
b "cfg3 h1_jog N=2048 H=16 (BASELINE)"       bench_cfg3.json --example unitree_h1_jog --nsample-per-gpu 2048 --hsample 16 --steps 200 --warmup 20 --no-cpu-baseline --no-strong-cfg5
b "cfg4 allegro N=4096 H=24 (BASELINE)"      bench_cfg4.json --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 20 --warmup 3 --ticks 60 --no-cpu-baseline --no-strong-cfg5
b "go2_seq_jump example (N=2048 H=20)"       bench_ex_seq_jump.json --example unitree_go2_seq_jump --steps 100 --warmup 10 --ticks 50 --no-cpu-baseline --no-strong-cfg5
b "h1_jog example (N=2048 H=25)"             bench_ex_h1_jog.json --example unitree_h1_jog --steps 100 --warmup 10 --ticks 50 --no-cpu-baseline --no-strong-cfg5
b "h1_loco example (N=2048 H=20)"            bench_ex_h1_loco.json --example unitree_h1_loco --steps 100 --warmup 10 --ticks 50 --no-cpu-baseline --no-strong-cfg5
b "allegro example (N=2048 H=20)"            bench_ex_allegro.json --example allegro_reorient --steps 30 --warmup 3 --ticks 100 --no-cpu-baseline --no-strong-cfg5
b "go2_crate_climb example (N=2048 H=25)"    bench_ex_crate_climb.json --example unitree_go2_crate_climb --steps 100 --warmup 10 --ticks 40 --no-cpu-baseline --no-strong-cfg5
b "h1_push_crate example (N=2048 H=24)"      bench_ex_push_crate.json --example unitree_h1_push_crate --steps 100 --warmup 10 --ticks 40 --no-cpu-baseline --no-strong-cfg5
b "go2_trot on the capacity-dimension kernel" bench_go2_generic.json --steps 100 --warmup 10 --ticks 20 --no-cpu-baseline --no-strong-cfg5 --option force_generic=1
b "go2_trot N=2048 sharded path forced"      bench_n1_force_sharded.json --steps 200 --warmup 20 --ticks 20 --no-cpu-baseline --no-strong-cfg5 --force-sharded
b "go2_trot N=8192 fused"                    bench_n8192.json --steps 200 --warmup 20 --ticks 20 --no-cpu-baseline --no-strong-cfg5 --nsample-per-gpu 8192
b "go2_trot N=8192 sharded path forced"      bench_n8192_force_sharded.json --steps 200 --warmup 20 --ticks 20 --no-cpu-baseline --no-strong-cfg5 --nsample-per-gpu 8192 --force-sharded
python -c "
import json; d=json.load(open('$OUT/bench_n1.json'))['strong_cfg5']; print('cfg5 batch N=65536 on ONE GPU (strong_cfg5)'.ljust(46), 'rollouts/s', round(d['value']), ' ms/iter full', round(d['ms_per_step'],3), 'lean', round(d['ms_per_step_lean'],3), ' kernel', round(d['avg_rollout_kernel_ms'],3), ' issue frac', d.get('valu_issue_frac'))" | tee -a $OUT/bench_all_envs.txt
run() { python bench.py --steps 60 --warmup 8 --no-cpu-baseline --ticks 2 --no-strong-cfg5 --full-only "${@:2}" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'ms/step', round(d['ms_per_step'],4), 'Mroll/s', round(d['value']/1e6,3))"; }
for N in 256 1024 2048 2304 2560 3072 4096 5120 6144 8192 16384 32768 65536; do run "N=$N" --nsample-per-gpu $N; done > $OUT/n_sweep_default.txt 2>&1
cat $OUT/n_sweep_default.txt
# ---- 3. rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp
ks() { # name, bench args...
  local nm=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_$nm -o k -- python $ROOT/bench.py --ticks 5 --full-only --no-cpu-baseline --no-strong-cfg5 "$@" > $OUT/ks_$nm.log 2>&1
  find $OUT/ks_$nm -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$nm.csv \;
  rm -rf $OUT/ks_$nm
}
ks headline --steps 100 --warmup 10
ks cfg2 --example unitree_go2_seq_jump --nsample-per-gpu 1024 --hsample 16 --steps 100 --warmup 10
ks cfg3 --example unitree_h1_jog --nsample-per-gpu 2048 --hsample 16 --steps 100 --warmup 10
ks cfg4 --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 10 --warmup 2
ks go2_N65536 --nsample-per-gpu 65536 --steps 30 --warmup 5
cd $ROOT
# ---- 5. lone-wavefront section cycles
for ex in unitree_go2_trot unitree_h1_jog; do
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py $ex > $OUT/sections_${ex}_cycles.txt 2>&1
done
if [ "$FAST" != "fast" ]; then
  DIAL_HIP_LIB=$ROOT/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py allegro_reorient > $OUT/sections_allegro_reorient_cycles.txt 2>&1
  python tools/allegro_iteration_times.py allegro_reorient 36 > $OUT/allegro_iteration_times.txt 2>&1
fi
du -sh $OUT; cat $OUT/pmc_summary.txt
