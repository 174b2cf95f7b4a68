for ex in unitree_go2_trot unitree_h1_jog unitree_h1_loco allegro_reorient unitree_go2_seq_jump unitree_go2_crate_climb unitree_h1_push_crate; do
  python bench.py --example $ex --steps 100 --warmup 10 --no-cpu-baseline --ticks 20 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4), 'plan p50', round(d['plan_latency_ms']['p50'],3), 'p95', round(d['plan_latency_ms']['p95'],3))"
done
python bench.py --example allegro_reorient --nsample-per-gpu 4096 --hsample 24 --steps 20 --warmup 3 --no-cpu-baseline --ticks 3 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('allegro cfg4', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"
python bench.py --nsample-per-gpu 8192 --steps 50 --warmup 5 --no-cpu-baseline --ticks 3 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('go2 N=8192', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"
