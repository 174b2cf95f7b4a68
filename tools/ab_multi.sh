for rep in 1 2; do for lib in base ilp iterative nopost memb; do
  for ex in unitree_go2_trot unitree_h1_jog; do
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/ab_$lib.so python bench.py --example $ex --steps 100 --warmup 10 --no-cpu-baseline --ticks 2 --no-strong-cfg5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$ex', '$lib', 'kernel_ms', round(d['roofline']['avg_kernel_ms'],4))"
  done
done; done
