#!/bin/bash
# PMC passes over the rollout kernel (separate rocprofv3 runs, kernel-trace only; see MI355X_MICROARCH.md).
# usage: [PMC_BENCH_ARGS="--nsample-per-gpu 65536 ..."] tools/pmc_passes.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --ticks 1 --full-only --no-cpu-baseline --no-strong-cfg5 ${PMC_BENCH_ARGS:-}"
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
            "SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
            "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED" \
            "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" ; do
  i=$((i+1))
  if [ -n "${PMC_PASSES:-}" ] && ! echo " $PMC_PASSES " | grep -q " $i "; then continue; fi   # PMC_PASSES="1 3 4": only those passes
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/pass$i -o p -- $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
ls -R $OUT | head -30
