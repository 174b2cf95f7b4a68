#!/bin/bash
# usage: tools/isa/probe.sh <Dims> [WPB] [OCC] [QUEUE] [extra flags...]   e.g.  tools/isa/probe.sh DimsGo2 1 3 false
# writes build/isa/<Dims>_<WPB>_<OCC>_<QUEUE>.s and prints the kernel's resource summary
cd "$(dirname "$0")/../.."
D=${1:-DimsMax}; W=${2:-1}; O=${3:-3}; Q=${4:-false}; shift 4 2>/dev/null
mkdir -p build/isa
OUT=build/isa/${D}_${W}_${O}_${Q}.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDIAL_FUSED_DPP -fno-hip-fp32-correctly-rounded-divide-sqrt \
  -Xarch_device -freciprocal-math -Xarch_device -fapprox-func -Xarch_device -fno-slp-vectorize -Xarch_device -fno-honor-nans \
  -DPROBE_D=$D -DPROBE_WPB=$W -DPROBE_OCC=$O -DPROBE_QUEUE=$Q "$@" --cuda-device-only -S -o $OUT tools/isa/probe.hip || exit 1
grep -E "^\s+\.(sgpr_count|vgpr_count|sgpr_spill_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size):" $OUT
echo "instructions: $(grep -cE '^\s+(v_|s_|ds_|global_|buffer_|scratch_|flat_)' $OUT)  scratch ops: $(grep -cE '^\s+scratch_' $OUT)  v_readlane: $(grep -c v_readlane $OUT)  ds ops: $(grep -cE '^\s+ds_' $OUT)"
