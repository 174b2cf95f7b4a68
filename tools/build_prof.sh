#!/bin/bash
# profiling build of the HIP library: per-section cycle counters + per-rollout timestamps (tools/profile_sections.py, tools/wave_times.py)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-hip-fp32-correctly-rounded-divide-sqrt \
  -Xarch_device -freciprocal-math -Xarch_device -fapprox-func -Xarch_device -fno-slp-vectorize -Xarch_device -fno-honor-nans \
  -DDIAL_PROFILE -o dial_mpc_amd/csrc/libdialhip_prof.so dial_mpc_amd/csrc/dial_hip.hip
