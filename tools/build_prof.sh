#!/bin/bash
# profiling build of the HIP library: per-section cycle counters + per-rollout timestamps (tools/profile_sections.py, tools/wave_times.py)
cd "$(dirname "$0")/.."
DIAL_HIPCC_EXTRA="-DDIAL_PROFILE" DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so python -c "from dial_mpc_amd import _lib; _lib.build(force=True)"
