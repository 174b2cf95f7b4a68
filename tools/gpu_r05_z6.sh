#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z6
for a in "unitree_go2_trot 2048 16" "allegro_reorient 2048 20" "unitree_h1_jog 2048 16"; do set -- $a
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so python tools/profile_sections.py $1 $2 $3 > gpurun_out/r05z6/sections_$1_cycles.txt 2>/dev/null
done
head -40 gpurun_out/r05z6/sections_unitree_go2_trot_cycles.txt
