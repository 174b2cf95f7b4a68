#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r04sec; mkdir -p $O
for a in "unitree_h1_push_crate 2048 24" "allegro_reorient 2048 20"; do set -- $a
  DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_prof.so timeout 100 python tools/profile_sections.py $1 $2 $3 > $O/sections_$1_cycles.txt 2>/dev/null
done
head -27 $O/sections_unitree_h1_push_crate_cycles.txt
