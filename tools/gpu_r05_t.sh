#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/allegro_drop_autopsy.py --seeds 0,11,31,36 --out gpurun_out/r05t/product 2>&1 | tee gpurun_out/r05t_product.txt
DIAL_HIP_LIB=$PWD/dial_mpc_amd/csrc/libdialhip_ieee.so python tools/allegro_drop_autopsy.py --seeds 0,11,31,36 --out gpurun_out/r05t/ieee 2>&1 | tee gpurun_out/r05t_ieee.txt
