#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/allegro_drop_autopsy.py --seeds 104,201,219,234 --out gpurun_out/r05t/product 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05x_product.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "closed_loop_behaviour and allegro" -s 2>&1 | grep -v amdgpu.ids | tail -22 | tee gpurun_out/r05x_test.txt
