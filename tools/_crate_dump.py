import numpy as np, sys, torch
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import oracle.oracle as O
from dial_mpc_amd import _lib
from conftest import setup_case, seeded_inputs
from test_gpu_crate import _poses, EX, _dev
N,H=2048,25
dc, env, model, task, cfg = setup_case(EX, N, H, per_rollout=True)
o32, o64 = O.Oracle(model, task, cfg, np.float32), O.Oracle(model, task, cfg, np.float64)
import os
ctx = _lib.Context(model, task, cfg, lib_path=(_lib.IEEE_LIB_PATH if os.environ.get("CRATE_IEEE") else None))
out={}
for pose in (1,4):
    q,qd=_poses(env,o64)[pose]
    s0,_,_=o32.env_reset(q,qd)
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=pose, Ybar_scale=0.2)
    o = ctx.reverse_once(_dev(s0), _dev(Ybar), _dev(sigma), _dev(eps))
    sc = ctx.debug_scratch()
    out[f's0_{pose}']=s0
    for k in ("rewss","qss","qdss","Y0s"): out[f'{k}_{pose}']=sc[k]
np.savez_compressed('/root/repo/gpurun_out/crate_dbg2.npz', **out)
print('saved')
