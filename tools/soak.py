#!/usr/bin/env python3
"""Soak: thousands of reverse_once iterations per env on one fixed state; the rewards must stay finite (exercises the
mean-trajectory relay / split launch hand-overs).  python tools/soak.py   (needs an MI355X)"""
import sys, time, numpy as np, torch
sys.path[:0]=['.','tests']
from conftest import setup_case, seeded_inputs
from dial_mpc_amd import _lib
for ex,N,H,iters in (("unitree_go2_trot",2048,16,20000),("unitree_h1_jog",2048,25,5000),("allegro_reorient",2048,20,600)):
    dc, env, model, task, cfg = setup_case(ex,N,H)
    ctx=_lib.Context(model,task,cfg)
    dev=lambda x: torch.as_tensor(np.ascontiguousarray(x,dtype=np.float32),device="cuda")
    s0,_,_=ctx.env_reset(dev(env._init_q),dev(np.zeros(model.nv)))
    _,sigma,Ybar=seeded_inputs(dc,model.nu,seed=0,Ybar_scale=0.1)
    Y=dev(Ybar); sg=dev(sigma); bad=0; t0=time.time()
    out=None
    for it in range(iters):
        out=ctx.reverse_once_rng(s0,Y,sg,1234,it,out=out)
        if it%500==499:
            r=out["rews"]
            if not bool(torch.isfinite(r).all()): bad+=1
    torch.cuda.synchronize()
    print(ex,"iterations",iters,"non-finite checks",bad,"wall %.1f s"%(time.time()-t0),"rews[-1]",float(out["rews"][-1]))
