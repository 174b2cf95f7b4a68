import os, sys, torch, yaml, numpy as np
sys.path.insert(0, os.getcwd())
from dial_mpc_amd.core.dial_core import MBDPI, load_dial_and_env
from dial_mpc_amd.utils.io_utils import get_example_path
cfgd = yaml.safe_load(open(get_example_path("allegro_reorient.yaml")))
cfgd["Nsample"], cfgd["Hsample"] = int(sys.argv[1]), int(sys.argv[2])
dc, _, env = load_dial_and_env(cfgd)
pl = MBDPI(dc, env, kernel_rng=True)
st = env.reset(0).packed
Y = torch.zeros((dc.Hnode + 1, pl.nu), dtype=torch.float32, device=pl.device)
res = {True: [], False: []}
for i in range(40):
    wb = (i % 2 == 0)
    pl.ctx.set_timing(True)
    _, Y2, _ = pl.reverse_once(st, None, Y, pl.sigma_control, eps=None, want_bars=wb)
    torch.cuda.synchronize()
    t, n = pl.ctx.rollout_ms(); pl.ctx.set_timing(False)
    res[wb].append(t / max(n, 1))
    if i % 2 == 1: Y = Y2     # every pair (full, lean) starts from the same plan; noise differs
print("N", sys.argv[1], "full", np.mean(res[True][2:]).round(3), "lean", np.mean(res[False][2:]).round(3), "n", len(res[True]) - 2)
