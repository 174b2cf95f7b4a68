import os, sys, time, torch, yaml, numpy as np
sys.path.insert(0, os.getcwd())
from dial_mpc_amd.core.dial_core import MBDPI, load_dial_and_env
from dial_mpc_amd.utils.io_utils import get_example_path
ex = sys.argv[1]
cfgd = yaml.safe_load(open(get_example_path(ex + ".yaml")))
if len(sys.argv) > 3: cfgd["Nsample"], cfgd["Hsample"] = int(sys.argv[2]), int(sys.argv[3])
dc, _, env = load_dial_and_env(cfgd)
pl = MBDPI(dc, env, kernel_rng=True)
state = env.reset(0)
Y = torch.zeros((dc.Hnode + 1, pl.nu), dtype=torch.float32, device=pl.device)
sig = pl.sigma_control.clone()
T = {k: [] for k in ("env.step", "shift", "rev0", "rev1", "total")}
sync = torch.cuda.synchronize
for tick in range(25):
    sync(); t0 = time.perf_counter()
    state = env.step(state, Y[0]); sync(); t1 = time.perf_counter()
    Y = pl.shift(Y); sync(); t2 = time.perf_counter()
    ts = [t2]
    for i in range(dc.Ndiffuse):
        _, Y, _ = pl.reverse_once(state, None, Y, sig * dc.traj_diffuse_factor ** i, eps=None, want_bars=(i == dc.Ndiffuse - 1)); sync(); ts.append(time.perf_counter())
    if tick > 2:
        T["env.step"].append(t1 - t0); T["shift"].append(t2 - t1); T["rev0"].append(ts[1] - ts[0]); T["rev1"].append(ts[-1] - ts[-2]); T["total"].append(ts[-1] - t0)
print(ex, cfgd["Nsample"], cfgd["Hsample"], "Ndiffuse", dc.Ndiffuse, {k: round(1e3 * float(np.median(v)), 3) for k, v in T.items()}, "ms (median, synchronised after every part)")
