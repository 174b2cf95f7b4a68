import sys, numpy as np, torch, yaml
sys.path.insert(0, '/root/repo')
from dial_mpc_amd.core.dial_core import MBDPI, load_dial_and_env
from dial_mpc_amd.utils.io_utils import get_example_path
tag = sys.argv[1]
for N in (512,):
  for seed in range(int(sys.argv[2]), int(sys.argv[3])):
    cfgd = yaml.safe_load(open(get_example_path("allegro_reorient.yaml")))
    cfgd["Nsample"] = N; cfgd["seed"] = seed
    dial_config, env_config, env = load_dial_and_env(cfgd)
    mbdpi = MBDPI(dial_config, env, kernel_rng=True)
    state = env.reset(0)
    Y = torch.zeros((dial_config.Hnode + 1, mbdpi.nu), device=mbdpi.device)
    zs = []
    for t in range(60):
        state = env.step(state, Y[0])
        Y = mbdpi.shift(Y)
        n_it = dial_config.Ndiffuse_init if t == 0 else dial_config.Ndiffuse
        for i in range(n_it):
            _, Y, info = mbdpi.reverse_once(state, None, Y, mbdpi.sigma_control * dial_config.traj_diffuse_factor ** i, want_bars=(i == n_it - 1))
        zs.append(float(state.pipeline_state.q[2]))
    print(tag, "N", N, "seed", seed, "ball z at ticks 20/40/60:", [round(zs[k], 3) for k in (19, 39, 59)], "DROPPED" if min(zs) < 0.0 else "", flush=True)
