"""Test double of the plant process (dial_sim / dial_real): owns the six shm segments of the async-planner protocol and
advances the same HIP env with the plan's first node -- enough to exercise ``MBDPublisher`` end to end.  Test
infrastructure; the product module is dial_mpc_amd/deploy/dial_plan.py."""
import numpy as np

from dial_mpc_amd.deploy.dial_plan import open_segments


class FakePlant:
    """Stand-in for dial_sim / dial_real."""

    def __init__(self, env, dial_config, shm_prefix: str = ""):
        self.env = env
        mj = env.sys.mj_model
        self.nq, self.nv, self.nu = mj.nq, mj.nv, mj.nu
        self.n_acts = dial_config.Hsample + 1
        self.ctrl_dt = env._config.dt
        self._seg = open_segments(self.nq, self.nv, self.nu, self.n_acts, create=True, prefix=shm_prefix)
        for _, arr in self._seg.values():
            arr[...] = 0.0
        self._seg["plan_time_shm"][1][0] = -self.ctrl_dt
        self.t = 0.0
        self.state = env.reset(0)
        self.publish()

    def publish(self):
        ps = self.state.pipeline_state
        self._seg["time_shm"][1][0] = self.t
        self._seg["state_shm"][1][:] = np.concatenate([ps.qpos.cpu().numpy(), ps.qvel.cpu().numpy()])

    def step_with_action(self, action):
        self.state = self.env.step(self.state, action)
        self.t += self.ctrl_dt
        self.publish()

    def close(self):
        for shm, _ in self._seg.values():
            shm.close()
            try:
                shm.unlink()
            except FileNotFoundError:
                pass
