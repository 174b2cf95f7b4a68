"""Environment base configuration -- the reference's ``BaseEnvConfig``
(dial_mpc/config/base_env_config.py:4-20)."""
from dataclasses import dataclass


@dataclass
class BaseEnvConfig:
    task_name: str = "default"
    randomize_tasks: bool = False  # Whether to randomize the task.
    kp: float = 30.0  # P gain, or a list of P gains for each joint.
    kd: float = 1.0  # D gain, or a list of D gains for each joint.
    debug: bool = False
    dt: float = 0.02  # dt of the environment step, not the underlying simulator step.
    timestep: float = 0.02  # timestep of the underlying simulator step.
    backend: str = "hip"  # the reference says "mjx"; any value is accepted, the HIP kernels always run.
    leg_control: str = "torque"  # "torque" or "position"
    action_scale: float = 1.0  # scale of the action space.
