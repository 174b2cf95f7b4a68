"""Env / config registry with the reference's names (dial_mpc/envs/__init__.py:14-30 and the
``brax_envs.register_environment`` calls at unitree_go2_env.py:806-808, unitree_h1_env.py:904-906)."""
from typing import Any, Callable, Dict

from dial_mpc_amd.envs.unitree_go2_env import (
    UnitreeGo2CrateEnv, UnitreeGo2CrateEnvConfig, UnitreeGo2Env, UnitreeGo2EnvConfig, UnitreeGo2SeqJumpEnv,
    UnitreeGo2SeqJumpEnvConfig)
from dial_mpc_amd.envs.manipulation import AllegroReorientEnv, AllegroReorientEnvConfig
from dial_mpc_amd.envs.unitree_h1_env import (
    UnitreeH1LocoEnv, UnitreeH1LocoEnvConfig, UnitreeH1PushCrateEnv, UnitreeH1PushCrateEnvConfig, UnitreeH1WalkEnv,
    UnitreeH1WalkEnvConfig)

_configs: Dict[str, Any] = {
    "unitree_h1_walk": UnitreeH1WalkEnvConfig,
    "unitree_h1_loco": UnitreeH1LocoEnvConfig,
    "unitree_h1_push_crate": UnitreeH1PushCrateEnvConfig,
    "unitree_go2_walk": UnitreeGo2EnvConfig,
    "unitree_go2_seq_jump": UnitreeGo2SeqJumpEnvConfig,
    "unitree_go2_crate_climb": UnitreeGo2CrateEnvConfig,
    "allegro_reorient": AllegroReorientEnvConfig,
}
_envs: Dict[str, Callable] = {
    "unitree_h1_walk": UnitreeH1WalkEnv,
    "unitree_h1_loco": UnitreeH1LocoEnv,      # unitree_h1_env.py:906
    "unitree_h1_push_crate": UnitreeH1PushCrateEnv,   # unitree_h1_env.py:905 (generic kernel instantiation)
    "unitree_go2_walk": UnitreeGo2Env,
    "unitree_go2_seq_jump": UnitreeGo2SeqJumpEnv,
    "unitree_go2_crate_climb": UnitreeGo2CrateEnv,   # unitree_go2_env.py:808 (generic kernel instantiation)
    "allegro_reorient": AllegroReorientEnv,
}
# reference envs that are not built (none: every env of the reference registry has a kernel)
_NOT_BUILT = ()


def register_config(name: str, config: Any):
    _configs[name] = config


def get_config(name: str) -> Any:
    if name not in _configs and name in _NOT_BUILT:
        raise NotImplementedError(f"env {name!r} exists in the reference but has no HIP kernel yet")
    return _configs[name]


def register_environment(name: str, env_class: Callable):
    """brax_envs.register_environment equivalent."""
    _envs[name] = env_class


def get_environment(env_name: str, **kwargs):
    """brax_envs.get_environment equivalent (dial_core.py:221).  User-defined JAX envs cannot run on
    the HIP path: unknown names raise instead of silently falling back to anything on the CPU."""
    if env_name not in _envs:
        if env_name in _NOT_BUILT:
            raise NotImplementedError(f"env {env_name!r} exists in the reference but has no HIP kernel yet")
        raise KeyError(f"unknown environment {env_name!r}; registered: {sorted(_envs)}")
    return _envs[env_name](**kwargs)
