"""Example list with the reference's names (dial_mpc/examples/__init__.py:1-15), restricted to the examples whose YAML
ships here and whose env runs on the HIP path -- all seven of the reference's envs."""
examples = [
    "unitree_h1_jog",
    "unitree_h1_loco",
    "unitree_go2_trot",
    "unitree_go2_seq_jump",
    "unitree_go2_crate_climb",
    "unitree_h1_push_crate",
    "allegro_reorient",
]

deploy_examples = [
    "unitree_go2_trot_deploy",
    "unitree_go2_seq_jump_deploy",
    "unitree_h1_loco_deploy",
]
