"""Example list with the reference's names (dial_mpc/examples/__init__.py:1-15).  Examples whose env
is a NEXT row are listed for discoverability but raise NotImplementedError when run."""
examples = [
    "unitree_h1_jog",
    "unitree_h1_push_crate",
    "unitree_h1_loco",
    "unitree_go2_trot",
    "unitree_go2_seq_jump",
    "unitree_go2_crate_climb",
    "allegro_reorient",
]

deploy_examples = [
    "unitree_go2_trot_deploy",
    "unitree_go2_seq_jump_deploy",
    "unitree_h1_loco_deploy",
]
