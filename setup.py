"""Packaging with the reference's console-script names (setup.py:23-32 upstream): `dial-mpc` (sync driver) and
`dial-mpc-plan` (async planner process).  The HIP library is built in-tree by `__graft_entry__.build()`."""
from setuptools import find_packages, setup

setup(
    name="dial-mpc-amd",
    version="0.1.0",
    description="MI355X-native DIAL-MPC inner loop (rollout + reward + softmax update) behind the dial-mpc Python surface",
    packages=find_packages(include=["dial_mpc_amd", "dial_mpc_amd.*"]),
    package_data={"dial_mpc_amd": ["models/*/*.json", "examples/*.yaml", "csrc/*", "../include/*.h"]},
    install_requires=["numpy", "pyyaml", "torch"],
    entry_points={"console_scripts": ["dial-mpc=dial_mpc_amd.core.dial_core:main",
                                      "dial-mpc-plan=dial_mpc_amd.deploy.dial_plan:main"]},
)
