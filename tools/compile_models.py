#!/usr/bin/env python3
"""Compile the reference's MJCF robot descriptions into the JSON constant blobs shipped under
dial_mpc_amd/models/ (the XML files themselves are NOT copied into this repository).

Usage (in the build container, where /root/reference is mounted):
    python tools/compile_models.py [--reference /root/reference]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dial_mpc_amd import mjcf  # noqa: E402

MODELS = [
    ("unitree_go2", "mjx_scene_force.xml"),
    ("unitree_go2", "mjx_scene_force_crate.xml"),
    ("unitree_h1", "mjx_scene_h1_walk.xml"),
    ("unitree_h1", "mjx_scene_h1_loco.xml"),
    ("unitree_h1", "mjx_scene_h1_push_crate.xml"),
    ("wonik_allegro", "scene_left.xml"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    out_root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dial_mpc_amd", "models")
    for robot, xml in MODELS:
        src = os.path.join(args.reference, "dial_mpc", "models", robot, xml)
        m = mjcf.compile_mjcf(src)
        m["source"] = f"dial_mpc/models/{robot}/{xml}"
        os.makedirs(os.path.join(out_root, robot), exist_ok=True)
        dst = os.path.join(out_root, robot, os.path.splitext(xml)[0] + ".json")
        with open(dst, "w") as f:
            f.write(mjcf.model_to_json(m))
        print(f"{src} -> {dst}  (nq={m['nq']} nv={m['nv']} nu={m['nu']} nbody={m['nbody']} "
              f"ncon={m['ncon']} nefc={m['nefc']})")


if __name__ == "__main__":
    main()
