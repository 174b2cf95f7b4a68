#!/usr/bin/env python3
"""Do two builds of the library produce the same BITS?  Runs one reverse_once (in-kernel noise off: seeded eps) per build in a child
process (DIAL_HIP_LIB=<lib>) and compares the rollouts' outputs (rews, qss, qdss, xss, Ybar) word for word.
usage: bit_compare.py <libA.so> <libB.so> [example] [N] [H]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
from conftest import seeded_inputs, setup_case
from dial_mpc_amd import _lib
example, N, H, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
dc, env, model, task, cfg = setup_case(example, N, H)
ctx = _lib.Context(model, task, cfg)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device='cuda')
s0, _, _ = ctx.env_reset(dev(env._init_q), dev(np.zeros(model.nv)))
eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=0)
o = ctx.reverse_once(s0, dev(Ybar), dev(sigma), dev(eps))
sc = ctx.debug_scratch()
np.savez(out, Ybar=o['Ybar'].cpu().numpy(), rews=o['rews'].cpu().numpy(), qss=sc['qss'], qdss=sc['qdss'], xss=sc['xss'], rewss=sc['rewss'])
""" % (ROOT, ROOT)


def run(lib, example, N, H, out):
    env = dict(os.environ, DIAL_HIP_LIB=os.path.abspath(lib))
    subprocess.check_call([sys.executable, "-c", CHILD, example, str(N), str(H), out], env=env)
    return np.load(out)


def main():
    a, b = sys.argv[1], sys.argv[2]
    example = sys.argv[3] if len(sys.argv) > 3 else "unitree_go2_trot"
    N, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (2048, 16)
    with tempfile.TemporaryDirectory() as td:
        ra, rb = run(a, example, N, H, os.path.join(td, "a.npz")), run(b, example, N, H, os.path.join(td, "b.npz"))
        for k in ("rews", "rewss", "qss", "qdss", "xss", "Ybar"):
            x, y = ra[k], rb[k]
            same = np.array_equal(x.view(np.uint32), y.view(np.uint32))
            nd = int((x.view(np.uint32) != y.view(np.uint32)).sum())
            print(f"{example} N={N} H={H} {k:6s}: {'BIT-IDENTICAL' if same else f'{nd} of {x.size} words differ, max |diff| {np.abs(x - y).max():.3e}'}")


if __name__ == "__main__":
    main()
