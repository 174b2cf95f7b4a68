#!/usr/bin/env python3
"""Build a MEASUREMENT variant of the HIP library for A/B runs (never the product path):
     tools/build_variant.py NAME [--rev GITREV] [--no-fast] [-- extra hipcc flags]
   -> dial_mpc_amd/csrc/ab_NAME.so, from the working tree's sources or from those of a git revision (exported to build/src_NAME),
   with the product flags plus the extra ones.  Run an A/B with DIAL_HIP_LIB=.../ab_NAME.so python bench.py ..."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dial_mpc_amd import _lib  # noqa: E402


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    name = args[0]
    rev = args[args.index("--rev") + 1] if "--rev" in args else None
    csrc = os.path.join(ROOT, "dial_mpc_amd", "csrc")
    if rev:
        dst = os.path.join(ROOT, "build", "src_" + name)
        subprocess.check_call(f"rm -rf {dst} && mkdir -p {dst} && git -C {ROOT} archive {rev} dial_mpc_amd/csrc include | tar -x -C {dst}", shell=True)
        csrc = os.path.join(dst, "dial_mpc_amd", "csrc")
    out = os.path.join(ROOT, "dial_mpc_amd", "csrc", f"ab_{name}.so")
    objdir = os.path.join(ROOT, "build", "obj_ab_" + name)
    os.makedirs(objdir, exist_ok=True)
    fast = [] if "--no-fast" in args else _lib._FAST      # --no-fast: without the product's fast-math flags (and without -DDIAL_FUSED_DPP)
    flags = _lib._COMMON + fast + extra
    nfam = int(subprocess.check_output(f"grep -h 'define DIAL_N_FAMILIES' {csrc}/kernel_list.h", shell=True).split()[-1])
    units = [(os.path.join(csrc, "dial_hip.hip"), [], os.path.join(objdir, "dial_hip.o"))]
    units += [(os.path.join(csrc, "kern_family.hip"), [f"-DDIAL_FAMILY={k}"] + _lib._FAMILY_FLAGS.get(k, []), os.path.join(objdir, f"kern_family_{k}.o")) for k in range(nfam)]

    def cc(u):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + u[1] + ["-c", "-o", u[2], u[0]])
        return u[2]
    with ThreadPoolExecutor(max_workers=8) as pool:
        objs = list(pool.map(cc, units))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main()
