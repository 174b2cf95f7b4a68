#!/usr/bin/env python3
"""Why does the CPU port lose per-thread pace in parallel?  Prints the box's CPU budget (cgroup quota,
affinity, topology) and the oracle's pace -- microseconds per env.step per thread -- over a thread-count scan of
`oracle_rollout` (2048 Go2 rollouts x 17 steps, OpenMP over rollouts, no barriers inside), for the OMP settings given in
the environment.  Run it once per setting, e.g.
    python tools/cpu_scaling.py;  OMP_PROC_BIND=close OMP_PLACES=cores python tools/cpu_scaling.py"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        if os.path.exists(f):
            print(f, "=", open(f).read().strip())
    print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "loadavg", os.getloadavg(),
          "OMP_PROC_BIND", os.environ.get("OMP_PROC_BIND"), "OMP_PLACES", os.environ.get("OMP_PLACES"), "OMP_WAIT_POLICY", os.environ.get("OMP_WAIT_POLICY"))
    os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA node\\(s\\)|MHz' | head -8")
    import oracle as O
    from conftest import setup_case
    dc, env, model, task, cfg = setup_case("unitree_go2_trot", 2048, 16)
    o32 = O.Oracle(model, task, cfg, np.float32, native=True)
    s0, _, _ = o32.env_reset(env._init_q, np.zeros(model.nv))
    us = np.random.default_rng(0).uniform(-0.3, 0.3, (2048, 17, model.nu)).astype(np.float32)
    gomp = ctypes.CDLL("libgomp.so.1")
    cores = len(os.sched_getaffinity(0))
    o32.rollout(s0, us)
    base = None
    for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if nt > cores:
            break
        gomp.omp_set_num_threads(nt)
        o32.rollout(s0, us[: max(64, 8 * nt)])
        n = 2048 if nt >= 8 else 256 * nt
        t0 = time.perf_counter()
        o32.rollout(s0, us[:n])
        dt = time.perf_counter() - t0
        pace = dt * nt / (n * 17) * 1e6
        base = base or pace
        print(f"threads {nt:4d}: {n * 17 / dt / 1e6:8.3f} M env.steps/s, {pace:7.2f} us per env.step per thread ({pace / base:5.2f} x single)")


if __name__ == "__main__":
    main()
