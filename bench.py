#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DIAL-MPC inner loop on MI355X.

Workload (BASELINE.json `metric`): unitree_go2_trot, Nsample = 2048 per GPU, Hsample = 16, Hnode = 4.
One "step" = one annealing iteration `MBDPI.reverse_once` (K1..K4: sample -> spline -> (N+1) x 17
env.steps -> softmax -> weighted means) over synthetic Go2 states that are already resident in HBM.  The
noise is drawn INSIDE the timed region, like the reference's `jax.random.normal` inside reverse_once: by the
in-kernel Philox generator of the rollout prologue (`kernel_rng=True`, the production setting).
`value` = sample-rollouts/s over all ranks ((N_total + 1) rollouts per step; one rollout = Hsample+1
env.steps).  `--scaling weak` (default): every GPU rolls out `--nsample-per-gpu` samples, N_total grows with
the GPU count.  `--scaling strong --nsample-total 65536`: BASELINE config 5 -- a fixed global sample count
sharded over the ranks (8192 per GPU on 8 GPUs), the configuration on which the >= 6x strong-scaling target
of `north_star` can be shown (the headline N = 2048 runs at the single-wavefront latency floor once it is
split 8 ways).

Extra objects on the JSON line:
  roofline      dominant kernel = rollout_kernel; achieved = ALGORITHMIC HBM bytes (356 B per
                Go2 env.step = us in + reward,q,qd,x.pos out) x env.steps per launch / average launch
                duration from hipEvents on the launch stream; peak 8000 GB/s.  The path is VALU/latency
                bound, so the estimated fp32-VALU fraction is reported next to it.
  cpu_baseline  the C oracle ("port": a CPU restatement, NOT the JAX reference -- JAX/Brax/MJX are not
                installable) timed on the host cores of this box on a bounded sample of the same workload.
  plan_latency_ms  p50/p95 of one control tick (env.step + shift + Ndiffuse x reverse_once) vs the 20 ms tick.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GO2_BYTES_PER_STEP = 356          # SURVEY 8d / BASELINE.md: 48 B in + 308 B out per env.step
# FLOP per env.step of the DENSE formulation the reference runs, COUNTED with the operation-counting build of the oracle
# (tools/opcount/count_flops.py -> profiles/r02_opcount.json; SURVEY 8d had estimated 6e4 for Go2)
FLOP_PER_STEP_FALLBACK = {"unitree_go2_trot": 62794.0, "unitree_go2_seq_jump": 60267.0, "unitree_h1_jog": 78884.0,
                          "unitree_h1_loco": 65119.0, "allegro_reorient": 755832.0}
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md
VALU_PEAK_TFLOPS = 157.3
SHADER_CLOCK_HZ = 2.4e9           # nominal; the chip clocks to its power budget (MI355X_MICROARCH.md: DVFS), issue fractions are at nominal
# cycles per wave64 VALU instruction with 1 / 2 / 3 / 4 wavefronts on a SIMD.  Preferred: the launch's OWN price --
# profiles/r06_issue_price_<example>[_N<n>].json (tools/isa/price_mix.py: every VALU instruction of the shipped kernel's disassembly
# priced by its operand form with the rows of profiles/r06_ubench_issue.txt, the hardware's dynamic classes weighted by the committed
# PMC pass).  Fallback: the hand-composed "rollout-kernel mix" row of the microbenchmark.
UBENCH_MIX_CYCLES = [5.00, 3.75, 3.54, 3.36]


def issue_price(example, n):
    """-> (cycles per VALU instruction at W = 1..4, source string)"""
    for name in (f"r06_issue_price_{example}_N{n}.json", f"r06_issue_price_{example}.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            d = json.load(open(p))
            cyc = d.get("cycles_per_valu_inst_pmc_weighted_W1_W4") or d.get("cycles_per_valu_inst_static_W1_W4")
            if cyc:
                kind = "PMC-weighted" if d.get("cycles_per_valu_inst_pmc_weighted_W1_W4") else "static"
                return cyc, f"profiles/{name} ({kind} per-form price of the shipped kernel, tools/isa/price_mix.py x profiles/r06_ubench_issue.txt)"
    return UBENCH_MIX_CYCLES, "profiles/r06_ubench_issue.txt (hand-composed mix row)"


def resident_waves_per_simd(example, rollouts, n_simd):
    """Wavefronts that share a SIMD while a launch of `rollouts` rollouts runs (what the issue ceiling of profiles/r05_ubench_issue.txt
    is looked up at): one wavefront per rollout -- the Go2 beyond 2304 rollouts: per PAIR of rollouts, at most two per SIMD (256 VGPRs,
    16 rollouts of LDS per CU) -- up to the kernels' occupancy of three to four."""
    pair = example in ("unitree_go2_trot", "unitree_go2_seq_jump") and rollouts > 2304
    waves = (rollouts + 1) // 2 if pair else rollouts
    # (the one-sample kernels keep nine wavefronts per CU resident -- LDS: 2304 rollout slots on 1024 SIMDs -- whatever the batch; round 5
    #  priced the Allegro's queue launch at four per SIMD)
    return float(min(2.0 if pair else 2.25, max(1.0, waves / n_simd)))


def cpu_baseline(example: str, N: int, H: int, budget_s: float = 16.0, min_wall_s: float = 2.0):
    """Time the CPU oracle (kind = "port") on this box's host cores.  Test-infrastructure code is used
    here ONLY as the reported baseline, never on the timed GPU path."""
    cores = len(os.sched_getaffinity(0))
    # The container's CPU BUDGET, not its affinity mask, is what the port can use: the GPU boxes of this pool expose 256 hardware
    # threads with a cgroup quota of 16 CPUs (profiles/r04_cpu_scaling.txt: perfect per-thread pace up to 32 threads, 2.9 x slower
    # per thread at 64, 134 x at 256 -- round 3's "parallel efficiency" problem was threads fighting over the quota).
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        pass
    budget = cores if quota is None else max(1, min(cores, int(round(2 * quota))))   # (up to 2 x: the quota is enforced per period)
    os.environ.setdefault("OMP_NUM_THREADS", str(budget))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O
    from conftest import seeded_inputs, setup_case
    dc, env, model, task, cfg = setup_case(example, N, H)
    try:
        o32, build_note = O.Oracle(model, task, cfg, np.float32, native=True), "gcc -O3 -march=native, built on this host"
    except Exception:                                           # no compiler on the box: the portable -O2 build
        o32, build_note = O.Oracle(model, task, cfg, np.float32), "gcc -O2 (portable build)"
    s0, _, _ = o32.env_reset(env._init_q, np.zeros(model.nv))
    eps, sigma, Ybar = seeded_inputs(dc, model.nu, seed=0)
    o32.reverse_once(s0, Ybar, sigma, eps)                      # warm-up (thread pool, page faults)
    # how many OpenMP threads serve this batch best?  (2049 rollouts are a small job for 256 hardware threads, and a
    # container's CPU quota can be far below its affinity mask: all of them spinning at barriers was 30x slower than one
    # thread's pace times the thread count.)  Short scan, then the bounded sample at the best count; `cores` = that count.
    import ctypes
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        best_nt, best_t = cores, None
        for nt in sorted({budget, max(1, budget // 2), max(1, budget // 4), min(budget, 32), min(budget, 16), min(budget, 8)}, reverse=True):
            gomp.omp_set_num_threads(nt)
            o32.reverse_once(s0, Ybar, sigma, eps)
            ta = time.perf_counter()
            o32.reverse_once(s0, Ybar, sigma, eps)
            tb = time.perf_counter() - ta
            if best_t is None or tb < best_t:
                best_nt, best_t = nt, tb
        gomp.omp_set_num_threads(best_nt)
        host_threads, cores = cores, best_nt
    except Exception:
        gomp, host_threads = None, cores
    reps, t0 = 0, time.perf_counter()
    while True:
        r = o32.reverse_once(s0, Ybar, sigma, eps)
        Ybar = r["Ybar"]
        reps += 1
        dt = time.perf_counter() - t0
        if (dt * cores >= budget_s and dt >= min_wall_s and reps >= 3) or dt >= 30.0:
            break
    n_frames = int(task.n_frames)
    ns_per_step = dt * cores / ((N + 1) * reps * (H + 1)) * 1e9
    # the same code on ONE thread (64 rollouts): what a core does when it is not waiting for the others
    single_us = None
    try:
        gomp.omp_set_num_threads(1)
        us1 = np.zeros((64, H + 1, model.nu), np.float32)
        o32.rollout(s0, us1)
        t1 = time.perf_counter()
        o32.rollout(s0, us1)
        single_us = (time.perf_counter() - t1) / (64 * (H + 1)) * 1e6
        gomp.omp_set_num_threads(cores)
    except Exception:
        pass
    return {"value": (N + 1) * reps / dt, "unit": "sample-rollouts/s", "cores": cores, "kind": "port",
            "ns_per_env_step_per_thread": ns_per_step, "single_thread_us_per_env_step": single_us, "host_threads_available": host_threads,
            "cgroup_cpu_quota": quota,
            "physics_steps_per_env_step": n_frames, "build": build_note,
            "sample": f"{reps} x reverse_once(N={N}, H={H}) fp32 C oracle ({build_note}), OpenMP over samples on {cores} "
                      f"threads (best of a scan up to the {host_threads} available), {dt:.2f} s wall = {dt * cores:.0f} core-s, {ns_per_step / 1e3:.0f} us per env.step per thread; "
                      f"a CPU restatement, not the JAX reference (not installable) -- a reported baseline, no quality claim"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nsample-per-gpu", type=int, default=2048)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--nsample-total", type=int, default=65536, help="global sample count of --scaling strong (BASELINE config 5)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="1 GPU only: run the sharded code path (1-rank RCCL group, all-gather + all-reduce) to measure its overhead")
    ap.add_argument("--host-noise", action="store_true", help="feed pre-generated eps from HBM instead of the in-kernel RNG")
    ap.add_argument("--hsample", type=int, default=None, help="default: the example's own Hsample (Go2 trot: 16)")
    ap.add_argument("--example", default="unitree_go2_trot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-cfg5", action="store_true", help="skip the N_total=65536 strong-scaling companion measurement")
    ap.add_argument("--ticks", type=int, default=200, help="control ticks for the plan-latency measurement")
    ap.add_argument("--full-only", action="store_true",
                    help="profiling runs: every launch is the FULL iteration (no lean / plan-pattern loops, plan ticks with want_bars on "
                         "every iteration), so that per-dispatch averages of rocprofv3 are those of the headline launch")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=INT",
                    help="a field of dial_options (include/dial_mpc.h), e.g. --option no_relay=1; the library reads no environment variables")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON record: whatever libraries print while the run is in progress (RCCL
    # prints a version banner to stdout when a process group is created) is routed to stderr at the file-descriptor level
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import yaml

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernels are the only compute path")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL over xGMI) by
        # re-executing this script under torch.distributed.run; rank 0's JSON line is this process's stdout
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: needs {args.gpus} visible GPUs, this box has {have}")
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.dup2(stdout_fd, 1)
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    options = {k: int(v) for k, v in (o.split("=", 1) for o in args.option)}

    from dial_mpc_amd.core.dial_core import MBDPI, load_dial_and_env
    from dial_mpc_amd.utils.io_utils import get_example_path

    cfgd = yaml.safe_load(open(get_example_path(args.example + ".yaml")))
    N_total = args.nsample_per_gpu * world if args.scaling == "weak" else args.nsample_total
    if args.scaling == "strong":
        args.nsample_per_gpu = (N_total + world - 1) // world
    if args.hsample is None:
        args.hsample = int(cfgd["Hsample"])
    cfgd["Nsample"], cfgd["Hsample"] = N_total, args.hsample
    dial_config, env_config, env = load_dial_and_env(cfgd)
    mbdpi = MBDPI(dial_config, env, kernel_rng=not args.host_noise, force_sharded=args.force_sharded, options=options)
    dev = mbdpi.device
    T, Hn1, nu = dial_config.Hsample + 1, dial_config.Hnode + 1, mbdpi.nu

    # ---- synthetic inputs, resident in HBM before the timed region
    # BASELINE.md section 2 / SURVEY 8d: 256 perturbed robot states, cycled through (sts[i % 256]); made by ONE batched env.reset
    # launch (dial_env_reset_batch), state 0 = the home key frame
    from dial_mpc_amd.utils.synthetic import perturbed_state
    N_STATES = 256
    qs, qds = [np.array(env._init_q, dtype=np.float64)], [np.zeros(env.sys.nv)]
    for seed in range(N_STATES - 1):
        q, qd = perturbed_state(env, seed)
        qs.append(q)
        qds.append(qd)
    states_t = mbdpi.ctx.env_reset_batch(torch.as_tensor(np.stack(qs), dtype=torch.float32, device=dev).contiguous(),
                                         torch.as_tensor(np.stack(qds), dtype=torch.float32, device=dev).contiguous())
    states = [states_t[i] for i in range(N_STATES)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)                              # same stream on every rank: eps is the global array
    eps_pool = [torch.randn((N_total, Hn1, nu), generator=gen, device=dev, dtype=torch.float32) for _ in range(4)] \
        if args.host_noise else [None] * 4
    sigma = mbdpi.sigma_control.clone()

    def timed_iterations(pl, sts, steps, warmup, host_eps=None, pattern=(True,)):
        """`warmup` untimed + exactly `steps` timed reverse_once calls of planner `pl`, bracketed by a barrier and a device
        synchronize on both sides; returns (max over ranks of the elapsed seconds, kernel ms total, launches).
        pattern: the want_bars flag of call i is pattern[i % len(pattern)] -- (True,) = the FULL iteration (every output of
        the reference's reverse_once: the headline), (False,) = the lean iteration (mean action only, what a plan asks of
        every annealing iteration but its last), (False, ..., True) = a plan's own sequence."""
        Yl = torch.zeros((pl.args.Hnode + 1, pl.nu), dtype=torch.float32, device=dev)

        def one_step(i, Yl):
            eps = host_eps[i % len(host_eps)] if host_eps is not None else None   # None: Philox noise inside the rollout kernel
            _, Yl, _ = pl.reverse_once(sts[i % len(sts)], None, Yl, pl.sigma_control, eps=eps, want_bars=pattern[i % len(pattern)])
            return Yl

        for i in range(warmup):
            Yl = one_step(i, Yl)
        pl.ctx.set_timing(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            Yl = one_step(i, Yl)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        k_ms, n_launch = pl.ctx.rollout_ms()
        pl.ctx.set_timing(False)
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item()), k_ms, n_launch

    # ORDER of the legs (round 6): lean loop, plan-pattern loop, plan-latency ticks, THEN the headline's W + K full iterations.  A planner
    # runs at 50 Hz without a pause, so the steady power state is the one to quote; as the first GPU work of the process, W = 5 warm-up
    # iterations (2 ms) before K = 20 timed ones measured the clock ramp with it (rounds 1-5: the driver's 20-step line read 4 % below the
    # builder's 300-step line of the same command, every round).  The timed region itself is unchanged: W untimed, exactly K timed.
    # the same K steps as LEAN iterations and in the sequence a plan runs them (Ndiffuse - 1 lean + 1 full, dial_core.py:257
    # here / :262-264 upstream); at world > 1 a lean iteration is ONE collective (all-gather of the rewards), a full one two
    # (+ all-reduce of the packed partial sums).  `value` stays the full iteration; these are reported next to it.
    plan_pat = tuple([False] * (dial_config.Ndiffuse - 1) + [True])
    # (the SAME iteration indices as the full run -- same warm-up, same count, each run restarts from a zero plan: where the rollouts'
    #  length depends on the iterate (the Allegro's solver runs to convergence, and the first iterations of an annealing run from a zero
    #  plan are its longest) a run over fewer iterations averages over a different stretch: round 5 reported the Allegro example's lean
    #  iteration SLOWER than its full one, 6.80 vs 6.57 ms, with 15 + 3 iterations against 30 + 3: profiles/r06_allegro_iteration_times.txt)
    lean_steps = args.steps
    lean_warm = args.warmup
    lean_pat, plan_run = ((True,), (True,)) if args.full_only else ((False,), plan_pat)
    el_lean, k_lean, nl_lean = timed_iterations(mbdpi, states, lean_steps if not args.full_only else 1, lean_warm if not args.full_only else 0,
                                                eps_pool if args.host_noise else None, pattern=lean_pat)
    el_plan, _, _ = timed_iterations(mbdpi, states, lean_steps if not args.full_only else 1, lean_warm if not args.full_only else 0,
                                     eps_pool if args.host_noise else None, pattern=plan_run)
    if args.full_only:
        el_lean, el_plan, lean_steps = 0.0, 0.0, 1     # (not measured in this mode)
    sharded = world > 1 or args.force_sharded

    # ---- plan latency: one control tick = env.step + shift + Ndiffuse x reverse_once (dial_core.py:245-264)
    lat = []
    state = env.reset(0)
    Yp = torch.zeros((Hn1, nu), dtype=torch.float32, device=dev)
    for tick in range(args.ticks + 1):
        torch.cuda.synchronize()
        a = time.perf_counter()
        state = env.step(state, Yp[0])
        Yp = mbdpi.shift(Yp)
        for i in range(dial_config.Ndiffuse):
            _, Yp, _ = mbdpi.reverse_once(state, None, Yp, sigma * dial_config.traj_diffuse_factor ** i,
                                          eps=eps_pool[(tick + i) % len(eps_pool)], want_bars=(args.full_only or i == dial_config.Ndiffuse - 1))
        torch.cuda.synchronize()
        if tick > 0:
            lat.append((time.perf_counter() - a) * 1e3)

    # ---- the headline: W untimed + exactly K timed FULL iterations
    elapsed, kernel_ms, launches = timed_iterations(mbdpi, states, args.steps, args.warmup,
                                                    eps_pool if args.host_noise else None)
    iteration_modes = {
        "ms_per_step_full": elapsed / args.steps * 1e3, "ms_per_step_lean": el_lean / lean_steps * 1e3,
        "ms_per_step_plan_pattern": el_plan / lean_steps * 1e3, "plan_pattern": f"{dial_config.Ndiffuse - 1} lean + 1 full",
        "lean_steps_timed": lean_steps, "avg_rollout_kernel_ms_lean": k_lean / max(nl_lean, 1),
        "collectives_per_iteration": {"full": 2 if sharded else 0, "lean": 1 if sharded else 0},
        "note": "full = every output of the reference's reverse_once (Ybar, rews, qbar, qdbar, xbar; the headline `value`); lean = "
                "want_bars=False: mean action only, the rollouts do not store their per-step states -- what the drivers ask of "
                "every annealing iteration of a plan but the last"}

    # ---- BASELINE config 5 beside the headline: unitree_go2_trot, a FIXED global N = 65536 sharded over the ranks (8192 per
    # GPU on 8 GPUs) -- the configuration of north_star's ">= 6x strong scaling at 8 GPUs"; the driver's per-N values of this
    # object give that curve, whatever `scaling` the headline value uses
    def strong_companion(n_total, s_steps, label):
        cfgs = dict(cfgd)
        cfgs["Nsample"], cfgs["Hsample"] = n_total, 16
        dcs, _, envs = load_dial_and_env(cfgs)
        pls = MBDPI(dcs, envs, kernel_rng=True, force_sharded=args.force_sharded, options=options)
        s_warm = 3
        s_el, s_kms, s_nl = timed_iterations(pls, states, s_steps, s_warm)
        l_el, _, _ = timed_iterations(pls, states, s_steps, s_warm, pattern=(False,))
        p_el, _, _ = timed_iterations(pls, states, s_steps, s_warm, pattern=plan_pat)
        rec = {"workload": f"unitree_go2_trot reverse_once, N_total={n_total} (fixed), Hsample=16, Hnode=4 -- {label}",
               "scaling": "strong", "value": (n_total + 1) * s_steps / s_el, "unit": "sample-rollouts/s", "n_gpus": world,
               "nsample_per_gpu": pls.n_local, "steps": s_steps, "warmup": s_warm, "ms_per_step": s_el / s_steps * 1e3,
               "ms_per_step_full": s_el / s_steps * 1e3, "ms_per_step_lean": l_el / s_steps * 1e3,
               "ms_per_step_plan_pattern": p_el / s_steps * 1e3, "value_plan_pattern": (n_total + 1) * s_steps / p_el,
               "collectives_per_iteration": {"full": 2 if sharded else 0, "lean": 1 if sharded else 0},
               "avg_rollout_kernel_ms": s_kms / max(s_nl, 1)}
        # the VALU-issue fraction of this launch (see `roofline.valu_issue` below), when a PMC pass of exactly this batch is committed
        pj = next((os.path.join(ROOT, "profiles", f"{r}_pmc_unitree_go2_trot_N{pls.n_local}.json") for r in ("r06", "r05")
                   if os.path.exists(os.path.join(ROOT, "profiles", f"{r}_pmc_unitree_go2_trot_N{pls.n_local}.json"))), "")
        if pj and s_kms > 0:
            pc = json.load(open(pj)).get("counters", {})
            n_simd = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
            wps = resident_waves_per_simd("unitree_go2_trot", pls.n_local + 1, n_simd)
            price, price_src = issue_price("unitree_go2_trot", pls.n_local)
            cyc = float(np.interp(wps, [1, 2, 3, 4], price))
            if pc.get("SQ_INSTS_VALU"):
                simd_cycles = n_simd * (s_kms / max(s_nl, 1)) * 1e-3 * SHADER_CLOCK_HZ
                rec["valu_issue_frac"] = pc["SQ_INSTS_VALU"] * cyc / simd_cycles
                rec["valu_issue_frac_at_2_cycles"] = pc["SQ_INSTS_VALU"] * 2.0 / simd_cycles
                rec["valu_issue_source"] = f"profiles/{os.path.basename(pj)} x {cyc:.2f} cycles per VALU instruction ({price_src}, {wps:.0f} wavefronts per SIMD)"
        del pls
        return rec

    strong, strong_hl = None, None
    if not args.no_strong_cfg5 and args.example == "unitree_go2_trot" and args.scaling == "weak" and not args.host_noise:
        strong = strong_companion(65536, max(10, args.steps // 8), "BASELINE config 5")
        if world > 1:   # north_star's 1/2/4/8 curve on the HEADLINE config: N_total = 2048 fixed (world 1: the headline itself)
            strong_hl = strong_companion(2048, max(20, args.steps // 2), "the headline config, strong scaling")

    if rank != 0:
        dist.destroy_process_group()
        return

    B_global = N_total + 1
    value = B_global * args.steps / elapsed
    n_local = mbdpi.n_local + 1                     # rollouts per launch on this rank (incl. the mean trajectory)
    avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
    # ALGORITHMIC bytes per env.step: us in + reward, q, qd, x.pos out, fp32
    bytes_per_step = 4 * (nu + 1 + mbdpi.ctx.nq + mbdpi.ctx.nv + mbdpi.ctx.nx)
    assert args.example != "unitree_go2_trot" or bytes_per_step == GO2_BYTES_PER_STEP
    opc_path = os.path.join(ROOT, "profiles", "r02_opcount.json")
    flop_per_step, flop_src = FLOP_PER_STEP_FALLBACK.get(args.example), "bench.py table (tools/opcount)"
    if os.path.exists(opc_path):
        opc = json.load(open(opc_path))
        if args.example in opc:
            flop_per_step, flop_src = opc[args.example]["flop_per_env_step"], "profiles/r02_opcount.json"
    alg_bytes = bytes_per_step * n_local * T
    achieved = alg_bytes / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
    valu_tflops = flop_per_step * n_local * T / avg_kernel_s / 1e12 if (avg_kernel_s > 0 and flop_per_step) else None
    # HBM traffic / instruction counters of THIS configuration from the committed PMC passes (separate rocprofv3 --pmc runs of
    # the same command, tools/pmc_passes.sh -> tools/pmc_to_json.py); null when the batch measured there is not the one run here
    traffic, traffic_src, valu_per_step, lane_util, stall = None, None, None, None, None
    valu_issue = None
    cands = [f"r06_pmc_{args.example}_N{args.nsample_per_gpu}.json", f"r06_pmc_{args.example}.json",
             f"r05_pmc_{args.example}_N{args.nsample_per_gpu}.json", f"r05_pmc_{args.example}.json",
             f"r04_pmc_{args.example}_N{args.nsample_per_gpu}.json", f"r04_pmc_{args.example}.json", f"r03_pmc_{args.example}.json"]
    pmc_path = next((os.path.join(ROOT, "profiles", c) for c in cands if os.path.exists(os.path.join(ROOT, "profiles", c))), None)
    if pmc_path is not None and world == 1:
        pmc = json.load(open(pmc_path))
        if pmc.get("Nsample") == args.nsample_per_gpu and pmc.get("Hsample") == args.hsample:
            traffic = pmc.get("hbm_bytes_per_launch")
            traffic_src = f"profiles/{os.path.basename(pmc_path)} (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, separate passes, per full reverse_once)"
            valu_per_step = pmc.get("valu_insts_per_wave_env_step")
            lane_util = pmc.get("valu_active_lanes_per_inst")
            stall = pmc.get("wave_time_breakdown")
            # What bounds the large batches and nearly bounds the headline: VALU ISSUE.  SQ_INSTS_VALU of the committed PMC pass x the
            # measured issue cost of this kernel's instruction mix / the SIMD-cycles of the launch timed HERE.  The cost per wave64 VALU
            # instruction is NOT the 2 cycles of the fp32 peak: tools/ubench/issue.hip (profiles/r05_ubench_issue.txt) measures 3.9
            # cycles for v_fma_f32 with three VGPR sources, 4.2 for compares / DPP / v_readlane, 8.1 for transcendentals and
            # v_permlane16_swap, 2.25 for two-source mul / mov, and for the kernels' mix 5.0 / 3.75 / 3.54 / 3.36 cycles with 1 / 2 / 3 / 4
            # wavefronts on the SIMD.
            insts = (pmc.get("counters") or {}).get("SQ_INSTS_VALU")
            if insts and avg_kernel_s > 0:
                n_simd = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
                waves_per_simd = resident_waves_per_simd(args.example, n_local, n_simd)
                price, price_src = issue_price(args.example, args.nsample_per_gpu)
                cyc = float(np.interp(waves_per_simd, [1, 2, 3, 4], price))
                simd_cycles = n_simd * avg_kernel_s * SHADER_CLOCK_HZ
                valu_issue = {"frac": insts * cyc / simd_cycles, "frac_at_2_cycles": insts * 2.0 / simd_cycles,
                              "assumption": "frac: every VALU instruction at the MEASURED issue cost of its operand form at this occupancy; "
                                            "frac_at_2_cycles: the guide's one wave64 instruction per 2 cycles (MI355X_MICROARCH.md), which only "
                                            "two-VGPR-source VOP1/VOP2 forms reach on this chip (profiles/r06_ubench_issue.txt)",
                              "valu_insts_per_launch_pmc": insts, "valu_mix_pmc": pmc.get("valu_mix"),
                              "cycles_per_valu_inst": cyc, "cycles_per_valu_inst_W1_W4": price, "wavefronts_per_simd": waves_per_simd, "simds": n_simd,
                              "clock_hz": SHADER_CLOCK_HZ,
                              "source": f"profiles/{os.path.basename(pmc_path)} (SQ_INSTS_VALU) x {price_src} "
                                        f"/ (SIMDs x this run's average kernel time x nominal clock)"}
    out = {
        "metric": "sample-rollouts/sec (N x H env.steps), Go2 N=2048 H=16" if args.example == "unitree_go2_trot"
        else f"sample-rollouts/sec (N x H env.steps), {args.example}", "value": value,
        "unit": "sample-rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.example} reverse_once: Nsample={args.nsample_per_gpu}/GPU "
                               f"(N_total={N_total}), Hsample={args.hsample}, Hnode={dial_config.Hnode}, "
                               f"{N_STATES} synthetic robot states (home + {N_STATES - 1} perturbed, cycled); noise: " +
                               ("eps ~ N(0,1) pre-generated, resident in HBM" if args.host_noise else
                                "Philox4x32-10 + Box-Muller inside the rollout kernel, i.e. inside the timed region"),
                   "env_steps_per_s": value * T, "parallelism": f"samples sharded over {world} rank(s)" +
                   (" (sharded code path forced: 1-rank RCCL all-gather + all-reduce per iteration)" if args.force_sharded else "")},
        # `achieved / peak / frac / traffic` are the HBM figures the bench contract defines (algorithmic bytes per launch /
        # kernel time vs 8 TB/s).  They are NOT what bounds this kernel: `bound` names that -- dependent issue latency (one
        # sample = one dependence chain of ~4.9 k VALU + 0.7 k LDS instructions per env.step; PMC: a wavefront issues half
        # of its time, is parked at s_waitcnt a third and stalled at issue a tenth) -- and the counted-FLOP and PMC figures
        # next to it quantify it.
        "roofline": {"bound": "valu-latency",
                     "bound_note": "neither hbm nor mfma: 176 counted FLOP per algorithmic byte vs a machine balance of 20 FLOP/B; "
                                   "achieved/peak/frac/traffic below are the HBM figures of the bench contract",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "rollout_kernel",
                     "avg_kernel_ms": avg_kernel_s * 1e3, "launches": launches,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "flop_per_env_step_counted": flop_per_step, "flop_source": flop_src,
                     "valu_tflops_counted": valu_tflops,
                     "valu_frac_counted": (valu_tflops / VALU_PEAK_TFLOPS) if valu_tflops is not None else None,
                     "valu_insts_per_wave_env_step_pmc": valu_per_step, "valu_active_lanes_per_inst_pmc": lane_util,
                     "wave_time_breakdown_pmc": stall,
                     "valu_issue_frac": valu_issue["frac"] if valu_issue else None, "valu_issue": valu_issue},
        "plan_latency_ms": {"p50": float(np.percentile(lat, 50)), "p95": float(np.percentile(lat, 95)),
                            "ticks": len(lat), "tick_budget_ms": 20.0,
                            "plan": f"env.step + shift + {dial_config.Ndiffuse} x reverse_once"},
    }
    out["iteration_modes"] = iteration_modes
    if strong is not None:
        out["strong_cfg5"] = strong
    if strong_hl is not None:
        out["strong_headline_n2048"] = strong_hl
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args.example, args.nsample_per_gpu, args.hsample)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    if world > 1 or args.force_sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
