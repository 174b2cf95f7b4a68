"""Loader and thin wrappers of libdialhip.so -- the HIP product path.

There is deliberately NO fallback: if the shared library is missing, or no HIP device is present, the
calls below raise.  Tensors are PyTorch-ROCm tensors used purely as HBM allocations; the kernels get
raw device pointers and the current torch stream through the C ABI of include/dial_mpc.h.
"""
from __future__ import annotations

import ctypes
import glob
import os
import subprocess
from typing import Optional

from dial_mpc_amd import _abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("DIAL_HIP_LIB", os.path.join(_CSRC, "libdialhip.so"))  # override: profiling builds
_lib = None


class DialHipError(RuntimeError):
    pass


IEEE_LIB_PATH = os.path.join(_CSRC, "libdialhip_ieee.so")


N_FAMILIES = 8     # robot families of csrc/kernel_list.h (7: the Go2's two-samples-per-wavefront kernels), one translation unit each (kern_family.hip -DDIAL_FAMILY=k)
# device fast-math flags of the product build (the IEEE measurement variant drops them):
# -fno-hip-fp32-correctly-rounded-divide-sqrt: fp32 divide / sqrt via v_rcp / v_sqrt sequences (<= 2.5 ulp)
# instead of the IEEE fix-up chains; well inside the fp32 parity tolerance.
# device only: -freciprocal-math (a/b -> a * v_rcp(b), no frexp/ldexp range scaling) and -fapprox-func
# (native v_sqrt / v_rsq / v_exp / v_log without denormal fix-ups); finite-math is NOT assumed (+-inf
# control ranges are compared against).  -fno-honor-nans: min / max / clip become single v_min / v_max instead of
# compare + select chains.
# -DDIAL_FUSED_DPP: the pair kernel's broadcast-multiply-adds as single v_fmac_f32_dpp instructions (csrc/wave.h: WaveH::fma_pick);
# product build only -- the fused form IS a contraction.
_FAST = ["-DDIAL_FUSED_DPP", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-Xarch_device", "-freciprocal-math", "-Xarch_device", "-fapprox-func",
         "-Xarch_device", "-fno-honor-nans"]
# -fno-slp-vectorize (both variants): the SLP pass packs pairs of scalar fp32 ops into v_pk_* and pays for it in v_mov
# shuffles -- measured 5-6% slower on this issue-bound kernel.
# the IEEE measurement variant: none of the fast-math flags and NO fused multiply-add contraction either -- every fp32 operation
# rounded on its own, as the CPU oracle computes.  Which a * b + c pairs the compiler fuses depends on the basic-block structure around
# them, i.e. differs between two kernels that run the same arithmetic on different lane layouts: this variant is where the Go2's
# two-samples-per-wavefront kernel is compared BIT FOR BIT with the one-sample kernel (tests/test_gpu_parity.py).
_IEEE = ["-Xarch_device", "-ffp-contract=off"]
_COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xarch_device", "-fno-slp-vectorize"]
# per robot family (kern_family.hip -DDIAL_FAMILY=k): LLVM's "max-ilp" machine-scheduling strategy for the Allegro's kernels -- their
# launch lasts as long as ONE lone wavefront's dependence chain, which the strategy shortens: A/B on one box
# (profiles/r05_ab_sched_max_ilp.txt) Allegro example 6.67 -> 6.51 ms (-2.5 %); the Go2 (+2 %), the push crate (+2 %) and the H1 (+-0)
# keep the default strategy.
# Family 4 (the capacity-dimension kernels, DimsMax): SimplifyCFG's common-code sinking is OFF.  A wavefront whose touching contacts
# exceed the capped LDS workspace runs a second inlined copy of the constraint code on an overflow area in GLOBAL memory
# (csrc/rollout_body.h: forward_tail); the sinking pass merges instructions of the two copies into one block behind pointer PHIs that
# mix the LDS and the global address space, and instruction selection then dies with "Illegal instruction detected: V_CMP_NE_U32 0,
# $src_shared_base" -- rounds 4-5 met this as "the translation unit trips an LLVM bug whenever <some loop> changes shape" and
# kept old code shapes alive for it; round 6 bisected the pass (tools/isa/llvm_sink_bug.md).
_FAMILY_FLAGS = {3: ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], 4: ["-mllvm", "-simplifycfg-sink-common=false"]}


def build(force: bool = False, verbose: bool = False, ieee: bool = False) -> str:
    """Compile the library for gfx950 in-tree (hipcc cross-compiles without a GPU): csrc/dial_hip.hip (host side, K4 / K5
    kernels) and csrc/kern_family.hip once per robot family, the translation units IN PARALLEL, then one link.

    ieee=True builds the MEASUREMENT variant libdialhip_ieee.so: the same sources without the device fast-math flags
    (correctly rounded divide / sqrt, no reciprocal-math, no approximate functions, NaNs honoured) and without fused
    multiply-add contraction (_IEEE above).  It is never the product
    path; the GPU suite loads it next to the product library to show how much of the knife-edge witness traffic is the
    fast-math rounding (tests/test_gpu_parity.py: test_ieee_build_needs_no_more_witnesses)."""
    from concurrent.futures import ThreadPoolExecutor
    # every source of the library takes part in the staleness check (a stale .so must never ship silently)
    srcs = sorted(glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_CSRC, "*.hip"))) + [_abi.HEADER, os.path.abspath(__file__)]   # (this file: the flags)
    out = IEEE_LIB_PATH if ieee else LIB_PATH
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _COMMON + (_IEEE if ieee else _FAST) + ([] if ieee else os.environ.get("DIAL_HIPCC_EXTRA", "").split())
    # ONE builder at a time per output file (ranks of a multi-process launch, pytest-xdist workers that all find a stale tree): an
    # exclusive flock on build/<library>.lock, staleness re-checked once the lock is held -- the others find a fresh library.
    # Objects go to a directory of THIS call (flags hashed into its name, pid-suffixed), the library is linked next to its final place
    # and moved there atomically; the directory and the temporary are removed whether the build succeeds or not.
    import fcntl
    import hashlib
    import shutil
    build_root = os.path.join(os.path.dirname(os.path.dirname(_CSRC)), "build")
    os.makedirs(build_root, exist_ok=True)
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:8]
    objdir = os.path.join(build_root, f"obj_{os.path.basename(out)}_{tag}_{os.getpid()}")
    tmp_out = f"{out}.{os.getpid()}.tmp"
    with open(os.path.join(build_root, os.path.basename(out) + ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
                return out
            os.makedirs(objdir, exist_ok=True)
            units = [(os.path.join(_CSRC, "dial_hip.hip"), [], os.path.join(objdir, "dial_hip.o"))]
            units += [(os.path.join(_CSRC, "kern_family.hip"), [f"-DDIAL_FAMILY={k}"] + _FAMILY_FLAGS.get(k, []), os.path.join(objdir, f"kern_family_{k}.o"))
                      for k in range(N_FAMILIES)]

            def compile_unit(u):
                src, defs, obj = u
                cmd = [hipcc] + flags + defs + ["-c", "-o", obj, src]
                if verbose:
                    print(" ".join(cmd))
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                if r.returncode != 0:   # (the compiler's own words, not just the exit status: a failed unit must be impossible to overlook)
                    raise DialHipError(f"hipcc failed ({r.returncode}) on {os.path.basename(src)} {' '.join(defs)}:\n{r.stdout[-4000:]}")
                return obj
            with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
                objs = list(pool.map(compile_unit, units))
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_out] + objs
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp_out, out)
            return out
        finally:
            shutil.rmtree(objdir, ignore_errors=True)
            if os.path.exists(tmp_out):
                os.remove(tmp_out)
            fcntl.flock(lock, fcntl.LOCK_UN)


def load(path: Optional[str] = None):
    """The product library (cached), or -- path given -- another build of it (measurement variants; not cached)."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    lib_path = LIB_PATH if path is None else path
    if not os.path.exists(lib_path):
        raise DialHipError(f"{lib_path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is the only compute path; there is no CPU fallback)")
    lib = ctypes.CDLL(lib_path)
    vp, ci, fp = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    lib.dial_create.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ci]
    lib.dial_create_sharded.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ci, ci]
    lib.dial_create_ex.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ci, ci, vp]
    lib.dial_set_state_trace.argtypes = [vp, fp, ci]
    lib.dial_destroy.argtypes = [vp]
    lib.dial_destroy.restype = None
    lib.dial_last_error.argtypes = [vp]
    lib.dial_last_error.restype = ctypes.c_char_p
    lib.dial_rollout.argtypes = [vp, fp, fp, ci, fp, fp, fp, fp, vp]
    lib.dial_reverse_once.argtypes = [vp, fp, fp, fp, ci, fp, fp, fp, fp, fp, fp, vp]
    lib.dial_shard_rollout.argtypes = [vp, fp, fp, fp, ci, fp, ci, ci, fp, vp]
    lib.dial_shard_reduce.argtypes = [vp, fp, ci, ci, ci, ci, fp, vp]
    lib.dial_shard_ybar.argtypes = [vp, fp, ci, fp, fp, fp, ci, fp, vp]
    u64, u32 = ctypes.c_uint64, ctypes.c_uint32
    lib.dial_reverse_once_rng.argtypes = [vp, fp, fp, fp, ci, u64, u32, fp, fp, fp, fp, fp, vp]
    lib.dial_shard_rollout_rng.argtypes = [vp, fp, fp, fp, ci, u64, u32, ci, ci, ci, fp, vp]
    lib.dial_rng_fill.argtypes = [vp, u64, u32, ci, ci, fp, vp]
    lib.dial_shard_ybar_rng.argtypes = [vp, fp, ci, u64, u32, fp, fp, ci, fp, vp]
    lib.dial_shard_pack_rewards.argtypes = [vp, fp, ci, ci, ci, fp, vp]
    try:
        lib.dial_shard_ybar_gathered.argtypes = [vp, fp, ci, ci, ci, fp, fp, fp, ci, fp, fp, vp]
        lib.dial_shard_ybar_gathered_rng.argtypes = [vp, fp, ci, ci, ci, u64, u32, fp, fp, ci, fp, fp, vp]
        lib.dial_shard_reduce_gathered.argtypes = [vp, fp, ci, ci, ci, ci, ci, ci, fp, fp, vp]
    except AttributeError:
        if path is None and "DIAL_HIP_LIB" not in os.environ:   # (an A/B build of an earlier round may lack them; the product library may not)
            raise
    lib.dial_shift.argtypes = [vp, fp, vp]
    lib.dial_env_step.argtypes = [vp, fp, fp, fp, fp, fp, vp]
    lib.dial_env_reset.argtypes = [vp, fp, fp, fp, fp, fp, vp]
    lib.dial_env_reset_batch.argtypes = [vp, fp, fp, fp, fp, fp, ci, vp]
    lib.dial_status.argtypes = [vp]
    lib.dial_set_timing.argtypes = [vp, ci]
    lib.dial_get_rollout_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ci)]
    lib.dial_abi_sizes.argtypes = [ctypes.POINTER(ci)] * 3
    lib.dial_selftest.argtypes = [ctypes.POINTER(ctypes.c_float)]
    lib.dial_debug_scratch.argtypes = [vp] + [ctypes.POINTER(vp)] * 6
    lib.dial_lds_bytes.argtypes = [vp]
    lib.dial_debug_resident_rollouts.argtypes = [vp, ctypes.c_int]
    if path is None:
        _lib = lib
    return lib


EXPORTED = ("dial_create", "dial_create_sharded", "dial_create_ex", "dial_set_state_trace", "dial_destroy", "dial_last_error", "dial_rollout", "dial_reverse_once",
            "dial_shard_rollout", "dial_shard_reduce", "dial_shard_ybar", "dial_reverse_once_rng",
            "dial_shard_rollout_rng", "dial_rng_fill", "dial_shard_ybar_rng", "dial_shard_pack_rewards",
            "dial_shard_ybar_gathered", "dial_shard_ybar_gathered_rng", "dial_shard_reduce_gathered", "dial_shift", "dial_env_step", "dial_env_reset", "dial_env_reset_batch",
            "dial_status", "dial_set_timing", "dial_get_rollout_ms", "dial_abi_sizes")


def _ptr(t) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous() and str(t.dtype) == "torch.float32", "need contiguous float32 device tensors"
    return t.data_ptr()


def _stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


class Context:
    """One dial_ctx: (device, model, task, cfg).  Not thread-safe (C ABI contract)."""

    def __init__(self, model: "_abi.DialModel", task: "_abi.DialTask", cfg: Optional["_abi.DialCfg"],
                 device: Optional[int] = None, n_local_cap: Optional[int] = None, lib_path: Optional[str] = None,
                 options: Optional[dict] = None):
        """n_local_cap: size the rollout scratch for that many local samples (one rank of a sharded run).
        lib_path: another build of the library (measurement variants, e.g. libdialhip_ieee.so).
        options: fields of `dial_options` (include/dial_mpc.h) -- launch-shape / measurement switches, e.g.
        dict(no_queue=1); none of them changes a result bit.  The library itself reads no environment variables."""
        import torch
        self.lib = load(lib_path)
        if not torch.cuda.is_available():
            raise DialHipError("no HIP device visible to PyTorch: the DIAL-MPC kernels only run on a GPU")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.torch_device = torch.device("cuda", self.device)
        self.model, self.task, self.cfg = model, task, cfg
        self.nq, self.nv, self.nu, self.nbody = model.nq, model.nv, model.nu, model.nbody
        self.nx = (model.nbody - 1) * 3
        self.state_size = _abi.state_size(model.nq, model.nv)
        h = ctypes.c_void_p()
        self.options = dict(options or {})
        unknown = set(self.options) - set(_abi.DialOptions._meta)
        if unknown:
            raise ValueError(f"unknown dial_options fields: {sorted(unknown)}")
        opts = _abi.fill(_abi.DialOptions(), self.options)
        rc = self.lib.dial_create_ex(ctypes.byref(h), ctypes.addressof(model), ctypes.addressof(task),
                                     ctypes.addressof(cfg) if cfg is not None else None, self.device,
                                     -1 if (n_local_cap is None or cfg is None) else int(n_local_cap), ctypes.addressof(opts))
        if rc != 0:
            raise DialHipError(f"dial_create failed ({rc}): {self.lib.dial_last_error(None).decode()}")
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.dial_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise DialHipError(f"{what} failed ({rc}): {self.lib.dial_last_error(self.h).decode()}")

    # ---- K6
    def env_reset(self, qpos, qvel):
        import torch
        state = torch.zeros(self.state_size, dtype=torch.float32, device=self.torch_device)
        xpos = torch.zeros((self.nbody - 1, 3), dtype=torch.float32, device=self.torch_device)
        xquat = torch.zeros((self.nbody - 1, 4), dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.dial_env_reset(self.h, _ptr(qpos), _ptr(qvel), _ptr(state), _ptr(xpos), _ptr(xquat),
                                            _stream()), "dial_env_reset")
        return state, xpos, xquat

    def env_reset_batch(self, qpos, qvel):
        """n states in one launch: qpos [n, nq], qvel [n, nv] -> packed states [n, state_size]."""
        import torch
        n = int(qpos.shape[0])
        states = torch.zeros((n, self.state_size), dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.dial_env_reset_batch(self.h, _ptr(qpos), _ptr(qvel), _ptr(states), None, None, n, _stream()),
                    "dial_env_reset_batch")
        return states

    def env_step(self, state, action):
        import torch
        state = state.clone()
        xpos = torch.zeros((self.nbody - 1, 3), dtype=torch.float32, device=self.torch_device)
        xquat = torch.zeros((self.nbody - 1, 4), dtype=torch.float32, device=self.torch_device)
        ctrl = torch.zeros(self.nu, dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.dial_env_step(self.h, _ptr(state), _ptr(action), _ptr(xpos), _ptr(xquat), _ptr(ctrl),
                                           _stream()), "dial_env_step")
        return state, xpos, xquat, ctrl

    # ---- K3
    def rollout(self, state, us, want_states: bool = True):
        import torch
        B, T = us.shape[0], us.shape[1]
        assert T == self.cfg.Hsample + 1 and us.shape[2] == self.nu
        dev = self.torch_device
        rewss = torch.empty((B, T), dtype=torch.float32, device=dev)
        qss = torch.empty((B, T, self.nq), dtype=torch.float32, device=dev) if want_states else None
        qdss = torch.empty((B, T, self.nv), dtype=torch.float32, device=dev) if want_states else None
        xss = torch.empty((B, T, self.nx), dtype=torch.float32, device=dev) if want_states else None
        self._check(self.lib.dial_rollout(self.h, _ptr(state), _ptr(us), B, _ptr(rewss), _ptr(qss), _ptr(qdss),
                                          _ptr(xss), _stream()), "dial_rollout")
        return rewss, qss, qdss, xss

    # ---- K1..K4
    def _out(self, want_bars: bool):
        import torch
        cfg, dev = self.cfg, self.torch_device
        N, Hn1, T = cfg.Nsample, cfg.Hnode + 1, cfg.Hsample + 1
        f32 = dict(dtype=torch.float32, device=dev)
        return dict(Ybar=torch.empty((Hn1, self.nu), **f32), rews=torch.empty(N + 1, **f32),
                    qbar=torch.empty((T, self.nq), **f32) if want_bars else None,
                    qdbar=torch.empty((T, self.nv), **f32) if want_bars else None,
                    xbar=torch.empty((T, self.nx), **f32) if want_bars else None)

    def reverse_once(self, state, Ybar, noise_scale, eps, out=None, want_bars: bool = True):
        """want_bars=False: qbar / qdbar / xbar are None and the rollouts do not write their per-step states."""
        cfg = self.cfg
        N, Hn1 = cfg.Nsample, cfg.Hnode + 1
        assert tuple(eps.shape) == (N, Hn1, self.nu) and tuple(Ybar.shape) == (Hn1, self.nu)
        ns = int(noise_scale.numel())
        if out is None:
            out = self._out(want_bars)
        self._check(self.lib.dial_reverse_once(self.h, _ptr(state), _ptr(Ybar), _ptr(noise_scale), ns, _ptr(eps),
                                               _ptr(out["Ybar"]), _ptr(out["rews"]), _ptr(out["qbar"]),
                                               _ptr(out["qdbar"]), _ptr(out["xbar"]), _stream()), "dial_reverse_once")
        return out

    def reverse_once_rng(self, state, Ybar, noise_scale, seed: int, counter: int, out=None, want_bars: bool = True):
        """reverse_once with the noise generated inside the rollout kernel (Philox keyed by seed / counter)."""
        if out is None:
            out = self._out(want_bars)
        self._check(self.lib.dial_reverse_once_rng(self.h, _ptr(state), _ptr(Ybar), _ptr(noise_scale),
                                                   int(noise_scale.numel()), int(seed), int(counter), _ptr(out["Ybar"]),
                                                   _ptr(out["rews"]), _ptr(out["qbar"]), _ptr(out["qdbar"]),
                                                   _ptr(out["xbar"]), _stream()), "dial_reverse_once_rng")
        return out

    def rng_fill(self, seed: int, counter: int, n_begin: int, n_count: int):
        import torch
        eps = torch.empty((n_count, self.cfg.Hnode + 1, self.nu), dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.dial_rng_fill(self.h, int(seed), int(counter), n_begin, n_count, _ptr(eps), _stream()),
                    "dial_rng_fill")
        return eps

    def shard_rollout_rng(self, state, Ybar, noise_scale, seed, counter, n_begin, n_local, with_mean, rews_local):
        self._check(self.lib.dial_shard_rollout_rng(self.h, _ptr(state), _ptr(Ybar), _ptr(noise_scale),
                                                    int(noise_scale.numel()), int(seed), int(counter), n_begin, n_local,
                                                    int(with_mean), _ptr(rews_local), _stream()), "dial_shard_rollout_rng")

    def shard_rollout(self, state, Ybar, noise_scale, eps_local, n_local: int, with_mean: bool, rews_local):
        ns = int(noise_scale.numel())
        self._check(self.lib.dial_shard_rollout(self.h, _ptr(state), _ptr(Ybar), _ptr(noise_scale), ns,
                                                _ptr(eps_local) if n_local > 0 else None, n_local, int(with_mean),
                                                _ptr(rews_local), _stream()), "dial_shard_rollout")

    def shard_reduce(self, rews_all, n_total: int, n_begin: int, n_local: int, include_mean: bool, packed_out):
        self._check(self.lib.dial_shard_reduce(self.h, _ptr(rews_all), n_total, n_begin, n_local, int(include_mean),
                                               _ptr(packed_out), _stream()), "dial_shard_reduce")

    def shard_ybar(self, rews_all, n_total: int, eps_all, Ybar, noise_scale, Ybar_out):
        ns = int(noise_scale.numel())
        self._check(self.lib.dial_shard_ybar(self.h, _ptr(rews_all), n_total, _ptr(eps_all), _ptr(Ybar), _ptr(noise_scale),
                                             ns, _ptr(Ybar_out), _stream()), "dial_shard_ybar")

    def shard_ybar_rng(self, rews_all, n_total: int, seed: int, counter: int, Ybar, noise_scale, Ybar_out):
        self._check(self.lib.dial_shard_ybar_rng(self.h, _ptr(rews_all), n_total, int(seed), int(counter), _ptr(Ybar),
                                                 _ptr(noise_scale), int(noise_scale.numel()), _ptr(Ybar_out), _stream()),
                    "dial_shard_ybar_rng")

    def shard_pack_rewards(self, gathered, world: int, per: int, n_total: int, rews_all):
        self._check(self.lib.dial_shard_pack_rewards(self.h, _ptr(gathered), world, per, n_total, _ptr(rews_all), _stream()),
                    "dial_shard_pack_rewards")

    # phase B straight from the all-gather's receive buffer (the packing step runs inside the weights kernel; rews_all receives the
    # packed rewards): two launches per iteration instead of four
    def shard_ybar_gathered(self, gathered, world: int, per: int, n_total: int, eps_all, Ybar, noise_scale, rews_all, Ybar_out):
        self._check(self.lib.dial_shard_ybar_gathered(self.h, _ptr(gathered), world, per, n_total, _ptr(eps_all), _ptr(Ybar), _ptr(noise_scale),
                                                      int(noise_scale.numel()), _ptr(rews_all), _ptr(Ybar_out), _stream()), "dial_shard_ybar_gathered")

    def shard_ybar_gathered_rng(self, gathered, world: int, per: int, n_total: int, seed: int, counter: int, Ybar, noise_scale, rews_all, Ybar_out):
        self._check(self.lib.dial_shard_ybar_gathered_rng(self.h, _ptr(gathered), world, per, n_total, int(seed), int(counter), _ptr(Ybar),
                                                          _ptr(noise_scale), int(noise_scale.numel()), _ptr(rews_all), _ptr(Ybar_out), _stream()),
                    "dial_shard_ybar_gathered_rng")

    def shard_reduce_gathered(self, gathered, world: int, per: int, n_total: int, n_begin: int, n_local: int, include_mean: bool, rews_all, packed_out):
        self._check(self.lib.dial_shard_reduce_gathered(self.h, _ptr(gathered), world, per, n_total, n_begin, n_local, int(include_mean),
                                                        _ptr(rews_all), _ptr(packed_out), _stream()), "dial_shard_reduce_gathered")

    def packed_size(self) -> int:
        T, Hn1 = self.cfg.Hsample + 1, self.cfg.Hnode + 1
        return Hn1 * self.nu + T * (self.nq + self.nv + self.nx)

    # ---- K5
    def shift(self, Y):
        Y = Y.clone()
        self._check(self.lib.dial_shift(self.h, _ptr(Y), _stream()), "dial_shift")
        return Y

    def status(self):
        """Raise if an earlier asynchronous launch gave up (sticky; costs no synchronisation)."""
        self._check(self.lib.dial_status(self.h), "dial_status")

    # ---- measurement
    def set_timing(self, enable: bool):
        self._check(self.lib.dial_set_timing(self.h, int(enable)), "dial_set_timing")

    def rollout_ms(self):
        tot, n = ctypes.c_double(0), ctypes.c_int(0)
        self._check(self.lib.dial_get_rollout_ms(self.h, ctypes.byref(tot), ctypes.byref(n)), "dial_get_rollout_ms")
        return tot.value, n.value

    def set_state_trace(self, rows: Optional[int]):
        """Diagnostics (parity tests): allocate a [rows, T, state_size] tensor into which every following rollout launch
        writes the packed state after each env.step; rows=None switches the trace off.  Returns the tensor."""
        import torch
        if rows is None:
            self._check(self.lib.dial_set_state_trace(self.h, None, 0), "dial_set_state_trace")
            self._trace = None
            return None
        self._trace = torch.zeros((int(rows), self.cfg.Hsample + 1, self.state_size), dtype=torch.float32, device=self.torch_device)
        self._check(self.lib.dial_set_state_trace(self.h, _ptr(self._trace), int(rows)), "dial_set_state_trace")
        return self._trace

    def debug_scratch(self):
        """Host copies of the scratch tensors of the last reverse_once (tests only)."""
        import numpy as np
        import torch
        ptrs = [ctypes.c_void_p() for _ in range(6)]
        self._check(self.lib.dial_debug_scratch(self.h, *[ctypes.byref(p) for p in ptrs]), "dial_debug_scratch")
        cfg = self.cfg
        B, T, Hn1 = cfg.Nsample + 1, cfg.Hsample + 1, cfg.Hnode + 1
        shapes = [(B, Hn1, self.nu), (B, T), (B, T, self.nq), (B, T, self.nv), (B, T, self.nx), (B,)]
        names = ["Y0s", "rewss", "qss", "qdss", "xss", "weights"]
        torch.cuda.synchronize()
        hip = ctypes.CDLL("libamdhip64.so")
        out = {}
        for name, p, shp in zip(names, ptrs, shapes):
            host = np.empty(shp, np.float32)
            rc = hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), p, ctypes.c_size_t(host.nbytes), ctypes.c_int(2))
            if rc != 0:
                raise DialHipError(f"hipMemcpy of scratch {name} failed ({rc})")
            out[name] = host
        return out
