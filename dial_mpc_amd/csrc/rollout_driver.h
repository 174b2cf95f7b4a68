// rollout_driver.h -- what one wavefront does for one sample: stage the shared initial state in LDS,
// build its control nodes (K1) and controls (K2), run T env.steps (K3) and stream the per-step
// outputs of MBDPI.rollout_us_vmap to HBM.  Shared by the HIP kernels and the host wave emulator.
// On the GPU a wavefront may also run one PIECE of the mean-trajectory rollout (relay, see rollout_sample).
//
// Reference: dial_mpc/core/dial_core.py:106-117 (sampling + node2u), :36-42 (rollout_us).
#pragma once
#include "rollout_body.h"
#include "philox.h"
// (struct RolloutIO: rollout_io.h, included by wave.h -- the phases that store a step's outputs name its fields)

namespace dial {


template <class W, class M>
DIAL_DEV void load_state(W& w, const M* m, const Ws& s, const float* state) {
  const int nq = dim_nq(m), nv = dim_nv(m);
  // (the Allegro's queue kernels: the lanes' 64-bit addresses into the hand-over slot are loop invariants of the queue's loop over rollout
  //  pieces -- hoisted and spilled; an opaque lane id keeps them inside the call)
  DIAL_LANE_SCOPE_IF(M::D::ell, w);
  w.items(nq + 2 * nv + DIAL_INFO_N, [&](int i) {
    float v = state[i];
    if (i < nq) s.qpos[i] = v;
    else if (i < nq + nv) s.qvel[i - nq] = v;
    else if (i < nq + 2 * nv) s.warm[i - nq - nv] = v;
    else s.info[i - nq - 2 * nv] = v;
  });
}
template <class W, class M>
DIAL_DEV void store_state(W& w, const M* m, const Ws& s, float* state) {
  const int nq = dim_nq(m), nv = dim_nv(m);
  DIAL_LANE_SCOPE_IF(M::D::ell, w);   // (see load_state)
  w.items(nq + 2 * nv + DIAL_INFO_N, [&](int i) {
    float v;
    if (i < nq) v = s.qpos[i];
    else if (i < nq + nv) v = s.qvel[i - nq];
    else if (i < nq + 2 * nv) v = s.warm[i - nq - nv];
    else v = s.info[i - nq - 2 * nv];
    state[i] = v;
  });
}

#ifndef DIAL_RSUM_FP64
#define DIAL_RSUM_FP64 1   // A/B switch: 0 = the fp32 running reward sum of rounds 1-5
#endif
#if DIAL_RSUM_FP64
using rsum_t = double;
#else
using rsum_t = float;
#endif
// the running reward sum (fp64) in two 32-bit words of a hand-over slot (the slots are 4-byte aligned)
#ifndef DIAL_EMU
DIAL_DEV void store_sum(float* p, double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  reinterpret_cast<uint32_t*>(p)[0] = (uint32_t)b;
  reinterpret_cast<uint32_t*>(p)[1] = (uint32_t)(b >> 32);
}
DIAL_DEV double load_sum(const float* p) {
  const uint32_t lo = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t hi = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#endif

// `relay` >= 0: this wavefront runs piece `relay` of the mean-trajectory rollout (RolloutIO::relay_*): it prepares
// everything that does not depend on its predecessor, waits for the predecessor's state, runs its control steps at the
// highest issue priority and hands the state on.  Every piece is a different wavefront on a different SIMD, so the
// "+1" rollout loads no SIMD for longer than relay_steps control steps.
// TRACE (compile time): also write the packed state after every env.step to io.trace -- its own kernel instantiation, because
// even a null-pointer test costs the production launch (measured +1.4 % on the Go2 headline: one more live kernel argument in a
// kernel that already spills scalar registers; profiles/r04_ab_trace_hook.txt).
// `helper` >= 0 (io.mean_inline launches, GPU only): after its own control step `helper` this wavefront parks its rollout's packed
// state in two registers per lane, takes the mean trajectory's state from its predecessor (wavefront helper - 1 of the same
// launch, which ran mean step helper - 1 one step earlier: both advance at the same pace, so the wait is short), runs mean
// step `helper` through the SAME env_step call site, hands the state on and resumes its own rollout.  The mean trajectory
// advances one step per step of everybody else and ends with them: a batch of k x slots + 1 rollouts takes k rounds and one
// STEP instead of k rounds and one lone ROLLOUT (Go2, N = 4096: 0.78 ms as a queue item).  Bit-identical: the same steps on
// the same states.
template <bool TRACE = false, class W, class M>
DIAL_DEV void rollout_sample(W& w, const M* m, const dial_task* tg, const dial_cfg* cfg, const Ws& s,
                             const RolloutIO& io, int n, int relay = -1, int helper = -1, bool init_ws = true) {
  const int nq = dim_nq(m), nv = dim_nv(m), nu = dim_nu(m), nx = (dim_nb(m) - 1) * 3, T = io.T, Hn1 = io.Hn1;
#ifdef DIAL_PROFILE
  w.tprev = __builtin_readcyclecounter();
#endif
  // The workspace entries no phase ever rewrites -- the world body's pose, the structural zeros of the square M / the contact
  // Jacobian, the fused contacts' frames -- are written by the wavefront's FIRST rollout (piece) only: a queue kernel's later items
  // find them in place.  (Per item, their lane-derived LDS addresses were loop invariants of the queue's loop, hoisted and spilled:
  // 4 VGPRs / 20 B of scratch of the Allegro's time-sliced queue kernel, ISA round 6.)
  if (init_ws) {
    init_world(w, s);
    init_square(w, m, s);
  }
  const int nstate = nq + 2 * nv + DIAL_INFO_N;
  int st_begin = 0, st_end = T;
  if (relay >= 0) { st_begin = relay * io.relay_steps; st_end = st_begin + io.relay_steps < T ? st_begin + io.relay_steps : T; }
  if (relay <= 0) load_state(w, m, s, io.state);
  if (!io.us) {
    // K1: candidate nodes (dial_core.py:110-115)
    w.items(Hn1 * nu, [&](int it) {
      const int k = it / nu, a = it - k * nu;
      float v;
      if (n < io.n_noise) {
        float sc = io.noise_scale[io.ns == 1 ? 0 : k];
        float e;
        if (io.use_rng) {
          float z[4];
          normal_quad((uint32_t)(io.n_offset + n), (uint32_t)(it >> 2), io.rng_iter, io.seed_lo, io.seed_hi, z);
          e = z[it & 3];
        } else {
          e = io.eps[((size_t)n * Hn1 + k) * nu + a];
        }
        v = e * sc + io.Ybar[k * nu + a];
        if (k == 0) v = io.Ybar[a];
      } else {
        v = io.Ybar[k * nu + a];
      }
      v = dm::clip(v, -1.f, 1.f);
      s.Y[it] = v;
      if (io.Y0s && relay <= 0) io.Y0s[(size_t)n * Hn1 * nu + it] = v;   // (pieces of one rollout rebuild the same nodes: the first writes them)
    });
  }
  // Dims::pre_ctrl: K2 (node2u as the constant map W, dial_core.py:92-95,117) and act2joint for ALL T control steps at once -- neither
  // depends on the state.  Per step the loop below then only runs act2tau.  (Per step, K2 + act2joint + the gait clock were 3.0 k of a
  // lone Go2 wavefront's 39.3 k cycles per env.step: 12 + 4 busy lanes waiting for scalar loads, LDS round trips and a cosine.)
  constexpr bool PRE = M::D::pre_ctrl;
  // the control of step t, actuator a: the same products in the same order as the per-step K2 of the other instantiations
  const auto node2u = [&](int t, int a) -> float {
    if (io.us) return io.us[(unsigned)((n * T + t) * nu + a)];   // 32-bit offset from the uniform base: no 64-bit VGPR pair kept live
    float u = 0.f;
    // the examples' node counts with a compile-time trip count: the row of W arrives with one or two scalar loads and every LDS
    // fetch is issued up front (a loop whose length is a run-time value waits for one scalar load + one LDS fetch per node)
    const auto k2 = [&](auto HN) {
      DIAL_UNROLL_FULL
      for (int k = 0; k < decltype(HN)::value; k++) u += cfg->W[t][k] * s.Y[k * nu + a];
    };
    if (Hn1 == 5) k2(std::integral_constant<int, 5>{});
    else if (Hn1 == 6) k2(std::integral_constant<int, 6>{});
    else if (Hn1 == 7) k2(std::integral_constant<int, 7>{});
    else for (int k = 0; k < Hn1; k++) u += cfg->W[t][k] * s.Y[k * nu + a];
    return u;
  };
  if constexpr (PRE) {
    w.items(T * nu, [&](int it) {
      const int t = it / nu, a = it - t * nu;
      s.jtab[it] = act2joint(m, node2u(t, a), a);
    });
  }
  // the rollout's reward sum in fp64 (round 6): the mean reward then is the correctly rounded mean of the T fp32 step rewards.  With
  // rewards near 10 (seq-jump: alive x 10) an fp32 running sum was up to 2.7 ulp off, and at an effective sample size of 1 .. 2 the
  // softmax turns one ulp of a mean reward into 1.5e-4 of logit (tools/k4_sensitivity.py): the device's Ybar left its 1e-4 gate
  // against the fp64 K4 of its own rollouts
  rsum_t rsum = 0;
  w.set_rollout(n);
#ifndef DIAL_EMU
  float* const rbuf = io.relay_buf ? io.relay_buf + (size_t)n * io.relay_stride : nullptr;   // this rollout's hand-over slot
  int* const rflag = io.relay_flag ? io.relay_flag + (io.relay_stride ? n : 0) : nullptr;
  if (relay > 0) {
    // wait for the predecessor (it was dispatched before this wavefront: it is running or done), then take its state
    // (bounded: a wavefront that never gets its turn -- ~0.2 s -- gives up instead of hanging the GPU: it raises the
    // context's sticky error word, which every later API call reports (dial_status), marks the rollout's reward with a
    // NaN bit pattern and RETURNS -- it neither runs on a stale state nor hands over, so its successors give up too)
    int timed_out = 0;
    if (w.lane == 0) {
      unsigned spins = 0;
      const unsigned spin_max = io.relay_stride ? (1u << 23) : (1u << 20);   // (sliced queue: a predecessor may itself be waiting)
      while (__hip_atomic_load(rflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != relay && ++spins < spin_max)
        __builtin_amdgcn_s_sleep(4);
      timed_out = spins >= spin_max;
    }
    timed_out = __builtin_amdgcn_readfirstlane(timed_out);
    if (timed_out) {
      if (w.lane == 0) {
        if (io.err_word) __hip_atomic_store(io.err_word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (io.rews) reinterpret_cast<uint32_t*>(io.rews)[n] = 0x7fc00000u;
      }
      return;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    load_state(w, m, s, rbuf);
    rsum = load_sum(rbuf + nstate);
  }
  if (relay >= 0 && io.relay_stride == 0) w.hold_priority(3);   // (the lone relay rollout; the sliced queue keeps the fair sharing)
#endif
  if constexpr (PRE) {
    // the gait clock of this (piece of the) rollout: step counter of control step t = the counter now + (t - st_begin), exactly
    // (env.step adds 1.f per step); the mean trajectory's steps (helper) share the own rollout's counter, both start from io.state
    const bool walk = m->kind == DIAL_TASK_GO2_WALK || m->kind == DIAL_TASK_H1_WALK || m->kind == DIAL_TASK_H1_LOCO || m->kind == DIAL_TASK_H1_PUSH_CRATE;
    // ... and the step's ramped velocity targets (unitree_go2_env.py:190-215; the expressions the reward lane evaluated in every step)
    static_assert(DIAL_ZTAB_W == DIAL_MAX_FEET + 4, "ztab row: foot heights | v_x v_y yaw-rate yaw targets");
    if (walk) w.items((st_end - st_begin) * DIAL_ZTAB_W, [&](int it) {
      const int dt_ = it / DIAL_ZTAB_W, f = it - dt_ * DIAL_ZTAB_W;
      const float step = s.info[DIAL_INFO_STEP] + (float)dt_;
      float* row = s.ztab + (st_begin + dt_) * DIAL_ZTAB_W;
      if (f < DIAL_MAX_FEET) { if (f < m->nfeet) row[f] = gait_ztar(m, f, step); return; }
      float cmd[6];
      step_cmd(m, tg, step, cmd);
      const float dt = m->dt;
      if (f < DIAL_MAX_FEET + 2) { const float v = cmd[f - DIAL_MAX_FEET]; row[f] = dm::fminf_(v * step * dt / m->ramp_up_time, v); return; }
      const float a2 = cmd[5];
      const float avt = dm::fminf_(a2 * step * dt / m->ramp_up_time, a2);
      row[f] = f == DIAL_MAX_FEET + 2 ? avt : s.info[DIAL_INFO_YAW_TAR] + avt * dt * step;
    });
  }
  // element i of the packed state [qpos | qvel | qacc_warmstart | info] in the LDS workspace
  // (an offset from ONE base pointer, not a select of pointers: keeps the accesses in the LDS address space)
  const auto packed = [&](int i) -> float* {
    const int off = i < nq ? i : (i < nq + nv ? (int)(s.qvel - s.qpos) + (i - nq)
                    : (i < nq + 2 * nv ? (int)(s.warm - s.qpos) + (i - nq - nv) : (int)(s.info - s.qpos) + (i - nq - 2 * nv)));
    return s.qpos + off;
  };
  bool mean_now = false;   // this pass of the loop runs the mean trajectory's step `helper`, not an own step
  // the parked state: element l + k * KSTRIDE of the packed state in register k of (logical) lane l
  constexpr int KSTRIDE = W::half2 ? 32 : 64, NKEEP = W::half2 ? 4 : 2;
  float keep[NKEEP] = {};
  rsum_t msum = 0;
  (void)packed; (void)keep; (void)msum;
  for (int st_own = st_begin; mean_now || st_own < st_end;) {
    const int st = mean_now ? helper : st_own;    // control step this pass runs ...
    const int row = mean_now ? io.n_noise : n;    // ... of this rollout (the mean trajectory is rollout n_noise)
    w.redraw_priority();
#ifndef DIAL_EMU
    // generic feature set: lane-derived LDS addresses and masks are loop invariants of this T-step loop; hoisted, they are
    // dozens of VGPRs that live for the whole kernel and get spilled to scratch (round 4 ISA probe of the crate-climb kernel:
    // 73 spilled VGPRs, 36 of them stored right here at the loop entry; with the opaque copy of the lane id per step and per
    // physics frame: 0 spilled VGPRs, 0 B of scratch).  (The dimension-specialised register-solver kernels have no spills to cure: there the hoisted addresses pay for
    // themselves -- measured 5 % slower with the laundering, DESIGN.md.)
    if (w.launder) { asm volatile("" : "+v"(w.lane)); w.lane_r = w.lane; }   // (and once per physics frame: rollout_body.h env_step)
#endif
    // K2: node2u as the constant linear map W (dial_core.py:92-95,117)
    if constexpr (!PRE) {
      // (row-layout robots: the lanes' addresses into the node array stay inside this phase -- hoisted out of the T-step loop one
      //  of them was the last VGPR the H1's four-wavefront kernel spilled)
      DIAL_LANE_SCOPE_IF(kRowsDims<typename M::D> && !M::D::gen, w);
      w.items(nu, [&](int a) {
        float u;
        if (mean_now && !io.us) {   // the mean trajectory's nodes are clip(Ybar): not in this wavefront's LDS (s.Y holds its own rollout's)
          u = 0.f;
          for (int k = 0; k < Hn1; k++) u += cfg->W[st][k] * dm::clip(io.Ybar[k * nu + a], -1.f, 1.f);
        } else {
          u = node2u(st, a);
        }
        s.act[a] = u;
      });
    } else {
#ifndef DIAL_EMU
      if (mean_now) {   // the table row of step `helper` (the own rollout is past it) takes the mean trajectory's joint targets
        w.items(nu, [&](int a) {
          float u = 0.f;
          for (int k = 0; k < Hn1; k++) u += cfg->W[st][k] * dm::clip(io.Ybar[k * nu + a], -1.f, 1.f);
          s.jtab[st * nu + a] = act2joint(m, u, a);
        });
      }
#endif
    }
    DIAL_MARK(w, 11);
    const int work0 = w.work;
    if constexpr (PRE) { w.out_io = &io; w.out_row = row * T + st; }   // (the step's outputs are stored by the phases that produce them)
    float rew = env_step<false, PRE>(w, m, tg, s, st);
    if (mean_now) msum += (rsum_t)rew; else rsum += (rsum_t)rew;
#ifndef DIAL_EMU
    if constexpr (M::D::ell) {
      if (io.work_stat && relay < 0) {   // once per control step: add own work to the launch totals, compare rates, set the level
        int lvl = 0;
        if (w.lane == 0) {
          const int tot_w = atomicAdd(io.work_stat, w.work - work0) + (w.work - work0);
          const int tot_s = atomicAdd(io.work_stat + 1, 1) + 1;
          // own rate vs the running average: w.work / (st + 1 - st_begin)  <>  tot_w / tot_s
          const float own = (float)w.work * (float)tot_s, avg = (float)tot_w * (float)(st + 1 - st_begin);
#ifndef DIAL_LAG_T1
#define DIAL_LAG_T1 1.0f    // thresholds measured on the Allegro example (profiles/r04_ab_lag_priority.txt): 0.85 / 1.05 / 1.25 -> 7.29 ms,
#define DIAL_LAG_T2 1.2f    // 1.0 / 1.2 / 1.4 -> 7.17 ms, 0.9 / 1.0 / 1.1 -> 7.42 ms, 1.1 / 1.3 / 1.5 and 1.0 / 1.3 / 1.6 -> 7.2 ms;
#define DIAL_LAG_T3 1.4f    // fair pseudo-random sharing (options.no_lag_priority) 7.48 ms
#endif
          lvl = own > DIAL_LAG_T3 * avg ? 3 : (own > DIAL_LAG_T2 * avg ? 2 : (own > DIAL_LAG_T1 * avg ? 1 : 0));
        }
        w.prio_level = __builtin_amdgcn_readfirstlane(lvl);
      }
    }
#endif
    const size_t o = (size_t)row * T + st;
    if constexpr (!PRE) {
      // per-step outputs: wave-uniform row pointers (scalar address arithmetic), one pass -- lane i stores element i
      // of each row that is that long
      float* const qrow = io.qss ? io.qss + o * nq : nullptr;
      float* const qdrow = io.qdss ? io.qdss + o * nv : nullptr;
      float* const xrow = io.xss ? io.xss + o * nx : nullptr;
      float* const rrow = io.rewss ? io.rewss + o : nullptr;
      const int nmax = nx > nq ? nx : nq;   // nv < nq
      w.items(nmax, [&](int i) {
        if (qrow && i < nq) qrow[i] = s.qpos[i];
        if (qdrow && i < nv) qdrow[i] = s.qvel[i];
        if (xrow && i < nx) xrow[i] = s.xpos[3 + i];
        if (rrow && i == 0) rrow[0] = rew;
      });
    }
    (void)o;
#ifndef DIAL_PROFILE   // (profiling builds carry no state trace: the combination trips an LLVM address-space bug in the DimsMax kernel)
    if constexpr (TRACE) { if (io.trace) store_state(w, m, s, io.trace + o * nstate); }
#endif
    DIAL_MARK(w, 24);
#ifndef DIAL_EMU
    if (mean_now) {
      // hand the mean trajectory on (or finish it), then resume the own rollout from the parked state
      if (helper + 1 < T) {
        store_state(w, m, s, io.relay_buf);
        w.items(1, [&](int) { store_sum(io.relay_buf + nstate, msum); });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (w.lane == 0) __hip_atomic_store(io.relay_flag, helper + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (io.rews) { const float mean = (float)(msum / (rsum_t)T); w.items(1, [&](int) { io.rews[row] = mean; }); }
        if (w.lane == 0) __hip_atomic_store(io.relay_flag, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
      }
      w.items(KSTRIDE, [&](int l) {
#pragma unroll
        for (int k = 0; k < NKEEP; k++) if (l + k * KSTRIDE < nstate) *packed(l + k * KSTRIDE) = keep[k];
      });
      mean_now = false;
      continue;
    }
#endif
    st_own++;
#ifndef DIAL_EMU
    if (helper >= 0 && st == helper && relay < 0) {   // own step `helper` is done: the mean trajectory's step `helper` comes next
#pragma unroll
      for (int k = 0; k < NKEEP; k++) keep[k] = w.lane + k * KSTRIDE < nstate ? *packed(w.lane + k * KSTRIDE) : 0.f;
      w.sync();
      bool ok = true;
      if (helper == 0) {
        load_state(w, m, s, io.state);
        msum = 0;
        if (io.Y0s) {
          // (an opaque lane id for this once-per-launch copy: its two lane-derived 64-bit global addresses were hoisted out of the
          //  queue's rollout loop and spilled -- the 8 B + 8 B of scratch of the Allegro's time-sliced queue kernel, ISA round 6)
          DIAL_LANE_SCOPE(w);
          w.items(Hn1 * nu, [&](int it) { io.Y0s[(size_t)io.n_noise * Hn1 * nu + it] = dm::clip(io.Ybar[it], -1.f, 1.f); });
        }
      } else {
        int timed_out = 0;
        if (w.lane == 0) {   // (bounded like the relay's wait: give up, raise the sticky error, never run on a stale state)
          unsigned spins = 0;
          while (__hip_atomic_load(io.relay_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != helper && ++spins < (1u << 20))
            __builtin_amdgcn_s_sleep(2);
          timed_out = spins >= (1u << 20);
        }
        timed_out = __builtin_amdgcn_readfirstlane(timed_out);
        if (timed_out) {
          if (w.lane == 0) {
            if (io.err_word) __hip_atomic_store(io.err_word, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (io.rews) reinterpret_cast<uint32_t*>(io.rews)[io.n_noise] = 0x7fc00000u;
          }
          ok = false;
        } else {
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          load_state(w, m, s, io.relay_buf);
          msum = load_sum(io.relay_buf + nstate);
        }
      }
      mean_now = ok;   // (!ok: the predecessor never came -- the own rollout resumes, its state is still in the workspace)
    }
#endif
  }
#ifdef DIAL_PROFILE
  DIAL_MARK(w, 11);
  if (io.prof && n == 0 && w.lane == 0) for (int k = 0; k < 28; k++) io.prof[k] = w.acc[k];
  // event counters 28..31 are summed over ALL samples (28/29: Newton iterations 2 / with an unchanged active set,
  // 30/31: line-search iterations / Newton iterations)
  if (io.prof && w.lane == 0) for (int k = 28; k < DIAL_NSEC; k++) atomicAdd(&io.prof[k], w.acc[k]);
#endif
#ifndef DIAL_EMU
  if (relay >= 0 && relay + 1 == io.debug_stall_piece1) return;   // test hook: a piece that never hands over (its successors time out)
  if (relay >= 0 && st_end < T) {   // hand over: state, running sum, then the flag (release)
    store_state(w, m, s, rbuf);
    w.items(1, [&](int) { store_sum(rbuf + nstate, rsum); });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (w.lane == 0) __hip_atomic_store(rflag, relay + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (relay >= 0 && w.lane == 0) __hip_atomic_store(rflag, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // last piece: re-arm
#endif
  if (io.rews) {
    const float mean = (float)(rsum / (rsum_t)T);
    w.items(1, [&](int) { io.rews[n] = mean; });
  }
}

// env.step on the true state (B = 1)
template <class W, class M>
DIAL_DEV void env_step_single(W& w, const M* m, const dial_task* tg, const Ws& s, float* state,
                              const float* action, float* xpos_out, float* xquat_out, float* ctrl_out) {
  init_world(w, s);
  init_square(w, m, s);
  load_state(w, m, s, state);
  w.items(dim_nu(m), [&](int a) { s.act[a] = action[a]; });
  env_step<true>(w, m, tg, s);
  store_state(w, m, s, state);
  const int nb1 = dim_nb(m) - 1;
  w.items(nb1 * 7 + dim_nu(m), [&](int i) {
    if (i < nb1 * 3) { if (xpos_out) xpos_out[i] = s.xpos[3 + i]; }
    else if (i < nb1 * 7) { if (xquat_out) xquat_out[i - nb1 * 3] = s.xquat[4 + (i - nb1 * 3)]; }
    else { if (ctrl_out) ctrl_out[i - nb1 * 7] = s.ctrl[i - nb1 * 7]; }
  });
}

// env.reset: pipeline_init(q, qd) = mjx.forward with ctrl = 0, then the task's initial info
template <class W, class M>
DIAL_DEV void env_reset_single(W& w, const M* m, const Ws& s, const float* qpos, const float* qvel, float* state,
                               float* xpos_out, float* xquat_out) {
  init_world(w, s);
  init_square(w, m, s);
  const int nq = dim_nq(m), nv = dim_nv(m);
  w.items(nq + 2 * nv + DIAL_INFO_N + dim_nu(m), [&](int i) {
    if (i < nq) s.qpos[i] = qpos[i];
    else if (i < nq + nv) s.qvel[i - nq] = qvel[i - nq];
    else if (i < nq + 2 * nv) s.warm[i - nq - nv] = 0.f;
    else if (i < nq + 2 * nv + DIAL_INFO_N) {
      const int k = i - nq - 2 * nv;
      s.info[k] = (k >= DIAL_INFO_POS_TAR && k < DIAL_INFO_POS_TAR + 3) ? m->init_pos_tar[k - DIAL_INFO_POS_TAR]
                  : ((k >= DIAL_INFO_ANG_VEL_TAR && k < DIAL_INFO_ANG_VEL_TAR + 3) ? m->init_ang_vel_tar[k - DIAL_INFO_ANG_VEL_TAR] : 0.f);
    } else s.ctrl[i - nq - 2 * nv - DIAL_INFO_N] = 0.f;
  });
  forward(w, m, s);
  store_state(w, m, s, state);
  const int nb1 = dim_nb(m) - 1;
  w.items(nb1 * 7, [&](int i) {
    if (i < nb1 * 3) { if (xpos_out) xpos_out[i] = s.xpos[3 + i]; }
    else { if (xquat_out) xquat_out[i - nb1 * 3] = s.xquat[4 + (i - nb1 * 3)]; }
  });
}

}  // namespace dial
