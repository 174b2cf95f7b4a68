// rollout_kernel.h -- the gfx950 kernels that run the per-sample body (rollout_driver.h): rollout_kernel (K1+K2+K3, one
// wavefront per sample, 1-9 wavefronts per workgroup sharing the staged constants), env_step_kernel / env_reset_kernel
// (K6, B = 1).  A header so that tools/isa/probe.hip can instantiate ONE kernel for ISA inspection (register / scratch /
// spill counts) in seconds instead of building the whole library; dial_hip.hip is the only product translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include "rollout_driver.h"

// Stage the dimension-specialised constants in LDS (static instantiations) and carve the workspace.
// WPB wavefronts (= samples) per workgroup share ONE staged copy of the constants; each wavefront has
// its own workspace.  WPB is chosen per robot so that N+1 = 2049 wavefronts are co-resident (>= 9 per CU):
// Go2 1 (16 KB/wave), H1 3 (10 KB constants + 3 x 13.7 KB: 3 workgroups = 9 wavefronts per CU), H1 loco 2, generic 1.
// This is the only workgroup-level barrier of the kernel (phase boundaries are
// wavefront-scope fences, wave.h).
template <class D, int WPB = 1>
__device__ __forceinline__ const CModel<D>* stage_model(const CModel<D>* gm, float* smem, Ws& s, int nnode,
                                                        int ws_words, int con_cap = 0, int tab_steps = 0) {
  const CModel<D>* m = gm;
  float* wsbase = smem;
  if constexpr (D::is_static) {
    constexpr int CMW = (int)((sizeof(CModel<D>) + 15) / 16) * 4;   // words, keeps the workspace 16-B aligned
    // 16-byte copies (the constants are padded to a multiple of 16 B on both sides: dial_create allocates CMW words): a quarter of
    // the load / store / loop instructions of the 4-byte copy, every wavefront of every launch runs this
    const uint4* src = reinterpret_cast<const uint4*>(gm);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < CMW / 4; i += 64 * WPB) dst[i] = src[i];
    __syncthreads();
    m = reinterpret_cast<const CModel<D>*>(smem);
    wsbase = smem + CMW;
  }
  // WPB == 1: LDS addresses stay immediates.  WPB > 1: the wavefront's workspace offset is made a scalar (it is
  // wave-uniform), so that addresses are SGPR base + lane offset instead of dozens of per-array VGPR bases
  if constexpr (WPB > 1) wsbase += __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * ws_words;
  ws_carve(s, wsbase, dim_nq(m), dim_nv(m), dim_nu(m), dim_nb(m), dim_nj(m), dim_ng(m), dim_ns(m), dim_nc(m),
           dim_ne(m), nnode, dial::kNeedL<D>, D::square, D::ell ? D::JCW : 0, D::gen ? con_cap : 0, D::NVP, D::pre_ctrl ? tab_steps : 0);
  return m;
}

// OCC: minimum resident wavefronts per SIMD the register allocation must allow (3: <= 168 VGPRs, 4: <= 128)
// QUEUE: the rollout-queue variant (see the loop below); the one-rollout-per-wavefront variant keeps nothing live across
// rollouts (the loop costs the headline kernel 6 spilled VGPRs, H1 21)
// TRACE: the diagnostics instantiation that also writes the per-step packed states (dial_set_state_trace)
template <class D, int WPB, int OCC = 3, bool QUEUE = false, bool TRACE = false>
__global__ void __launch_bounds__(64 * WPB, OCC)
rollout_kernel(const CModel<D>* __restrict__ gm, const dial_task* __restrict__ tg,
               const dial_cfg* __restrict__ cfg, dial::RolloutIO io, int B, int ws_words, int* __restrict__ next) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Ws s;
  const CModel<D>* m = stage_model<D, WPB>(gm, smem, s, io.Hn1, ws_words, io.con_cap, io.T);
  if constexpr (D::gen)   // this wavefront's overflow area (a slot of the grid, not of the batch: the queue reuses it)
    s.ovf = io.ovf ? io.ovf + (size_t)(blockIdx.x * WPB + (threadIdx.x >> 6)) * io.ovf_words : nullptr;
  // (the wavefront's index in its workgroup as a SCALAR: the rollout index and every row pointer derived from it are wave-uniform;
  //  left in a VGPR, the 64-bit row offsets n * T * nq ... become VGPR pairs that live through the whole kernel -- the H1's
  //  three-wavefront kernel spilled five of them to scratch, ISA probe round 5)
  const int wv = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  int n = (WPB > 1 ? (int)blockIdx.x * WPB + wv : (int)blockIdx.x) + io.n_first;
  // spread launch (a batch that fits the resident grid of multi-wavefront workgroups without filling it): wavefront w of workgroup b
  // runs rollout w x gridDim + b, so that EVERY workgroup -- every CU -- carries ceil(B / gridDim) rollouts instead of the first
  // B / WPB workgroups being full and the rest empty (Go2, N = 3000 in workgroups of 8: 119 CUs with 16 wavefronts, 137 with 8)
  if constexpr (WPB > 1 && QUEUE) {
    if (io.spread) n = wv * (int)gridDim.x + (int)blockIdx.x + io.n_first;
  }
  int relay = -1;
  if constexpr (WPB == 1 && !QUEUE) {
    if (io.relay_flag && (int)blockIdx.x >= io.relay_base) { relay = (int)blockIdx.x - io.relay_base; n = B - 1; }
  }
  int helper = -1;   // interleaved mean trajectory (RolloutIO::mean_inline): the queue holds the B - 1 noisy rollouts only
  if constexpr (QUEUE) {
    if (io.mean_inline) {
      if (n >= B - 1) return;
      if (n - io.n_first < io.T) helper = n - io.n_first;   // wavefront q < T of the first round also runs mean step q
    }
  }
  if (n >= B) return;
  Wave w;
  w.lane = threadIdx.x & 63;
  w.lane_r = w.lane;
#ifndef DIAL_LAUNDER_STATIC
#define DIAL_LAUNDER_STATIC 0   // A/B switch: the opaque lane id per step for EVERY instantiation
#endif
#ifndef DIAL_LAUNDER_ELL_QUEUE
#define DIAL_LAUNDER_ELL_QUEUE 0   // A/B switch (round 6): an opaque lane id per control step / physics frame for the Allegro's queue kernels -- 7 spilled
                                   // VGPRs -> 0, but cfg 4 +3 % (9.72 -> 10.0 ms) and the queue is no longer bit-identical to the plain launch: off
#endif
#ifndef DIAL_LAUNDER_H1
#define DIAL_LAUNDER_H1 0       // A/B switch: the opaque lane id per step for the H1's 25-dof kernels (10 spilled VGPRs without it, 1 with)
#endif
  w.launder = D::gen || OCC >= 4 || DIAL_LAUNDER_STATIC || (DIAL_LAUNDER_H1 && std::is_same<typename D::Topo, TopoH1>::value) ||
              (DIAL_LAUNDER_ELL_QUEUE && D::ell && QUEUE);   // (see wave.h)
#ifdef DIAL_PROFILE
  w.acc = reinterpret_cast<unsigned long long*>(smem + (ws_words * WPB + (D::is_static ? (int)((sizeof(CModel<D>) + 15) / 16) * 4 : 0) + 2) / 2 * 2) + 32 * (threadIdx.x >> 6);
  if (w.lane < 32) w.acc[w.lane] = 0;
  __syncthreads();
#endif
#ifdef DIAL_PROFILE
  unsigned long long t_start = wall_clock64();
#endif
  // `next` == nullptr: the grid covers the batch, one rollout per wavefront.  Otherwise the grid is exactly what the chip
  // keeps resident and every wavefront draws its next rollout from the queue head when it finishes one: rollouts differ
  // in length (solver iterations), and a workgroup's LDS is only handed to a new workgroup when its slowest wavefront
  // is done -- the queue keeps every wavefront slot busy until the batch is empty
  // Time-sliced queue (io.slice_pieces > 0, QUEUE variant): the queue holds (piece, rollout) items in piece-major order --
  // item q = piece q / B of rollout q % B -- so that all rollouts advance together and the launch ends when the WORK is done,
  // not when the slot that drew two long rollouts is (Allegro, 4097 rollouts on 2304 slots: durations 3.3 .. 11.5 ms,
  // profiles/r04_wave_times.txt).  A piece waits for its predecessor's hand-over (rollout_driver.h: the relay protocol, one
  // slot per rollout); predecessors are always EARLIER items, i.e. running or done: no deadlock.
  int q = n - io.n_first;   // queue position of this wavefront's first item (= its grid index)
  if constexpr (QUEUE) {   // (the workspace's constant entries once per wavefront, not once per queue item: see rollout_sample)
    dial::init_world(w, s);
    dial::init_square(w, m, s);
  }
  for (;;) {
    if constexpr (QUEUE) {
      if (io.slice_pieces > 0) { relay = q / B; n = q - relay * B; }
    }
    dial::rollout_sample<TRACE>(w, m, tg, cfg, s, io, n, relay, QUEUE ? helper : -1, /*init_ws=*/!QUEUE);
    helper = -1;
#ifdef DIAL_PROFILE
    if (io.prof && w.lane == 0) {   // 100 MHz wall clock; then this rollout's event counters (on-units, solver calls, LS iters, Newton iters)
      unsigned long long* p = io.prof + 32 + 6 * (size_t)n;
      p[0] = t_start; p[1] = wall_clock64(); p[2] = w.acc[27]; p[3] = w.acc[28]; p[4] = w.acc[30]; p[5] = w.acc[31];
      for (int k = 27; k < 32; k++) w.acc[k] = 0;
    }
#endif
    if constexpr (!QUEUE) break;
    if (!next) break;
    int nn = 0;
    if (w.lane == 0) nn = atomicAdd(next, 1);
    n = __builtin_amdgcn_readfirstlane(nn);
    q = n;
    if (n >= (io.slice_pieces > 0 ? io.slice_pieces * B : B - io.mean_inline)) break;
#ifdef DIAL_PROFILE
    t_start = wall_clock64();
#endif
  }
}

// TWO samples per wavefront (wave.h: WaveH): every 32-lane half owns one rollout -- its own workspace in LDS, its own rows in the
// output tensors -- and the staged constants are shared by the 2 x WPB rollouts of the workgroup.  Wavefront p of the launch runs
// rollouts 2 p and 2 p + 1 (an odd batch leaves the last wavefront's upper half idle: it returns, the EXEC mask does the rest).
// QUEUE: the grid is what the chip keeps resident and every wavefront draws its next PAIR from `next`.
// io.mean_inline (a batch whose last rollout is the mean trajectory, i.e. every reverse_once: N + 1 rollouts, N even): the mean
// trajectory is not a rollout of the grid -- N / 2 wavefronts hold the noisy pairs, and wavefront q < T of the launch's first round
// runs control step q of the mean trajectory between its own steps q and q + 1 (rollout_driver.h: helper; here BOTH halves run
// it, redundantly, on their own workspaces: same state in, same bits out, the stores coincide).  N = 2048 is then 1024
// wavefronts, one per SIMD, and N = 4096 / 8192 are one / two full rounds of the resident grid instead of "+ 1 pair".
// The body is rollout_sample -- the same per-lane program as the one-sample kernel with the 32-lane layouts of smooth_quad2.h /
// solver_reg2.h -- and produces the same bits per rollout (GPU test: test_two_samples_per_wavefront_is_bit_identical).
template <class D, int WPB, int OCC = 2, bool QUEUE = false, bool INLINE = QUEUE>
__global__ void __launch_bounds__(64 * WPB, OCC)
rollout_kernel2(const CModel<D>* __restrict__ gm, const dial_task* __restrict__ tg,
                const dial_cfg* __restrict__ cfg, dial::RolloutIO io, int B, int ws_words, int* __restrict__ next) {
  static_assert(D::is_static && dial::kQuadDims<D>, "the half-wave layouts exist for the Go2's own instantiation");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CMW = (int)((sizeof(CModel<D>) + 15) / 16) * 4;   // words, keeps the workspaces 16-B aligned
  {
    const uint4* src = reinterpret_cast<const uint4*>(gm);   // (16-byte copies, see stage_model)
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = threadIdx.x; i < CMW / 4; i += 64 * WPB) dst[i] = src[i];
    __syncthreads();   // (the only workgroup-level barrier: phase boundaries are wavefront-scope fences, wave.h)
  }
  const CModel<D>* m = reinterpret_cast<const CModel<D>*>(smem);
  WaveH w;
  w.init((int)threadIdx.x);
#ifndef DIAL_PAIR_LAUNDER
#define DIAL_PAIR_LAUNDER 0   // A/B switch: the opaque lane id per control step / physics frame (wave.h: launder) for the pair kernels
#endif
  w.launder = DIAL_PAIR_LAUNDER;
  Ws s;
  ws_carve(s, smem + CMW + (int)(threadIdx.x >> 5) * ws_words, dim_nq(m), dim_nv(m), dim_nu(m), dim_nb(m), dim_nj(m), dim_ng(m),
           dim_ns(m), dim_nc(m), dim_ne(m), io.Hn1, false, true, 0, 0, D::NVP, D::pre_ctrl ? io.T : 0);
#ifdef DIAL_PROFILE
  w.acc = reinterpret_cast<unsigned long long*>(smem + (CMW + 2 * WPB * ws_words + 2) / 2 * 2) + 32 * (threadIdx.x >> 6);
  if ((threadIdx.x & 63) < 32) w.acc[threadIdx.x & 63] = 0;
  __syncthreads();
#endif
  int pair = (WPB > 1 ? (int)blockIdx.x * WPB + (int)(threadIdx.x >> 6) : (int)blockIdx.x);
  // (INLINE: the instantiation carries the interleaved mean trajectory's hand-over code -- the queue variant always, the plain grid as
  //  a second instantiation: the code costs registers and a lone wavefront's pace, which batches that do not need it should not pay)
  const int Bq = INLINE ? B - io.mean_inline : B;                  // rollouts the grid / the queue holds
  int helper = INLINE && io.mean_inline && pair < io.T ? pair : -1;   // (first round only)
  if constexpr (QUEUE) {   // (the workspaces' constant entries once per wavefront, not once per pair drawn from the queue)
    dial::init_world(w, s);
    dial::init_square(w, m, s);
  }
  for (;;) {
    const int n = 2 * pair + w.half + io.n_first;
    // (measured and not kept: the highest issue priority for the odd wavefront of a batch -- N + 1 = 2049 is 1024 full
    //  wavefronts and the mean trajectory alone in the 1025th, which shares a SIMD -- starves its SIMD-mate: 0.408 -> 0.440 ms)
    if (n < Bq) dial::rollout_sample<false>(w, m, tg, cfg, s, io, n, -1, helper, /*init_ws=*/!QUEUE);
    helper = -1;
    if constexpr (!QUEUE) break;
    if (!next) break;
    int nn = 0;
    if ((threadIdx.x & 63) == 0) nn = atomicAdd(next, 1);
    pair = __builtin_amdgcn_readfirstlane(nn);
    if (2 * pair + io.n_first >= Bq) break;
  }
}

template <class D>
__global__ void __launch_bounds__(64)
env_step_kernel(const CModel<D>* __restrict__ gm, const dial_task* __restrict__ tg, float* state,
                const float* action, float* xpos_out, float* xquat_out, float* ctrl_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Ws s;
  const CModel<D>* m = stage_model<D>(gm, smem, s, 0, 0);
  Wave w;
  w.lane = threadIdx.x;
  w.lane_r = w.lane;
  w.launder = D::gen;
  dial::env_step_single(w, m, tg, s, state, action, xpos_out, xquat_out, ctrl_out);
}

template <class D>
__global__ void __launch_bounds__(64)
env_reset_kernel(const CModel<D>* __restrict__ gm, const float* qpos, const float* qvel, float* state,
                 float* xpos_out, float* xquat_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Ws s;
  const CModel<D>* m = stage_model<D>(gm, smem, s, 0, 0);
  Wave w;
  w.lane = threadIdx.x;
  w.lane_r = w.lane;
  // (dial_env_reset_batch: workgroup b resets state b of a batch -- rows of qpos / qvel / state / the optional pose outputs)
  const int b = (int)blockIdx.x, nq = dim_nq(m), nv = dim_nv(m), nb1 = dim_nb(m) - 1;
  dial::env_reset_single(w, m, s, qpos + (size_t)b * nq, qvel + (size_t)b * nv, state + (size_t)b * (nq + 2 * nv + DIAL_INFO_N),
                         xpos_out ? xpos_out + (size_t)b * nb1 * 3 : nullptr, xquat_out ? xquat_out + (size_t)b * nb1 * 4 : nullptr);
}

