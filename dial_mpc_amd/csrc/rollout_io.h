// rollout_io.h -- the arguments of one rollout launch (K1 + K2 + K3): inputs, output tensors of MBDPI.rollout_us_vmap, launch-level
// mechanisms.  A header of its own because the phases that PRODUCE a control step's outputs (smooth_quad.h, rollout_body.h: euler,
// reward) store them straight to these tensors (Wave::out_io / out_row) in the instantiations with Dims::pre_ctrl.
//
// Reference: dial_mpc/core/dial_core.py:36-42 (rollout_us), :80-81 (the vmap over samples), :106-117 (sampling + node2u).
#pragma once
#include <cstdint>

namespace dial {

struct RolloutIO {
  const float* state;        // packed initial state, shared by all samples
  const float* us;           // [B,T,nu] controls, or nullptr -> build them from nodes
  const float* eps;          // [n_noise,Hn1,nu] standard-normal draws (nodes mode)
  const float* Ybar;         // [Hn1,nu]
  const float* noise_scale;  // [ns]
  int ns;
  int n_noise;               // samples with index >= n_noise roll out the mean trajectory Ybar
  int T, Hn1;
  float* Y0s;                // out [B,Hn1,nu] (nodes mode) or nullptr
  float* rewss;              // out [B,T] or nullptr
  float* rews;               // out [B] mean over T, or nullptr
  float* qss;                // out [B,T,nq] or nullptr
  float* qdss;               // out [B,T,nv] or nullptr
  float* xss;                // out [B,T,(nbody-1)*3] or nullptr
  unsigned long long* prof;  // DIAL_PROFILE builds: per-section cycle counts of sample 0, else nullptr
  // in-kernel noise (eps == nullptr && use_rng): Philox keyed by seed, counter = (n_offset + n, quad, rng_iter)
  int use_rng;
  uint32_t seed_lo, seed_hi, rng_iter;
  int n_offset;              // global index of this launch's sample 0 (sample shards)
  // mean-trajectory relay (GPU launches whose last rollout is the mean trajectory; nullptr / 0 otherwise): the extra
  // rollout is cut into pieces of `relay_steps` control steps, each run by its own wavefront on a different SIMD
  float* relay_buf;          // packed state + running reward sum handed from piece to piece
  int* relay_flag;           // index of the piece that may run
  int relay_steps;
  // time-sliced rollout queue (batches beyond the resident set whose rollouts differ in length, rollout_kernel.h): EVERY rollout
  // is cut into pieces, relay_buf / relay_flag are arrays with one slot per rollout (relay_stride floats apart); 0: the classic
  // relay of the mean trajectory alone (one slot)
  int relay_stride;
  int slice_pieces;          // pieces per rollout (time-sliced queue), else 0
  // lag-based issue priority (wave.h; models with data-dependent rollout lengths, everything resident): [0] solver iterations,
  // [1] control steps completed by all rollouts of the launch so far; nullptr: the pseudo-random fair sharing
  int* work_stat;
  int relay_base;            // index of the first relay workgroup of the launch
  int n_first;               // rollout index of the launch's first wavefront (split launches)
  int* err_word;             // host-visible sticky error word of the context (relay time-out), or nullptr
  int debug_stall_piece1;    // test hook (DIAL_DEBUG_RELAY_STALL=k): relay piece k - 1 never hands over; 0 = off
  // generic instantiation: contact cap of the LDS workspace (derived.h: ws_carve) and the per-wavefront overflow areas in
  // global memory (ovf_words each, indexed by the wavefront's slot in the grid); con_cap = 0: full-size LDS workspace
  int con_cap;
  float* ovf;
  int ovf_words;
  // diagnostics (dial_set_state_trace; nullptr in production): the packed state [qpos|qvel|qacc_warmstart|info] after every
  // env.step, trace:[B,T,nstate] -- what the per-transition parity tests restart the oracle from
  float* trace;
  // interleaved mean trajectory (rollout-queue launches whose last rollout is the mean trajectory: N + 1 = k x the resident set
  // + 1 whenever Nsample is a power of two): the mean rollout is NOT a queue item; wavefront q < T of the launch's first round
  // runs control step q of it between its own steps q and q + 1 (state handed on through relay_buf / relay_flag, T hand-overs
  // per launch), so that no wavefront slot runs two whole rollouts one after the other -- see rollout_sample
  int mean_inline;
  int spread;   // rollout_kernel.h: the spread launch (rollout index = wavefront-in-workgroup x grid + workgroup)
};

}  // namespace dial
