// dial_hip.hip -- gfx950 kernels and the C ABI of libdialhip.so (see include/dial_mpc.h).
//
// Kernels
//   rollout_kernel   K1+K2+K3: one wavefront per sample (1-9 wavefronts per workgroup share the staged
//                    constants); per-sample state and constants in LDS, per-step outputs streamed to HBM in
//                    the layout of MBDPI.rollout_us_vmap.  How the N + 1 wavefronts of a launch are put on the
//                    chip is decided in launch_rollout(): rollout queue (batch > resident set), mean-trajectory
//                    relay (one-wavefront workgroups, N a multiple of the SIMD count), split launch (H1 / Allegro
//                    at N = 8 x CUs); every wavefront re-draws its issue priority (wave.h: redraw_priority).
//   weights_kernel   K4a: rew_bar, std, softmax over all N+1 mean rewards (one workgroup, fixed
//                    reduction order => bit-identical on every GPU of a sharded run).
//   wsum_*_kernel    K4b: weighted means of Y0s / q / qd / x.pos, two deterministic passes.
//   shift_kernel     K5, env_step_kernel / env_reset_kernel  K6 (B = 1).
// There is no CPU fallback: every entry point needs a HIP device and fails with DIAL_ERR_HIP otherwise.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernel_list.h"

// the kernels are compiled in their own translation units, one per robot family (kern_family.hip, built in parallel)
#define DIAL_X(D, WPB, OCC, Q, TR) \
  extern template __global__ void rollout_kernel<D, WPB, OCC, Q, TR>(const CModel<D>*, const dial_task*, const dial_cfg*, dial::RolloutIO, int, int, int*);
#define DIAL_XE(D)                                                                                                              \
  extern template __global__ void env_step_kernel<D>(const CModel<D>*, const dial_task*, float*, const float*, float*, float*, float*); \
  extern template __global__ void env_reset_kernel<D>(const CModel<D>*, const float*, const float*, float*, float*, float*);
DIAL_KERNELS_ALL(DIAL_X, DIAL_XE)
#define DIAL_X2(D, WPB, OCC, Q, MI) \
  extern template __global__ void rollout_kernel2<D, WPB, OCC, Q, MI>(const CModel<D>*, const dial_task*, const dial_cfg*, dial::RolloutIO, int, int, int*);
DIAL_KERNELS2_GO2(DIAL_X2)
#undef DIAL_X2
#undef DIAL_X
#undef DIAL_XE

#define WSUM_CHUNKS 64
#define YB_CHUNKS_HOST 128   // = YB_CHUNKS (defined with its kernel below)

#ifndef DIAL_GO2_LARGE_B
#define DIAL_GO2_LARGE_B 2304   /* batches above this many rollouts use the large-batch instantiation */
#endif
#ifndef DIAL_GO2_PAIR_MIN_B
#define DIAL_GO2_PAIR_MIN_B 2304   /* batches above this many rollouts run two per wavefront (rollout_kernel2); see launch_rollout */
#endif

// ------------------------------------------------------------------ kernels (rollout / env.step / env.reset: rollout_kernel.h)
// Deterministic block reductions (1024 threads = 16 wavefronts): DPP butterfly inside each wavefront, then a
// fixed-order sum of the 16 partials.  The order never depends on timing => bit-identical on every rank.
#define WK_THREADS 1024
#ifndef DIAL_K4_FP64
#define DIAL_K4_FP64 0   // measurement switch: the softmax statistics in fp64 (see weights_kernel)
#endif
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int tid = threadIdx.x;
  float ws = dialwave::wave_sum_dpp(v);
  if ((tid & 63) == 0) red[tid >> 6] = ws;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int k = 0; k < WK_THREADS / 64; k++) r += red[k];
  __syncthreads();
  return r;
}
// the same in fp64 (the softmax's mean / variance / normaliser: see weights_kernel): xor-butterfly inside the wavefront (the same
// association in every lane), then the fixed-order sum of the 16 partials
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  const int tid = threadIdx.x;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int k = 0; k < WK_THREADS / 64; k++) r += red[k];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  const int tid = threadIdx.x;
  float wm = dialwave::wave_max_shfl(v);
  if ((tid & 63) == 0) red[tid >> 6] = wm;
  __syncthreads();
  float r = -INFINITY;
#pragma unroll
  for (int k = 0; k < WK_THREADS / 64; k++) r = red[k] > r ? red[k] : r;
  __syncthreads();
  return r;
}

// K4a (dial_core.py:121-128): rews [B] (last = mean trajectory) -> softmax weights [B].
// logp0 = (rews - rew_bar) / std(rews) / temp; std is the population std over all B samples.
// `gathered` != nullptr (sharded runs): the rewards arrive as the all-gather delivered them -- [world][per + 1]: every rank's noisy
// samples, then its copy of the mean-trajectory reward -- and are put into the order K4 wants ([n_total noisy | mean], written to
// `rews_out`) by this kernel's first pass: the packing launch of rounds 2-5 (dial_shard_pack_rewards) is gone from the iteration.
extern "C" __global__ void __launch_bounds__(WK_THREADS)
weights_kernel(const float* __restrict__ rews_in, int B, float temp, float* __restrict__ weights, const float* __restrict__ gathered,
               int per, float* __restrict__ rews_out) {
  __shared__ float red[WK_THREADS / 64];
  const int tid = threadIdx.x;
  const float* rews = rews_in;
  if (gathered) {
    const int n_total = B - 1;
    for (int n = tid; n < B; n += WK_THREADS)
      rews_out[n] = n < n_total ? gathered[(size_t)(n / per) * (per + 1) + (n % per)] : gathered[per];   // (rank 0's mean-trajectory reward: bit-identical on every rank)
    __threadfence_block();
    __syncthreads();
    rews = rews_out;
  }
#if DIAL_K4_FP64
  // (-DDIAL_K4_FP64=1, a measurement switch of round 6: mean, variance and normaliser in fp64, the logits from an fp64 1 / (std temp).
  //  Closer to the exact softmax of the same rewards -- and therefore FURTHER from the reference's fp32 arithmetic, which the oracle
  //  restates: the chained-plan replay test left its gate with it (acts 3.18e-3 off after five ticks vs the 3e-3 allowed).  What the
  //  peaked-softmax sensitivity needed was the mean rewards themselves correctly rounded: rollout_driver.h sums them in fp64.)
  __shared__ double redd[WK_THREADS / 64];
  double accd = 0.0;
  float mxr = -INFINITY;
  for (int n = tid; n < B; n += WK_THREADS) { float r = rews[n]; accd += (double)r; mxr = r > mxr ? r : mxr; }
  const double mean = block_sum_d(accd, redd) / (double)B;
  const float rmax = block_max(mxr, red);
  accd = 0.0;
  for (int n = tid; n < B; n += WK_THREADS) { double d = (double)rews[n] - mean; accd += d * d; }
  const double stdd = sqrt(block_sum_d(accd, redd) / (double)B);
  const float stdv = (float)stdd;
  const float rew_bar = rews[B - 1];
  const double inv = 1.0 / (stdd * (double)temp);
  const double mxd = ((double)rmax - (double)rew_bar) * inv;   // max of logp0: the map r -> logp0 is increasing
  accd = 0.0;
  for (int n = tid; n < B; n += WK_THREADS) { const float l = (float)(((double)rews[n] - (double)rew_bar) * inv - mxd); accd += (double)expf(l); }
  const float den = (float)block_sum_d(accd, redd);
#else   // the reference's own arithmetic: fp32 throughout (dial_core.py:126-128)
  float acc = 0.f, mxr = -INFINITY;
  for (int n = tid; n < B; n += WK_THREADS) { float r = rews[n]; acc += r; mxr = r > mxr ? r : mxr; }
  const float mean = block_sum(acc, red) / (float)B;
  const float rmax = block_max(mxr, red);
  acc = 0.f;
  for (int n = tid; n < B; n += WK_THREADS) { float d = rews[n] - mean; acc += d * d; }
  const float stdv = sqrtf(block_sum(acc, red) / (float)B);
  const float rew_bar = rews[B - 1];
  const float mx = (rmax - rew_bar) / stdv / temp;   // max of logp0: the map r -> logp0 is increasing
  acc = 0.f;
  for (int n = tid; n < B; n += WK_THREADS) { float l = (rews[n] - rew_bar) / stdv / temp; acc += expf(l - mx); }
  const float den = block_sum(acc, red);
#endif
  // std(rews) == 0 (all rewards identical): the reference divides 0 by 0 (dial_core.py:126) and every weight, hence
  // Ybar, becomes NaN.  Kept bug-compatible and DEFINED: the NaN is written as a bit pattern, because the device
  // code is compiled with -fno-honor-nans and a floating-point 0/0 would be undefined there.
  const bool degenerate = stdv == 0.f;
  uint32_t* wbits = reinterpret_cast<uint32_t*>(weights);
  for (int n = tid; n < B; n += WK_THREADS) {
#if DIAL_K4_FP64
    const float l = (float)(((double)rews[n] - (double)rew_bar) * inv - mxd);
    const float wv = expf(l) / den;
#else
    float l = (rews[n] - rew_bar) / stdv / temp;
    const float wv = expf(l - mx) / den;
#endif
    wbits[n] = degenerate ? 0x7fc00000u : __builtin_bit_cast(uint32_t, wv);
  }
}

// K4b (dial_core.py:132-135): out[c] = sum_n w[widx(n)] * X_seg[n][c] over the local samples.
struct WsumSeg { const float* X; float* out; int C; int c0; };
struct WsumArgs { WsumSeg seg[4]; int nseg, Ctot, n_rows, w_begin, mean_row, mean_widx; };

// Block = 64 columns x 4 wavefronts; wavefront g takes every 4th row of the block's row chunk, the four partial
// sums are combined in a fixed order through LDS (deterministic, bit-identical on every rank).
// Two launches.  (Round 6 tried ONE -- the block that draws the last ticket of its column block sums the chunks -- and measured the full
// iteration 95 us SLOWER at N = 2048: on this multi-XCD part a device-scope release / acquire fence inside a kernel writes back and
// invalidates the XCD's L2, 1408 blocks each paying for it; a kernel boundary does the same once.  profiles/r06_k4_fusion.txt)
extern "C" __global__ void __launch_bounds__(256)
wsum_partial_kernel(WsumArgs a, const float* __restrict__ weights, float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (c < a.Ctot) {
    int sg = 0;
    for (int k = 1; k < a.nseg; k++) if (c >= a.seg[k].c0) sg = k;
    const float* X = a.seg[sg].X;
    const int C = a.seg[sg].C, cl = c - a.seg[sg].c0;
    const int chunk = blockIdx.y, per = (a.n_rows + WSUM_CHUNKS - 1) / WSUM_CHUNKS;
    const int r0 = chunk * per, r1 = (r0 + per < a.n_rows) ? r0 + per : a.n_rows;
#pragma unroll 4
    for (int r = r0 + g; r < r1; r += 4) {
      // rows [0, mean_row) are this shard's noisy samples -> weight index w_begin + r; the mean row (if
      // present and included) uses mean_widx; a mean row that is not included has mean_widx < 0.
      int wi = (r == a.mean_row) ? a.mean_widx : a.w_begin + r;
      if (wi >= 0) acc += weights[wi] * X[(size_t)r * C + cl];
    }
  }
  red[g][lane] = acc;
  __syncthreads();
  if (g == 0 && c < a.Ctot) partial[(size_t)blockIdx.y * a.Ctot + c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}
extern "C" __global__ void __launch_bounds__(256)
wsum_final_kernel(WsumArgs a, const float* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.Ctot) return;
  int sg = 0;
  for (int k = 1; k < a.nseg; k++) if (c >= a.seg[k].c0) sg = k;
  float acc = 0.f;
#pragma unroll 16   // (sixteen independent loads in flight; the additions keep their order)
  for (int ch = 0; ch < WSUM_CHUNKS; ch++) acc += partial[(size_t)ch * a.Ctot + c];
  if (a.seg[sg].out) a.seg[sg].out[c - a.seg[sg].c0] = acc;
}

// Weighted mean of ALL candidate nodes, regenerated from the noise (dial_core.py:110-115,132), with a fixed reduction order
// (bit-identical on every rank).  Round 6: parallel over (row chunk, Philox quad) -- rounds 2-5 ran one 64-thread block
// per chunk with one thread per COLUMN, i.e. every Philox quad was drawn four times and a rank spent 70 us here at N = 8192 (0.3 ms at
// cfg 5's N = 65536: more than a third of its rollout launch).
//   block = 16 row lanes x 16 quads (quad = four consecutive columns = one Philox4x32 draw), grid = (ceil(quads / 16), YB_CHUNKS);
//   thread (r, q) adds rows r0 + r, r0 + r + 16, ... of its chunk for the quad's four columns; the 16 row lanes are combined in a
//   fixed order through LDS; a second launch sums the chunks (order 0, 1, ...; see wsum_partial_kernel for why not a ticket).
#define YB_CHUNKS 128
extern "C" __global__ void __launch_bounds__(256)
ybar_partial_kernel(const float* __restrict__ weights, const float* __restrict__ eps, const float* __restrict__ Ybar,
                    const float* __restrict__ noise_scale, int ns, int n_total, int C, int nu, float* __restrict__ partial,
                    uint32_t seed_lo, uint32_t seed_hi, uint32_t iter) {
  __shared__ float red[16][64];
  const int ql = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int q = blockIdx.x * 16 + ql, c0 = 4 * q;                 // this thread's quad: columns c0 .. c0 + 3
  const int chunk = blockIdx.y, per = (n_total + 1 + YB_CHUNKS - 1) / YB_CHUNKS;
  const int r0 = chunk * per, r1 = (r0 + per < n_total + 1) ? r0 + per : n_total + 1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, yb[4] = {0.f, 0.f, 0.f, 0.f}, y0[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f};
  int kk[4] = {0, 0, 0, 0};
  if (c0 < C) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int c = c0 + e < C ? c0 + e : C - 1, k = c / nu;
      kk[e] = k; yb[e] = Ybar[c]; y0[e] = Ybar[c - k * nu]; sc[e] = noise_scale[ns == 1 ? 0 : k];
    }
    for (int n = r0 + rl; n < r1; n += 16) {
      float z[4] = {0.f, 0.f, 0.f, 0.f};
      if (n < n_total) {
        if (eps) {
#pragma unroll
          for (int e = 0; e < 4; e++) z[e] = c0 + e < C ? eps[(size_t)n * C + c0 + e] : 0.f;
        } else {   // the sample's noise exactly as the rollout prologue drew it (Philox keyed by seed / iteration / sample / quad)
          dial::normal_quad((uint32_t)n, (uint32_t)q, iter, seed_lo, seed_hi, z);
        }
      }
      const float wn = weights[n];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float v = n < n_total ? (kk[e] == 0 ? y0[e] : z[e] * sc[e] + yb[e]) : yb[e];
        v = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
        acc[e] += wn * v;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) red[rl][4 * ql + e] = acc[e];
  __syncthreads();
  const int cl = threadIdx.x, c = blockIdx.x * 64 + cl;            // threads 0 .. 63: one column of the block each
  if (cl < 64 && c < C) {
    float t = 0.f;
    for (int r = 0; r < 16; r++) t += red[r][cl];
    partial[(size_t)chunk * C + c] = t;
  }
}
extern "C" __global__ void __launch_bounds__(64)
ybar_final_kernel(const float* __restrict__ partial, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
#pragma unroll 16
  for (int ch = 0; ch < YB_CHUNKS; ch++) acc += partial[(size_t)ch * C + c];
  out[c] = acc;
}

// Materialise the in-kernel noise: eps_out[n - n_begin, :] for samples n_begin .. n_begin + n_count
extern "C" __global__ void __launch_bounds__(256)
rng_fill_kernel(uint32_t seed_lo, uint32_t seed_hi, uint32_t iter, int n_begin, int n_count, int C, float* eps_out) {
  const int nq = (C + 3) / 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)n_count * nq) return;
  const int n = (int)(t / nq), q = (int)(t - (long long)n * nq);
  float z[4];
  dial::normal_quad((uint32_t)(n_begin + n), (uint32_t)q, iter, seed_lo, seed_hi, z);
  for (int e = 0; e < 4; e++)
    if (4 * q + e < C) eps_out[(size_t)n * C + 4 * q + e] = z[e];
}

// Sharded runs: the all-gather delivers [world][per + 1] mean rewards (every rank's noisy samples, then its copy of the
// mean-trajectory reward); K4 wants [n_total noisy | mean].  One tiny launch instead of slicing / concatenating tensors.
extern "C" __global__ void __launch_bounds__(256)
pack_rewards_kernel(const float* __restrict__ gathered, int per, int n_total, float* __restrict__ rews_all) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < n_total) rews_all[n] = gathered[(size_t)(n / per) * (per + 1) + (n % per)];
  else if (n == n_total) rews_all[n] = gathered[per];   // rank 0's mean-trajectory reward (bit-identical on every rank)
}

// K5 (dial_core.py:160-166): u = W Y; u = roll(u,-1); u[-1] = 0; Y = V u.  One small workgroup.
extern "C" __global__ void __launch_bounds__(64)
shift_kernel(const dial_cfg* __restrict__ cfg, int nu, float* Y) {
  __shared__ float u[DIAL_MAX_T * DIAL_MAX_U];
  const int T = cfg->Hsample + 1, Hn1 = cfg->Hnode + 1;
  for (int it = threadIdx.x; it < T * nu; it += 64) {
    const int st = it / nu, a = it - st * nu;
    float acc = 0.f;
    if (st + 1 < T)
      for (int k = 0; k < Hn1; k++) acc += cfg->W[st + 1][k] * Y[k * nu + a];
    u[it] = acc;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < Hn1 * nu; it += 64) {
    const int k = it / nu, a = it - k * nu;
    float acc = 0.f;
    for (int st = 0; st < T; st++) acc += cfg->V[k][st] * u[st * nu + a];
    Y[it] = acc;
  }
}

// wave primitive self-test: out[0] = DPP sum of lane ids, out[1] = shuffle sum, out[2] = max
extern "C" __global__ void __launch_bounds__(64) selftest_kernel(float* out) {
  float v = (float)threadIdx.x + 0.5f;
  float a = dialwave::wave_sum_dpp(v), b = dialwave::wave_sum_shfl(v), c = dialwave::wave_max_shfl(v);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; out[2] = c; }
}

// ------------------------------------------------------------------ host side
struct dial_ctx {
  int device = 0;
  dial_model hm;
  dial_task ht;
  dial_cfg hc;
  dial_derived hd;
  bool has_cfg = false;
  int inst = 0;               // 0 generic (DimsMax), 1 Go2, 2 H1, 3 H1 loco, 4 Allegro (elliptic cones), 5 Go2 crate climb, 6 H1 push crate
  void* dcm = nullptr;        // CModel<D> of the chosen instantiation (device)
  dial_task* dtask = nullptr;
  dial_cfg* dcfg = nullptr;
  int con_cap = 0, ovf_words = 0;   // generic instantiation: contact cap of the rollout kernel's LDS workspace, words of one overflow area
  float* ovf = nullptr;            // B_cap overflow areas
  int B_cap = 0, W_cap = 0, T = 0, Hn1 = 0, nx = 0;   // B_cap: rollouts this context can hold (local shard + mean), W_cap: global N + 1
  float *Y0s = nullptr, *rewss = nullptr, *rews = nullptr, *qss = nullptr, *qdss = nullptr, *xss = nullptr;
  float *weights = nullptr, *partial = nullptr;
  unsigned long long* prof = nullptr;
  size_t lds_bytes = 0;        // env_step / env_reset kernels (one wavefront, no node array)
  size_t lds_rollout = 0;      // rollout kernel: constants + DIAL_WPB workspaces
  size_t lds_large = 0;        // Go2 large-batch instantiation (more wavefronts per workgroup)
  int ws_words = 0, cm_bytes = 0, wpb = 1;
  int* next = nullptr;         // rollout queue head (batches larger than the chip keeps resident)
  float* relay_buf = nullptr;  // mean-trajectory relay: state handed from piece to piece, and the turn flag
  float* slice_buf = nullptr;  // time-sliced rollout queue (rollout_kernel.h): one hand-over slot and one turn flag per rollout
  int* slice_flag = nullptr;
  int slice_cap = 0, slice_stride = 0, slice_steps = 3;
  int* work_stat = nullptr;    // lag-based issue priority (wave.h): the launch's running totals (solver iterations, control steps)
  int* relay_flag = nullptr;
  int* err_host = nullptr;     // sticky error word: pinned host memory the kernels can write (relay time-out) ...
  int* err_dev = nullptr;      // ... and its device-side address
  bool relay_ok = false, relay_always = false;
  bool no_spread = false;        // options: batches below the large-batch kernel's resident set fill workgroup after workgroup
  bool no_mean_inline = false;   // options: the mean trajectory of a large Go2 batch as an ordinary queue item
  // Allegro split launch (see DIAL_ALLEGRO_WPB_EVEN)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  size_t lds_even = 0, lds_one = 0;
  bool split_ok = false;
  int wpb_even = 0;            // wavefronts per workgroup of the even launch (8 x CUs rollouts in all)
  int n_simd = 0;              // SIMDs of the device (4 per CU)
  int debug_stall_piece1 = 0;  // DIAL_DEBUG_RELAY_STALL=k (tests): relay piece k - 1 never hands over
  int relay_steps = 3;         // control steps per relay piece (measured: 1 -> no gain, 2 -4.9 %, 3 -5.3 %, 4 -5.0 %, 6 -4.0 %)
  int resident_blocks = 0, resident_blocks_large = 0;   // workgroups of the rollout kernel the whole chip holds at once
  // Go2, two rollouts per wavefront (rollout_kernel2): LDS of the one-wavefront / DIAL_GO2_PAIR_WPB-wavefront workgroups and how
  // many of each the chip keeps resident; pair_ok: the context launches them (dial_options::pair_mode)
  bool pair_ok = false;
  size_t lds_pair = 0, lds_pair_large = 0;
  int resident_pair = 0, resident_pair_large = 0;
  bool timing = false;
  dial_options opt{};          // launch-shape / measurement options (dial_create_ex); all zero = shipped behaviour
  float* trace = nullptr;      // diagnostics: per-step packed states of the rollouts (dial_set_state_trace), caller-owned
  int trace_rows = 0;
  int ovf_slots = 0;           // overflow areas allocated (>= the largest grid this context ever launches)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  std::string err;
};

static thread_local std::string g_err;   // last error of calls that have no context (per calling thread)

static int fail(dial_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  g_err = msg;
  return code;
}
#define HIP_TRY(ctx, expr)                                                                     \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(ctx, DIAL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

extern "C" {

const char* dial_last_error(const dial_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int dial_abi_sizes(int* a, int* b, int* c) {
  if (a) *a = (int)sizeof(dial_model);
  if (b) *b = (int)sizeof(dial_task);
  if (c) *c = (int)sizeof(dial_cfg);
  return DIAL_OK;
}

void dial_destroy(dial_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  void* ptrs[] = {ctx->dcm, ctx->dtask, ctx->dcfg, ctx->prof, ctx->next, ctx->relay_buf, ctx->relay_flag, ctx->Y0s, ctx->rewss, ctx->rews, ctx->qss,
                  ctx->qdss,   ctx->xss,   ctx->weights, ctx->partial, ctx->ovf, ctx->slice_buf, ctx->slice_flag, ctx->work_stat};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& ev : ctx->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->side) (void)hipStreamDestroy(ctx->side);
  if (ctx->err_host) (void)hipHostFree(ctx->err_host);
  delete ctx;
}

int dial_create(dial_ctx** out, const dial_model* model, const dial_task* task, const dial_cfg* cfg, int device) {
  return dial_create_ex(out, model, task, cfg, device, -1, nullptr);
}

int dial_create_sharded(dial_ctx** out, const dial_model* model, const dial_task* task, const dial_cfg* cfg, int device,
                        int n_local_cap) {
  if (cfg && (n_local_cap < 0 || n_local_cap > cfg->Nsample)) return fail(nullptr, DIAL_ERR_ARG, "dial_create_sharded: n_local_cap must be in [0, Nsample]");
  return dial_create_ex(out, model, task, cfg, device, n_local_cap, nullptr);
}

int dial_create_ex(dial_ctx** out, const dial_model* model, const dial_task* task, const dial_cfg* cfg, int device,
                   int n_local_cap, const dial_options* opts) {
  if (!out || !model || !task) return fail(nullptr, DIAL_ERR_ARG, "dial_create: null argument");
  if (n_local_cap < 0) n_local_cap = cfg ? cfg->Nsample : 0;
  if (cfg && n_local_cap > cfg->Nsample) return fail(nullptr, DIAL_ERR_ARG, "dial_create_ex: n_local_cap must be in [0, Nsample]");
  const dial_options opt = opts ? *opts : dial_options{};
  if (opt.relay_steps < 0 || opt.relay_steps > 16) return fail(nullptr, DIAL_ERR_ARG, "dial_create_ex: options.relay_steps must be in 0 .. 16");
  if (opt.slice_steps < 0 || opt.slice_steps > 16) return fail(nullptr, DIAL_ERR_ARG, "dial_create_ex: options.slice_steps must be in 0 .. 16");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, DIAL_ERR_HIP, "dial_create: no HIP device available (the HIP path has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, DIAL_ERR_ARG, "dial_create: bad device index");
  if (task->kind < DIAL_TASK_GO2_WALK || task->kind > DIAL_TASK_H1_PUSH_CRATE)
    return fail(nullptr, DIAL_ERR_UNSUPPORTED, "dial_create: unknown task kind");
  {   // the reward phase unrolls over the feet of the task's robot (rollout_body.h: reward_phase)
    const bool go2 = task->kind == DIAL_TASK_GO2_WALK || task->kind == DIAL_TASK_GO2_SEQ_JUMP || task->kind == DIAL_TASK_GO2_CRATE;
    const bool h1 = task->kind == DIAL_TASK_H1_WALK || task->kind == DIAL_TASK_H1_LOCO || task->kind == DIAL_TASK_H1_PUSH_CRATE;
    if ((go2 && task->nfeet != 4) || (h1 && task->nfeet != 2))
      return fail(nullptr, DIAL_ERR_ARG, "dial_create: task.nfeet must be 4 for the Go2 tasks and 2 for the H1 tasks");
  }
  if (task->kind == DIAL_TASK_GO2_CRATE)
    for (int f = 0; f < 4; f++)
      if (task->crate_contact[f] < 0 || task->crate_contact[f] >= model->ncon)
        return fail(nullptr, DIAL_ERR_ARG, "dial_create: task.crate_contact must index the model's contact list");
  if (task->kind == DIAL_TASK_H1_PUSH_CRATE) {
    bool ok = task->pc_n_unwanted >= 0 && task->pc_n_unwanted <= 16;
    for (int f = 0; f < 2 && ok; f++) {
      ok = task->pc_wanted[f] >= 0 && task->pc_wanted[f] < model->ncon;
      for (int k = 0; k < 2; k++) ok = ok && task->pc_foot_contact[f][k] >= 0 && task->pc_foot_contact[f][k] < model->ncon;
    }
    for (int q = 0; ok && q < task->pc_n_unwanted; q++) ok = task->pc_unwanted[q] >= 0 && task->pc_unwanted[q] < model->ncon;
    if (!ok) return fail(nullptr, DIAL_ERR_ARG, "dial_create: the push-crate task's contact indices must index the model's contact list");
  }
  if (model->nfri < 0 || model->nfri > DIAL_MAX_FRI || (model->nfri > 0 && model->cone != DIAL_CONE_PYRAMIDAL))
    return fail(nullptr, DIAL_ERR_UNSUPPORTED, "dial_create: dry friction (frictionloss) is supported for pyramidal models with <= DIAL_MAX_FRI such dofs");
  {   // box-box candidates keep their clipping polygons in LDS, in slices of the dead velocity temporaries cdofdot | cacc | cfl
      // (box_collide.h: box_box, rollout_body.h: forward; 6 nv + 12 nbody words, derived.h: ws_carve)
    int nbb = 0;
    for (int c = 0; c < model->ncon; c++) nbb += model->con_kind[c] == DIAL_CON_BOX_BOX ? 1 : 0;
    if (nbb * DIAL_BOX_POLY_WORDS > 6 * model->nv + 12 * model->nbody)
      return fail(nullptr, DIAL_ERR_UNSUPPORTED, "dial_create: too many box-box candidate contacts for the polygon scratch");
  }
  // The box narrow phases park a candidate whose shapes are provably more than DIAL_BOX_PARK_DIST apart (box_collide.h: second
  // broad phases) -- with a lower BOUND of the distance, a centre-line frame and a midpoint position.  A contact is active while
  // dist < margin, so a box contact whose margin reaches that distance would be activated with those placeholder values.
  for (int c = 0; c < model->ncon; c++)
    if (model->con_kind[c] >= DIAL_CON_PLANE_BOX && !(model->con_margin[c] < DIAL_BOX_PARK_DIST))
      return fail(nullptr, DIAL_ERR_UNSUPPORTED, "dial_create: box contacts need margin < 0.01 m (the box narrow phases park candidates that are further apart)");
  if (model->cone != DIAL_CONE_PYRAMIDAL && model->cone != DIAL_CONE_ELLIPTIC)
    return fail(nullptr, DIAL_ERR_UNSUPPORTED, "dial_create: unknown friction cone type");
  if (model->ls_rule != DIAL_LS_SWAP && model->ls_rule != DIAL_LS_IN_BRACKET)
    return fail(nullptr, DIAL_ERR_ARG, "dial_create: unknown line-search rule");
  // pyramidal models: explicit Euler damping off; elliptic models run on their dimension-specialised instantiation only
  // (checked below)
  if (model->cone == DIAL_CONE_PYRAMIDAL && model->eulerdamp)
    return fail(nullptr, DIAL_ERR_UNSUPPORTED, "dial_create: eulerdamp is only supported on the elliptic-cone instantiations");
  if (model->nefc > DIAL_MAX_EFC || model->nv > DIAL_MAX_V ||
      model->nq + 2 * model->nv + DIAL_INFO_N > 4096)
    return fail(nullptr, DIAL_ERR_ARG, "dial_create: model exceeds kernel capacities");
  dial_ctx* ctx = new dial_ctx();
  ctx->device = device;
  ctx->opt = opt;
  ctx->hm = *model;
  ctx->ht = *task;
  int rc = dial_build_derived(model, &ctx->hd);
  if (rc != DIAL_OK) { delete ctx; return fail(nullptr, rc, "dial_create: unsupported model topology"); }
  ctx->nx = (model->nbody - 1) * 3;
  // from here on every failure path releases the half-built context (dial_destroy frees whatever was allocated)
#define HIP_TRY_CREATE(expr)                                                                          \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      dial_destroy(ctx);                                                                              \
      return fail(nullptr, DIAL_ERR_HIP, std::string("dial_create: " #expr ": ") + hipGetErrorString(e_)); \
    }                                                                                                 \
  } while (0)
  HIP_TRY_CREATE(hipSetDevice(device));
  {
    // pick the kernel instantiation and upload its constants
    auto upload = [&](auto tag) -> int {
      using D = decltype(tag);
      CModel<D>* h = new CModel<D>();
      fill_cmodel(h, model, task, &ctx->hd);
      Ws s;
      const int nnode = cfg ? cfg->Hnode + 1 : 0;
      const int ws0 = ws_carve(s, (float*)0, model->nq, model->nv, model->nu, model->nbody, model->njnt,
                               model->ngeom, model->nsite, model->ncon, model->nefc, 0, dial::kNeedL<D>, D::square,
                               D::ell ? D::JCW : 0, 0, D::NVP);
      ctx->cm_bytes = D::is_static ? (int)(((sizeof(CModel<D>) + 15) / 16) * 16) : 0;
      if constexpr (D::gen) {
        // many candidate contacts, few of which touch (crate scene: 52 / 4-8): the rollout kernel's LDS workspace is sized for
        // DIAL_CON_CAP touching contacts, samples beyond run on an overflow area in global memory.  Default 14: for the crate
        // scene that is 17.4 KB per wavefront -- NINE per CU, so that the 2049 rollouts of its example are resident at once
        // (at 16 the workspace is 18.3 KB, eight per CU = 2048 slots, and the 2049th rollout waits for a whole round)
        const bool given = opt.con_cap != 0;   // options.con_cap: 0 chosen here, > 0 given, < 0 no cap
        int cap = given ? opt.con_cap : 14;
        if (cfg && model->cone == DIAL_CONE_PYRAMIDAL && cap > 0 && model->ncon > cap) {
          // unless the cap was given: the largest one in 20 .. 4 with which NINE wavefronts fit a CU (8 x 256 + 1 rollouts of
          // the examples' N = 2048 resident at once).  Round 4, the scenes' own instantiations (nine wavefronts share one staged
          // copy of the constants): crate climb 18 (16.7 KB per wavefront), push crate 8 (16.8 KB); capacity-dimension
          // instantiation: 16 (17.2 KB) / 9 (17.1 KB).
          // (workgroups of wpb wavefronts that share one staged copy of the constants: 9 / wpb workgroups per CU)
          if (!given) {
#ifdef DIAL_PROFILE
            const size_t prof_bytes = 16 + (size_t)ctx->wpb * 32 * sizeof(unsigned long long);
#else
            const size_t prof_bytes = 0;
#endif
            const size_t budget = ctx->wpb > 1 ? ((size_t)160 * 1024 / (9 / ctx->wpb) - ctx->cm_bytes - prof_bytes) / ctx->wpb / 16 * 16
                                               : (size_t)(160 * 1024) / 9 / 512 * 512;
            for (int c = 20; c >= 4; c--) {
              Ws st;
              const int wds = ws_carve(st, (float*)0, model->nq, model->nv, model->nu, model->nbody, model->njnt, model->ngeom, model->nsite,
                                       model->ncon, model->nefc, nnode, dial::kNeedL<D>, D::square, 0, c, D::NVP);
              if ((size_t)wds * sizeof(float) <= budget) { cap = c; break; }
              if (c == 4) cap = 4;   // (nothing fits nine: the smallest cap; dial_create then fails on the LDS check if that is too much)
            }
          }
          ctx->con_cap = cap;
          Ws so;
          ctx->ovf_words = ws_overflow(so, (float*)0, model->nv, model->ncon, model->nefc);
        }
      }
      ctx->ws_words = ws_carve(s, (float*)0, model->nq, model->nv, model->nu, model->nbody, model->njnt,
                               model->ngeom, model->nsite, model->ncon, model->nefc, nnode, dial::kNeedL<D>, D::square,
                               D::ell ? D::JCW : 0, ctx->con_cap, D::NVP, (D::pre_ctrl && cfg) ? cfg->Hsample + 1 : 0);
#ifdef DIAL_WS_SKEW
      // measurement switch (round 6, LDS bank conflicts of the two-rollouts-per-wavefront kernels): the workspaces of a workgroup
      // start DIAL_WS_SKEW words past a multiple of 32 words (= the 32 LDS banks) from each other
      ctx->ws_words = (ctx->ws_words + 31) / 32 * 32 + DIAL_WS_SKEW;
#endif
      ctx->lds_bytes = ctx->cm_bytes + (size_t)ws0 * sizeof(float);
      ctx->lds_rollout = ctx->cm_bytes + (size_t)ctx->wpb * ctx->ws_words * sizeof(float);
#ifdef DIAL_PROFILE
      ctx->lds_rollout += 16 + (size_t)ctx->wpb * 32 * sizeof(unsigned long long);
#endif
      hipError_t e = hipMalloc(&ctx->dcm, (sizeof(CModel<D>) + 15) / 16 * 16);   // (the kernels stage it with 16-byte copies)
      if (e == hipSuccess) e = hipMemcpy(ctx->dcm, h, sizeof(CModel<D>), hipMemcpyHostToDevice);
      delete h;
      return e == hipSuccess ? DIAL_OK : DIAL_ERR_HIP;
    };
    int urc;
    const auto kind_ok = [&](uint32_t mask) { return ((mask >> task->kind) & 1u) != 0; };   // the robot's own task kinds only
    // the robot's own dimension-specialised instantiation (default) -- unless the model has more distinct (solref, solimp) rows than
    // their impedance table holds (CModel::kbi_tab; the shipped robots have two or three)
    const bool kbi_ok = kbi_unique_rows(model) <= DIAL_KBI_ROWS;
    const bool own = !opt.force_generic && kbi_ok;
    // (Dims::pre_ctrl instantiations run ONE physics step per control step -- every shipped Go2 task; another ratio: the capacity-dimension kernel)
    if (own && dims_match<DimsGo2>(model) && derived_fits<DimsGo2>(&ctx->hd) && kind_ok(dial::task_kind_mask<DimsGo2>()) &&
        (!DimsGo2::pre_ctrl || task->n_frames == 1)) { ctx->inst = 1; ctx->wpb = 1; urc = upload(DimsGo2{}); }
    else if (own && dims_match<DimsH1>(model) && derived_fits<DimsH1>(&ctx->hd) && kind_ok(dial::task_kind_mask<DimsH1>())) { ctx->inst = 2; ctx->wpb = 3; urc = upload(DimsH1{}); }
    else if (own && dims_match<DimsH1Loco>(model) && derived_fits<DimsH1Loco>(&ctx->hd) && kind_ok(dial::task_kind_mask<DimsH1Loco>())) { ctx->inst = 3; ctx->wpb = 2; urc = upload(DimsH1Loco{}); }
    else if (model->cone == DIAL_CONE_ELLIPTIC) {
      if (!(kbi_ok && dims_match<DimsAllegro>(model) && ell_fits<DimsAllegro>(model, &ctx->hd))) {
        dial_destroy(ctx);
        return fail(nullptr, DIAL_ERR_UNSUPPORTED, "dial_create: elliptic-cone models need a dimension-specialised instantiation (built: Allegro hand)");
      }
      ctx->inst = 4; ctx->wpb = DIAL_ALLEGRO_WPB; urc = upload(DimsAllegro{});
    }
    // (options.con_cap < 0 -- no cap, the full-size workspace -- does not fit nine wavefronts per workgroup: capacity-dimension kernel)
    else if (own && opt.con_cap >= 0 && dims_match<DimsGo2Crate>(model) && kind_ok(dial::task_kind_mask<DimsGo2Crate>())) { ctx->inst = 5; ctx->wpb = DIAL_CRATE_WPB; urc = upload(DimsGo2Crate{}); }
    else if (own && opt.con_cap >= 0 && dims_match<DimsH1PushCrate>(model) && kind_ok(dial::task_kind_mask<DimsH1PushCrate>())) { ctx->inst = 6; ctx->wpb = DIAL_CRATE_WPB; urc = upload(DimsH1PushCrate{}); }
    else { ctx->inst = 0; ctx->wpb = 1; urc = upload(DimsMax{}); }
    if (urc != DIAL_OK) { dial_destroy(ctx); return fail(nullptr, urc, "dial_create: uploading the model constants failed"); }
  }
  if (!cfg) ctx->lds_rollout = ctx->lds_bytes;   // no cfg: env.step / env.reset only -- no rollout launch, no capped workspace to size
  if (ctx->lds_rollout > 160 * 1024 || ctx->lds_bytes > 64 * 1024) {
    dial_destroy(ctx);
    return fail(nullptr, DIAL_ERR_ARG, "dial_create: LDS workspace exceeds the 160 KiB of a CU");
  }
  if (ctx->inst == 1) ctx->lds_large = ctx->cm_bytes + (size_t)DIAL_GO2_WPB_LARGE * ctx->ws_words * sizeof(float);
  {   // more than the default 64 KiB dynamic-LDS limit of a launch: opt in (gfx950: up to 160 KiB per workgroup)
    hipError_t e = hipSuccess;
    if (ctx->inst == 4 && ctx->lds_rollout > 64 * 1024)
      e = hipFuncSetAttribute((const void*)rollout_kernel<DimsAllegro, DIAL_ALLEGRO_WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_rollout);
    if (e == hipSuccess && ctx->inst == 4 && ctx->lds_rollout > 64 * 1024)
      e = hipFuncSetAttribute((const void*)rollout_kernel<DimsAllegro, DIAL_ALLEGRO_WPB, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_rollout);
    if (e == hipSuccess && ctx->inst == 4 && ctx->lds_rollout > 64 * 1024)
      e = hipFuncSetAttribute((const void*)rollout_kernel<DimsAllegro, DIAL_ALLEGRO_WPB, 3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_rollout);
#define DIAL_BIG_LDS(D)                                                                                                                         \
    if (e == hipSuccess && ctx->lds_rollout > 64 * 1024) {                                                                                     \
      e = hipFuncSetAttribute((const void*)rollout_kernel<D, DIAL_CRATE_WPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_rollout); \
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rollout_kernel<D, DIAL_CRATE_WPB, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_rollout); \
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rollout_kernel<D, DIAL_CRATE_WPB, 3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_rollout); \
    }
    if (ctx->inst == 5) { DIAL_BIG_LDS(DimsGo2Crate) }
    if (ctx->inst == 6) { DIAL_BIG_LDS(DimsH1PushCrate) }
#undef DIAL_BIG_LDS
    if (e == hipSuccess && ctx->inst == 1 && ctx->lds_large > 64 * 1024)
      e = hipFuncSetAttribute((const void*)rollout_kernel<DimsGo2, DIAL_GO2_WPB_LARGE, DIAL_GO2_OCC_LARGE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_large);
    if (ctx->inst == 1 && cfg) {
      ctx->pair_ok = opt.pair_mode != 1;
      ctx->lds_pair = ctx->cm_bytes + (size_t)2 * ctx->ws_words * sizeof(float);
      ctx->lds_pair_large = ctx->cm_bytes + (size_t)2 * DIAL_GO2_PAIR_WPB * ctx->ws_words * sizeof(float);
#ifdef DIAL_PROFILE
      ctx->lds_pair += 16 + 32 * sizeof(unsigned long long);
      ctx->lds_pair_large += 16 + (size_t)DIAL_GO2_PAIR_WPB * 32 * sizeof(unsigned long long);
#endif
      if (ctx->lds_pair_large > 160 * 1024) ctx->pair_ok = false;
      if (e == hipSuccess && ctx->pair_ok && ctx->lds_pair_large > 64 * 1024)
        e = hipFuncSetAttribute((const void*)rollout_kernel2<DimsGo2, DIAL_GO2_PAIR_WPB, DIAL_GO2_PAIR_OCC, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_pair_large);
    }
    if (e != hipSuccess) { dial_destroy(ctx); return fail(nullptr, DIAL_ERR_HIP, std::string("dial_create: hipFuncSetAttribute: ") + hipGetErrorString(e)); }
  }
  {   // how many workgroups of the rollout kernel the chip keeps resident (larger batches go through the rollout queue)
    hipDeviceProp_t prop;
    HIP_TRY_CREATE(hipGetDeviceProperties(&prop, device));
    int nb = 0;
    hipError_t e = hipSuccess;
#define DIAL_RESIDENT(D, WPB) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rollout_kernel<D, WPB>, 64 * WPB, ctx->lds_rollout)
    if (ctx->inst == 1) DIAL_RESIDENT(DimsGo2, 1);
    else if (ctx->inst == 2) DIAL_RESIDENT(DimsH1, 3);
    else if (ctx->inst == 3) DIAL_RESIDENT(DimsH1Loco, 2);
    else if (ctx->inst == 4) DIAL_RESIDENT(DimsAllegro, DIAL_ALLEGRO_WPB);
    else if (ctx->inst == 5) DIAL_RESIDENT(DimsGo2Crate, DIAL_CRATE_WPB);
    else if (ctx->inst == 6) DIAL_RESIDENT(DimsH1PushCrate, DIAL_CRATE_WPB);
    else DIAL_RESIDENT(DimsMax, 1);
#undef DIAL_RESIDENT
    ctx->n_simd = 4 * prop.multiProcessorCount;
    if (e == hipSuccess) ctx->resident_blocks = nb * prop.multiProcessorCount;
    if (e == hipSuccess && ctx->inst == 1) {
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rollout_kernel<DimsGo2, DIAL_GO2_WPB_LARGE, DIAL_GO2_OCC_LARGE, true>,
                                                       64 * DIAL_GO2_WPB_LARGE, ctx->lds_large);
      if (e == hipSuccess) ctx->resident_blocks_large = nb * prop.multiProcessorCount;
    }
    if (e == hipSuccess && ctx->pair_ok) {
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rollout_kernel2<DimsGo2, 1, DIAL_GO2_PAIR_OCC, false, false>, 64, ctx->lds_pair);
      if (e == hipSuccess) ctx->resident_pair = nb * prop.multiProcessorCount;
      if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rollout_kernel2<DimsGo2, DIAL_GO2_PAIR_WPB, DIAL_GO2_PAIR_OCC, true, true>,
                                                                            64 * DIAL_GO2_PAIR_WPB, ctx->lds_pair_large);
      if (e == hipSuccess) ctx->resident_pair_large = nb * prop.multiProcessorCount;
      if (e == hipSuccess && (ctx->resident_pair <= 0 || ctx->resident_pair_large <= 0)) ctx->pair_ok = false;
    }
    if (e != hipSuccess) { dial_destroy(ctx); return fail(nullptr, DIAL_ERR_HIP, std::string("dial_create: occupancy query: ") + hipGetErrorString(e)); }
    if (opt.no_queue) ctx->resident_blocks = ctx->resident_blocks_large = 0;   // options: one wavefront per rollout at any batch size
    HIP_TRY_CREATE(hipMalloc(&ctx->next, sizeof(int)));
    HIP_TRY_CREATE(hipMalloc(&ctx->relay_buf, sizeof(float) * (DIAL_MAX_Q + 2 * DIAL_MAX_V + DIAL_INFO_N + 4)));
    HIP_TRY_CREATE(hipMalloc(&ctx->relay_flag, sizeof(int)));
    HIP_TRY_CREATE(hipMemset(ctx->relay_flag, 0, sizeof(int)));
    HIP_TRY_CREATE(hipHostMalloc((void**)&ctx->err_host, sizeof(int), hipHostMallocMapped));
    *ctx->err_host = 0;
    HIP_TRY_CREATE(hipHostGetDevicePointer((void**)&ctx->err_dev, ctx->err_host, 0));
    ctx->relay_ok = ctx->wpb == 1 && !opt.no_relay;
    ctx->relay_always = opt.relay_always != 0;
    ctx->no_mean_inline = opt.no_mean_inline != 0;
    ctx->no_spread = opt.no_spread != 0;
    ctx->wpb_even = ctx->inst == 4 ? DIAL_ALLEGRO_WPB_EVEN : ctx->inst == 2 ? DIAL_H1_WPB_EVEN : 0;
    if ((opt.no_split_mask >> ctx->inst) & 1) ctx->wpb_even = 0;
    if (ctx->wpb_even > 0) {
      ctx->lds_even = ctx->cm_bytes + (size_t)ctx->wpb_even * ctx->ws_words * sizeof(float);
      ctx->lds_one = ctx->cm_bytes + (size_t)ctx->ws_words * sizeof(float);
#ifdef DIAL_PROFILE
      ctx->lds_even += 16 + (size_t)ctx->wpb_even * 32 * sizeof(unsigned long long);
      ctx->lds_one += 16 + 32 * sizeof(unsigned long long);
#endif
      // (8 / wpb_even) even workgroups and the one-wavefront workgroup on one CU: 160 KiB of LDS
      if ((8 / ctx->wpb_even) * ctx->lds_even + ctx->lds_one <= 160 * 1024) {
        hipError_t e2 = hipSuccess;
        if (ctx->lds_even > 64 * 1024) {
          if (ctx->inst == 4) e2 = hipFuncSetAttribute((const void*)rollout_kernel<DimsAllegro, DIAL_ALLEGRO_WPB_EVEN, DIAL_EVEN_OCC_ALLEGRO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_even);
          else e2 = hipFuncSetAttribute((const void*)rollout_kernel<DimsH1, DIAL_H1_WPB_EVEN, DIAL_EVEN_OCC_H1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_even);
        }
        if (e2 == hipSuccess) e2 = hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking);
        if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
        if (e2 == hipSuccess) e2 = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
        if (e2 != hipSuccess) { dial_destroy(ctx); return fail(nullptr, DIAL_ERR_HIP, std::string("dial_create: split-launch set-up: ") + hipGetErrorString(e2)); }
        ctx->split_ok = true;
      }
    }
    if (opt.relay_steps >= 1) ctx->relay_steps = opt.relay_steps;
    ctx->debug_stall_piece1 = opt.debug_relay_stall;
  }
  HIP_TRY_CREATE(hipMalloc(&ctx->dtask, sizeof(dial_task)));
  HIP_TRY_CREATE(hipMemcpy(ctx->dtask, task, sizeof(dial_task), hipMemcpyHostToDevice));
  if (cfg) {
    if (cfg->Hsample + 1 > DIAL_MAX_T || cfg->Hnode + 1 > DIAL_MAX_NODE || cfg->Hnode < 2 || cfg->Nsample < 1) {
      dial_destroy(ctx);
      return fail(nullptr, DIAL_ERR_ARG, "dial_create: Hsample/Hnode/Nsample out of range");
    }
    ctx->hc = *cfg;
    ctx->has_cfg = true;
    ctx->B_cap = n_local_cap + 1;
    ctx->W_cap = cfg->Nsample + 1;
    // one overflow area per wavefront slot of the LARGEST grid this context launches: B_cap rollouts, or B_cap - 1 noisy
    // ones + the relay pieces of the mean trajectory (at most Hsample + 1 of them) -- launch_rollout checks the grid
    ctx->ovf_slots = ctx->B_cap + (cfg->Hsample + 1 > 16 ? cfg->Hsample + 1 : 16);   // (and a last workgroup of up to 9 wavefronts)
    if (ctx->con_cap > 0) HIP_TRY_CREATE(hipMalloc(&ctx->ovf, (size_t)ctx->ovf_slots * ctx->ovf_words * sizeof(float)));
    ctx->T = cfg->Hsample + 1;
    ctx->Hn1 = cfg->Hnode + 1;
    if (ctx->inst == 4 && !opt.no_lag_priority) HIP_TRY_CREATE(hipMalloc(&ctx->work_stat, 2 * sizeof(int)));
    // time-sliced queue: models whose rollouts differ in length (the elliptic solver runs to convergence), batches beyond the
    // resident set.  (Not for the Go2's large batches: one hand-over per rollout and piece through global memory -- 4096 release /
    // acquire pairs per piece level -- cost more than the lone last rollout it would hide: N = 4096 1.34 vs 0.80 ms,
    // profiles/r04_ab_call_v_sliced_go2_k2_crate_quad.txt; those batches interleave the mean trajectory instead, see launch_rollout.)
    if (ctx->inst == 4 && !opt.no_slice && ctx->resident_blocks > 0 && ctx->B_cap > ctx->resident_blocks * ctx->wpb) {
      ctx->slice_stride = model->nq + 2 * model->nv + DIAL_INFO_N + 4;
      ctx->slice_cap = ctx->B_cap;
      if (opt.slice_steps >= 1) ctx->slice_steps = opt.slice_steps;
      HIP_TRY_CREATE(hipMalloc(&ctx->slice_buf, sizeof(float) * (size_t)ctx->slice_stride * ctx->B_cap));
      HIP_TRY_CREATE(hipMalloc(&ctx->slice_flag, sizeof(int) * (size_t)ctx->B_cap));
      HIP_TRY_CREATE(hipMemset(ctx->slice_flag, 0, sizeof(int) * (size_t)ctx->B_cap));
    }
    const size_t B = ctx->B_cap, T = ctx->T;
    HIP_TRY_CREATE(hipMalloc(&ctx->dcfg, sizeof(dial_cfg)));
    HIP_TRY_CREATE(hipMemcpy(ctx->dcfg, cfg, sizeof(dial_cfg), hipMemcpyHostToDevice));
    HIP_TRY_CREATE(hipMalloc(&ctx->Y0s, sizeof(float) * B * ctx->Hn1 * model->nu));
    HIP_TRY_CREATE(hipMalloc(&ctx->rewss, sizeof(float) * B * T));
    HIP_TRY_CREATE(hipMalloc(&ctx->rews, sizeof(float) * ctx->W_cap));
    HIP_TRY_CREATE(hipMalloc(&ctx->qss, sizeof(float) * B * T * model->nq));
    HIP_TRY_CREATE(hipMalloc(&ctx->qdss, sizeof(float) * B * T * model->nv));
    HIP_TRY_CREATE(hipMalloc(&ctx->xss, sizeof(float) * B * T * ctx->nx));
    HIP_TRY_CREATE(hipMalloc(&ctx->weights, sizeof(float) * ctx->W_cap));
    const size_t Ctot = (size_t)ctx->Hn1 * model->nu + T * (model->nq + model->nv + ctx->nx);
    // K4b's partial sums: [chunk][Ctot] (wsum_partial_kernel) or the mean-action kernel's [YB_CHUNKS][C] (ybar_partial_kernel), whichever is larger
    const size_t part_w = ((Ctot + 63) / 64) * (size_t)WSUM_CHUNKS * 64, part_y = (((size_t)ctx->Hn1 * model->nu + 63) / 64) * (size_t)YB_CHUNKS_HOST * 64;
    HIP_TRY_CREATE(hipMalloc(&ctx->partial, sizeof(float) * (part_w > part_y ? part_w : part_y)));
    // 32 section / event counters + (profile builds) a start / end timestamp per rollout of the last launch
    HIP_TRY_CREATE(hipMalloc(&ctx->prof, sizeof(unsigned long long) * (32 + 6 * B)));
    HIP_TRY_CREATE(hipMemset(ctx->prof, 0, sizeof(unsigned long long) * (32 + 6 * B)));
  }
#undef HIP_TRY_CREATE
  *out = ctx;
  return DIAL_OK;
}

// Sticky asynchronous error (today: a relay piece that never got its turn).  The word lives in pinned host memory, so
// looking at it costs nothing and needs no synchronisation; once it is seen the device is drained, the relay's turn flag
// is re-armed (a late predecessor may have left it non-zero) and the error is reported ONCE.
static int check_sticky(dial_ctx* ctx) {
  if (!ctx->err_host || __atomic_load_n(ctx->err_host, __ATOMIC_ACQUIRE) == 0) return DIAL_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  if (ctx->relay_flag) (void)hipMemset(ctx->relay_flag, 0, sizeof(int));
  if (ctx->slice_flag) (void)hipMemset(ctx->slice_flag, 0, sizeof(int) * (size_t)ctx->slice_cap);
  __atomic_store_n(ctx->err_host, 0, __ATOMIC_RELEASE);
  return fail(ctx, DIAL_ERR_HIP, "an earlier rollout launch gave up: a piece of the mean-trajectory relay never got its turn "
                                 "(results of that launch and of the launches queued behind it are invalid)");
}

int dial_status(dial_ctx* ctx) {
  if (!ctx) return DIAL_ERR_ARG;
  return check_sticky(ctx);
}

int dial_set_state_trace(dial_ctx* ctx, float* trace, int rows) {
  if (!ctx || (trace && rows < 1)) return fail(ctx, DIAL_ERR_ARG, "dial_set_state_trace: bad argument");
  ctx->trace = trace;
  ctx->trace_rows = trace ? rows : 0;
  return DIAL_OK;
}

int dial_set_timing(dial_ctx* ctx, int enable) {
  if (!ctx) return DIAL_ERR_ARG;
  ctx->timing = enable != 0;
  ctx->events_used = 0;
  return DIAL_OK;
}

int dial_get_rollout_ms(dial_ctx* ctx, double* total_ms, int* launches) {
  if (!ctx || !total_ms || !launches) return DIAL_ERR_ARG;
  double tot = 0.0;
  for (size_t i = 0; i < ctx->events_used; i++) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->events[i].second));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int)ctx->events_used;
  ctx->events_used = 0;
  return DIAL_OK;
}

static bool opt_no_queue(const dial_ctx* ctx) { return ctx->opt.no_queue != 0; }
static int launch_rollout(dial_ctx* ctx, const dial::RolloutIO& io_in, int B, hipStream_t st) {
  if (int rc = check_sticky(ctx)) return rc;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->timing) {
    if (ctx->events_used == ctx->events.size()) {
      hipEvent_t a, b;
      HIP_TRY(ctx, hipEventCreate(&a));
      HIP_TRY(ctx, hipEventCreate(&b));
      ctx->events.emplace_back(a, b);
    }
    e0 = ctx->events[ctx->events_used].first;
    e1 = ctx->events[ctx->events_used].second;
    ctx->events_used++;
    HIP_TRY(ctx, hipEventRecord(e0, st));
  }
  // batches beyond what the chip keeps resident: launch exactly the resident grid and let the wavefronts draw the
  // remaining rollouts from a queue (see rollout_kernel); the queue head starts behind the grid's own first rollouts
  const bool tracing = ctx->trace != nullptr;   // diagnostics: the TRACE instantiation on the plain grid (no queue / split / large-batch variant)
  // Go2: two rollouts per wavefront.  Up to the resident set of one-wavefront workgroups the grid covers the batch (wavefront p:
  // rollouts 2 p, 2 p + 1); beyond it the resident grid of DIAL_GO2_PAIR_WPB-wavefront workgroups draws pairs from the queue.
  // Which batches: the pair kernel executes ~1.1 x the instructions of the one-sample kernel for TWO rollouts, so it wins wherever
  // the SIMDs are short of issue slots -- N = 8192: 1.25 -> 1.10 ms, N = 65536: 8.96 -> 6.75 ms (7.15 -> 9.4 M rollouts/s) -- and
  // loses where the launch is one rollout's dependence chain long: a lone pair wavefront needs 46.7 k cycles per env.step against
  // 42.7 k (the DPP / permlane broadcasts sit on the chain, and it waits for the slower of its two rollouts' line searches):
  // N = 2048 0.380 -> 0.408 ms (final build of round 5: 0.363 -> 0.382).  Default: above DIAL_GO2_PAIR_MIN_B rollouts; dial_options.pair_mode 1 = never, 2 = always.
  // (profiles/r05_ab_pair_kernel.txt, profiles/r05_sections_pair_cycles.txt)
  if (ctx->pair_ok && !tracing && (ctx->opt.pair_mode == 2 || B > DIAL_GO2_PAIR_MIN_B)) {
    dial::RolloutIO io = io_in;
    const bool mean_last = !io.us && io.n_noise == B - 1 && ((B - 1) & 1) == 0 && ctx->relay_buf && ctx->relay_flag && !ctx->no_mean_inline;
    if ((B + 1) / 2 <= ctx->resident_pair || opt_no_queue(ctx)) {
      // plain grid: wavefront p runs rollouts 2 p, 2 p + 1 (the mean trajectory alone in the last one).  The interleaved mean
      // trajectory was measured here as well (N = 2048 as exactly 1024 wavefronts, one per SIMD): 0.418 ms against 0.389 ms with
      // the 1025th wavefront -- the hand-over code slows every wavefront more than the odd one costs (profiles/r05_ab_pair_n2048.txt)
      const int pairs = (B + 1) / 2;
      hipLaunchKernelGGL((rollout_kernel2<DimsGo2, 1, DIAL_GO2_PAIR_OCC, false, false>), dim3(pairs), dim3(64), ctx->lds_pair, st,
                         (const CModel<DimsGo2>*)ctx->dcm, (const dial_task*)ctx->dtask, (const dial_cfg*)ctx->dcfg, io, B, ctx->ws_words, (int*)nullptr);
    } else {
      // beyond the resident set: the queue -- and the mean trajectory rides along with the first T wavefronts (rollout_kernel2)
      // whenever it is the batch's last rollout and the noisy ones pair up exactly, so that N = 4096 / 8192 are one / two full
      // rounds of the resident grid (0.70 -> 0.48 ms, 1.24 -> 0.90 ms) instead of "+ 1 pair"
      const int blocks = ctx->resident_pair_large;
      if (mean_last && ctx->T <= blocks * DIAL_GO2_PAIR_WPB) {
        io.mean_inline = 1;
        io.relay_buf = ctx->relay_buf;
        io.relay_flag = ctx->relay_flag;
        io.err_word = ctx->err_dev;
      }
      HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->next, blocks * DIAL_GO2_PAIR_WPB, 1, st));
      hipLaunchKernelGGL((rollout_kernel2<DimsGo2, DIAL_GO2_PAIR_WPB, DIAL_GO2_PAIR_OCC, true, true>), dim3(blocks), dim3(64 * DIAL_GO2_PAIR_WPB),
                         ctx->lds_pair_large, st, (const CModel<DimsGo2>*)ctx->dcm, (const dial_task*)ctx->dtask, (const dial_cfg*)ctx->dcfg, io, B,
                         ctx->ws_words, ctx->next);
    }
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(e1, st));
    return DIAL_OK;
  }
  const bool large = ctx->inst == 1 && B > DIAL_GO2_LARGE_B && !tracing;
  const int wpb = large ? DIAL_GO2_WPB_LARGE : ctx->wpb;
  const int resident = large ? ctx->resident_blocks_large : ctx->resident_blocks;
  // mean-trajectory relay (one wavefront per workgroup, the launch's last rollout is the mean trajectory, everything is
  // resident): the "+1" rollout runs as ceil(T / 2) two-step pieces on as many SIMDs instead of one more wavefront on one
  dial::RolloutIO io = io_in;
  io.con_cap = ctx->ovf ? ctx->con_cap : 0;
  io.ovf = ctx->ovf;
  io.ovf_words = ctx->ovf_words;
  // (only when the noisy rollouts fill the SIMDs evenly and the mean trajectory is the odd one out: N = k x 1024)
  if (ctx->relay_ok && !large && !io.us && io.n_noise == B - 1 && B > 1 && ctx->T >= 4 && ctx->n_simd > 0 &&
      ((B - 1) % ctx->n_simd == 0 || ctx->relay_always)) {
    const int pieces = (ctx->T + ctx->relay_steps - 1) / ctx->relay_steps;
    if ((B - 1) + pieces <= ctx->resident_blocks) {
      io.relay_buf = ctx->relay_buf;
      io.relay_flag = ctx->relay_flag;
      io.relay_steps = ctx->relay_steps;
      io.relay_base = B - 1;
      io.err_word = ctx->err_dev;
      io.debug_stall_piece1 = ctx->debug_stall_piece1;
    }
  }
  int blocks = io.relay_flag ? io.relay_base + (ctx->T + io.relay_steps - 1) / io.relay_steps : (B + wpb - 1) / wpb;
  // Go2's large-batch kernel below its resident set (2304 < B <= 4096): the whole resident grid is launched and the rollouts are
  // dealt round-robin over the workgroups, so that every CU carries the same number of wavefronts
  // (blocks and resident count workgroups: the batch fits the resident grid without filling it)
  if (large && !io.relay_flag && resident > 0 && blocks < resident && !ctx->no_spread) {
    blocks = resident;
    io.spread = 1;
  }
  int* next = nullptr;
  const bool has_queue_variant = large || ctx->inst != 1;   // Go2's small-batch kernel never exceeds the resident set (B <= DIAL_GO2_LARGE_B)
  if (has_queue_variant && resident > 0 && blocks > resident && ctx->next && !tracing) {
    blocks = resident;
    next = ctx->next;
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)next, blocks * wpb, 1, st));
    // Go2's large batches whose last rollout is the mean trajectory: it is not a queue item -- the first T wavefronts of the
    // launch run one step of it each between two steps of their own (rollout_driver.h: mean_inline)
    if (large && !ctx->no_mean_inline && !io.us && io.n_noise == B - 1 && ctx->T <= blocks * wpb && io.relay_stride == 0) {
      io.mean_inline = 1;
      io.relay_buf = ctx->relay_buf;
      io.relay_flag = ctx->relay_flag;
      io.err_word = ctx->err_dev;
    }
    // (one hand-over protocol per launch: the slices exist for the elliptic models only, the interleaved mean trajectory for the Go2 only;
    //  the test keeps it that way should either condition widen -- both drive io.relay_buf / io.relay_flag)
    if (!io.mean_inline && ctx->slice_buf && B <= ctx->slice_cap && ctx->T > ctx->slice_steps) {   // time-sliced: (piece, rollout) items
      io.relay_buf = ctx->slice_buf;
      io.relay_flag = ctx->slice_flag;
      io.relay_stride = ctx->slice_stride;
      io.relay_steps = ctx->slice_steps;
      io.slice_pieces = (ctx->T + ctx->slice_steps - 1) / ctx->slice_steps;
      io.err_word = ctx->err_dev;
    }
  }
  if (ctx->work_stat && !io.slice_pieces && !tracing) {   // everything resident (or the plain queue): the slow rollouts first
    HIP_TRY(ctx, hipMemsetAsync(ctx->work_stat, 0, 2 * sizeof(int), st));
    io.work_stat = ctx->work_stat;
  }
  if (io.ovf && blocks * wpb > ctx->ovf_slots)
    return fail(ctx, DIAL_ERR_ARG, "rollout launch: more wavefronts than overflow areas (batch larger than the context's Nsample + 1)");
  if (ctx->trace) {
    if (B > ctx->trace_rows) return fail(ctx, DIAL_ERR_ARG, "rollout launch: more rollouts than rows of the state trace (dial_set_state_trace)");
    io.trace = ctx->trace;
  }
  // Batch = 8 x CUs + 1 (N = 2048 on 256 CUs) with multi-wavefront workgroups: the N noisy rollouts as evenly sized
  // workgroups that load every CU with 8 wavefronts (Allegro: one workgroup of 8, H1: two of 4, one wavefront per SIMD
  // each), the mean trajectory as a one-wavefront workgroup launched on the side stream (fork / join by events)
  if (ctx->split_ok && !tracing && !next && !io.us && io.n_noise == B - 1 && (B - 1) == 8 * (ctx->n_simd / 4)) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    dial::RolloutIO io1 = io;
    io1.n_first = B - 1;
#define DIAL_SPLIT_LAUNCH(D, WPBE, OCCE)                                                                                       \
    hipLaunchKernelGGL((rollout_kernel<D, 1>), dim3(1), dim3(64), ctx->lds_one, ctx->side, (const CModel<D>*)ctx->dcm,     \
                       (const dial_task*)ctx->dtask, (const dial_cfg*)ctx->dcfg, io1, B, ctx->ws_words, (int*)nullptr);    \
    HIP_TRY(ctx, hipGetLastError());                                                                                       \
    HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->side));                                                                 \
    hipLaunchKernelGGL((rollout_kernel<D, WPBE, OCCE>), dim3((B - 1) / WPBE), dim3(64 * WPBE), ctx->lds_even, st,                \
                       (const CModel<D>*)ctx->dcm, (const dial_task*)ctx->dtask, (const dial_cfg*)ctx->dcfg, io, B - 1,    \
                       ctx->ws_words, (int*)nullptr)
    if (ctx->inst == 4) { DIAL_SPLIT_LAUNCH(DimsAllegro, DIAL_ALLEGRO_WPB_EVEN, DIAL_EVEN_OCC_ALLEGRO); }
    else { DIAL_SPLIT_LAUNCH(DimsH1, DIAL_H1_WPB_EVEN, DIAL_EVEN_OCC_H1); }
#undef DIAL_SPLIT_LAUNCH
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(e1, st));
    return DIAL_OK;
  }
#define DIAL_LAUNCH_ROLLOUT_Q(D, WPB, Q)                                                                    \
  hipLaunchKernelGGL((rollout_kernel<D, WPB, 3, Q>), dim3(blocks), dim3(64 * WPB), ctx->lds_rollout,        \
                     st, (const CModel<D>*)ctx->dcm, (const dial_task*)ctx->dtask,                          \
                     (const dial_cfg*)ctx->dcfg, io, B, ctx->ws_words, next)
#define DIAL_LAUNCH_ROLLOUT(D, WPB)                                        \
  do {                                                                     \
    if (next) DIAL_LAUNCH_ROLLOUT_Q(D, WPB, true);                         \
    else DIAL_LAUNCH_ROLLOUT_Q(D, WPB, false);                             \
  } while (0)
#define DIAL_LAUNCH_TRACE(D, WPB)                                                                           \
  hipLaunchKernelGGL((rollout_kernel<D, WPB, 3, false, true>), dim3(blocks), dim3(64 * WPB), ctx->lds_rollout, \
                     st, (const CModel<D>*)ctx->dcm, (const dial_task*)ctx->dtask,                          \
                     (const dial_cfg*)ctx->dcfg, io, B, ctx->ws_words, (int*)nullptr)
  if (tracing) {
    if (ctx->inst == 1) DIAL_LAUNCH_TRACE(DimsGo2, 1);
    else if (ctx->inst == 2) DIAL_LAUNCH_TRACE(DimsH1, 3);
    else if (ctx->inst == 3) DIAL_LAUNCH_TRACE(DimsH1Loco, 2);
    else if (ctx->inst == 4) DIAL_LAUNCH_TRACE(DimsAllegro, DIAL_ALLEGRO_WPB);
    else if (ctx->inst == 5) DIAL_LAUNCH_TRACE(DimsGo2Crate, DIAL_CRATE_WPB);
    else if (ctx->inst == 6) DIAL_LAUNCH_TRACE(DimsH1PushCrate, DIAL_CRATE_WPB);
    else DIAL_LAUNCH_TRACE(DimsMax, 1);
  } else
  if (large)
    hipLaunchKernelGGL((rollout_kernel<DimsGo2, DIAL_GO2_WPB_LARGE, DIAL_GO2_OCC_LARGE, true>), dim3(blocks),
                       dim3(64 * DIAL_GO2_WPB_LARGE), ctx->lds_large, st, (const CModel<DimsGo2>*)ctx->dcm, (const dial_task*)ctx->dtask,
                       (const dial_cfg*)ctx->dcfg, io, B, ctx->ws_words, next);
  else if (ctx->inst == 1) DIAL_LAUNCH_ROLLOUT_Q(DimsGo2, 1, false);
  else if (ctx->inst == 2) DIAL_LAUNCH_ROLLOUT(DimsH1, 3);
  else if (ctx->inst == 3) DIAL_LAUNCH_ROLLOUT(DimsH1Loco, 2);
  else if (ctx->inst == 4) DIAL_LAUNCH_ROLLOUT(DimsAllegro, DIAL_ALLEGRO_WPB);
  else if (ctx->inst == 5) DIAL_LAUNCH_ROLLOUT(DimsGo2Crate, DIAL_CRATE_WPB);
  else if (ctx->inst == 6) DIAL_LAUNCH_ROLLOUT(DimsH1PushCrate, DIAL_CRATE_WPB);
  else DIAL_LAUNCH_ROLLOUT(DimsMax, 1);
#undef DIAL_LAUNCH_ROLLOUT
#undef DIAL_LAUNCH_ROLLOUT_Q
#undef DIAL_LAUNCH_TRACE
  HIP_TRY(ctx, hipGetLastError());
  if (ctx->timing) HIP_TRY(ctx, hipEventRecord(e1, st));
  return DIAL_OK;
}

int dial_rollout(dial_ctx* ctx, const float* state, const float* us, int B, float* rewss, float* qss, float* qdss,
                 float* xposs, void* stream) {
  if (!ctx || !state || !us || !rewss || B < 1) return fail(ctx, DIAL_ERR_ARG, "dial_rollout: bad argument");
  if (!ctx->has_cfg) return fail(ctx, DIAL_ERR_ARG, "dial_rollout: context was created without a dial_cfg");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dial::RolloutIO io{state, us, nullptr, nullptr, nullptr, 0, 0, ctx->T, ctx->Hn1, nullptr, rewss, nullptr, qss, qdss, xposs, nullptr, 0, 0u, 0u, 0u, 0};
  return launch_rollout(ctx, io, B, (hipStream_t)stream);
}

static int shard_rollout_impl(dial_ctx* ctx, const float* state, const float* Ybar_in, const float* noise_scale,
                              int ns, const float* eps, int use_rng, uint64_t seed, uint32_t counter, int n_begin,
                              int n_local, int with_mean, float* rews_local, void* stream, const char* who,
                              bool store_states = true) {
  if (!ctx || !state || !Ybar_in || !noise_scale || (!eps && !use_rng && n_local > 0) || !rews_local)
    return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": null argument");
  if (!ctx->has_cfg) return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": context was created without a dial_cfg");
  if (ns != 1 && ns != ctx->Hn1) return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": noise_scale must have 1 or Hnode+1 entries");
  // with_mean: bit 0 = roll out the mean trajectory as an extra sample; bit 1 (DIAL_SHARD_LEAN) = the caller wants the mean action only
  // (dial_shard_ybar*): the rollouts do not materialise their per-step states nor their candidate nodes
  if (with_mean & 2) store_states = false;
  const bool lean = (with_mean & 2) != 0;
  with_mean &= 1;
  const int B = n_local + (with_mean ? 1 : 0);
  if (n_local < 0 || B < 1 || B > ctx->B_cap || n_begin < 0) return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": shard larger than Nsample+1");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dial::RolloutIO io{state, nullptr, eps, Ybar_in, noise_scale, ns, n_local, ctx->T, ctx->Hn1,
                     lean ? nullptr : ctx->Y0s, ctx->rewss, rews_local, store_states ? ctx->qss : nullptr, store_states ? ctx->qdss : nullptr,
                     store_states ? ctx->xss : nullptr, ctx->prof,
                     use_rng, (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), counter, n_begin};
  return launch_rollout(ctx, io, B, (hipStream_t)stream);
}

int dial_shard_rollout(dial_ctx* ctx, const float* state, const float* Ybar_in, const float* noise_scale, int ns,
                       const float* eps, int n_local, int with_mean, float* rews_local, void* stream) {
  return shard_rollout_impl(ctx, state, Ybar_in, noise_scale, ns, eps, 0, 0, 0, 0, n_local, with_mean, rews_local,
                            stream, "dial_shard_rollout");
}

int dial_shard_rollout_rng(dial_ctx* ctx, const float* state, const float* Ybar_in, const float* noise_scale, int ns,
                           uint64_t seed, uint32_t counter, int n_begin, int n_local, int with_mean,
                           float* rews_local, void* stream) {
  return shard_rollout_impl(ctx, state, Ybar_in, noise_scale, ns, nullptr, 1, seed, counter, n_begin, n_local,
                            with_mean, rews_local, stream, "dial_shard_rollout_rng");
}

int dial_rng_fill(dial_ctx* ctx, uint64_t seed, uint32_t counter, int n_begin, int n_count, float* eps_out,
                  void* stream) {
  if (!ctx || !eps_out || n_count < 0 || n_begin < 0) return fail(ctx, DIAL_ERR_ARG, "dial_rng_fill: bad argument");
  if (!ctx->has_cfg) return fail(ctx, DIAL_ERR_ARG, "dial_rng_fill: context was created without a dial_cfg");
  if (n_count == 0) return DIAL_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int C = ctx->Hn1 * ctx->hm.nu, nq = (C + 3) / 4;
  const long long total = (long long)n_count * nq;
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), counter, n_begin, n_count, C, eps_out);
  HIP_TRY(ctx, hipGetLastError());
  return DIAL_OK;
}

static int launch_wsum(dial_ctx* ctx, const float* weights, int n_rows, int w_begin, int mean_row, int mean_widx,
                       float* Yo, float* qo, float* qdo, float* xo, hipStream_t st, bool nodes_only = false) {
  const dial_model& m = ctx->hm;
  WsumArgs a;
  const int T = ctx->T;
  a.nseg = nodes_only ? 1 : 4;   // nodes_only: the weighted mean action alone (the per-step states were not materialised)
  a.seg[0] = {ctx->Y0s, Yo, ctx->Hn1 * m.nu, 0};
  a.seg[1] = {ctx->qss, qo, T * m.nq, a.seg[0].C};
  a.seg[2] = {ctx->qdss, qdo, T * m.nv, a.seg[1].c0 + a.seg[1].C};
  a.seg[3] = {ctx->xss, xo, T * ctx->nx, a.seg[2].c0 + a.seg[2].C};
  a.Ctot = nodes_only ? a.seg[0].C : a.seg[3].c0 + a.seg[3].C;
  a.n_rows = n_rows; a.w_begin = w_begin; a.mean_row = mean_row; a.mean_widx = mean_widx;
  const int gx = (a.Ctot + 255) / 256;
  hipLaunchKernelGGL(wsum_partial_kernel, dim3((a.Ctot + 63) / 64, WSUM_CHUNKS), dim3(256), 0, st, a, weights, ctx->partial);
  hipLaunchKernelGGL(wsum_final_kernel, dim3(gx), dim3(256), 0, st, a, (const float*)ctx->partial);
  HIP_TRY(ctx, hipGetLastError());
  return DIAL_OK;
}

// gathered != nullptr: the rewards as the all-gather delivered them ([world][per + 1]); the weights kernel puts them in order on the
// way (-> rews_out).  Otherwise rews_all is read as is.
static int shard_reduce_impl(dial_ctx* ctx, const float* rews_all, const float* gathered, int world, int per, float* rews_out,
                             int n_total, int n_begin, int n_local, int with_mean, float* packed_out, void* stream, const char* who) {
  if (!ctx || (!rews_all && !gathered) || !packed_out) return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": null argument");
  if (gathered && (!rews_out || world < 1 || per < 1 || n_total < 1 || (long long)world * per < n_total))
    return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": bad description of the gathered rewards");
  if (!ctx->has_cfg || n_local < 0 || n_local + 1 > ctx->B_cap || n_begin < 0 || n_begin + n_local > n_total)
    return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": bad shard description");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  // the global weights need n_total+1 floats of ctx->weights
  if (n_total + 1 > ctx->W_cap) return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": n_total exceeds the context's Nsample (create it with Nsample = global sample count)");
  hipLaunchKernelGGL(weights_kernel, dim3(1), dim3(WK_THREADS), 0, st, rews_all, n_total + 1, ctx->hc.temp_sample, ctx->weights, gathered, per, rews_out);
  HIP_TRY(ctx, hipGetLastError());
  const dial_model& m = ctx->hm;
  float* Yo = packed_out;
  float* qo = Yo + ctx->Hn1 * m.nu;
  float* qdo = qo + ctx->T * m.nq;
  float* xo = qdo + ctx->T * m.nv;
  // local rows: [0,n_local) noisy, row n_local = mean trajectory (always rolled out by dial_shard_rollout
  // when requested there); it contributes to the partial sums only when with_mean != 0.
  return launch_wsum(ctx, ctx->weights, n_local + 1, n_begin, n_local, with_mean ? n_total : -1, Yo, qo, qdo, xo, st);
}
int dial_shard_reduce(dial_ctx* ctx, const float* rews_all, int n_total, int n_begin, int n_local, int with_mean,
                      float* packed_out, void* stream) {
  return shard_reduce_impl(ctx, rews_all, nullptr, 0, 0, nullptr, n_total, n_begin, n_local, with_mean, packed_out, stream, "dial_shard_reduce");
}
int dial_shard_reduce_gathered(dial_ctx* ctx, const float* gathered, int world, int per, int n_total, int n_begin, int n_local,
                               int with_mean, float* rews_all_out, float* packed_out, void* stream) {
  return shard_reduce_impl(ctx, nullptr, gathered, world, per, rews_all_out, n_total, n_begin, n_local, with_mean, packed_out, stream,
                           "dial_shard_reduce_gathered");
}

static int shard_ybar_impl(dial_ctx* ctx, const float* rews_all, const float* gathered, int world, int per, float* rews_out,
                           int n_total, const float* eps_all, int use_rng, uint64_t seed,
                           uint32_t counter, const float* Ybar_in, const float* noise_scale, int ns, float* Ybar_out,
                           void* stream, const char* who) {
  if (!ctx || (!rews_all && !gathered) || (!eps_all && !use_rng) || !Ybar_in || !noise_scale || !Ybar_out)
    return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": null argument");
  if (gathered && (!rews_out || world < 1 || per < 1 || (long long)world * per < n_total))
    return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": bad description of the gathered rewards");
  if (!ctx->has_cfg || n_total + 1 > ctx->W_cap || n_total < 1 || (ns != 1 && ns != ctx->Hn1))
    return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": bad arguments (create the context with Nsample = global sample count)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  const int C = ctx->Hn1 * ctx->hm.nu, nquad = (C + 3) / 4;
  static_assert(YB_CHUNKS == YB_CHUNKS_HOST, "the scratch buffer is sized with YB_CHUNKS_HOST");
  hipLaunchKernelGGL(weights_kernel, dim3(1), dim3(WK_THREADS), 0, st, rews_all, n_total + 1, ctx->hc.temp_sample, ctx->weights, gathered, per, rews_out);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(ybar_partial_kernel, dim3((nquad + 15) / 16, YB_CHUNKS), dim3(256), 0, st, (const float*)ctx->weights,
                     eps_all, Ybar_in, noise_scale, ns, n_total, C, ctx->hm.nu, ctx->partial,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), counter);
  hipLaunchKernelGGL(ybar_final_kernel, dim3((C + 63) / 64), dim3(64), 0, st, (const float*)ctx->partial, C, Ybar_out);
  HIP_TRY(ctx, hipGetLastError());
  return DIAL_OK;
}

int dial_shard_ybar(dial_ctx* ctx, const float* rews_all, int n_total, const float* eps_all, const float* Ybar_in,
                    const float* noise_scale, int ns, float* Ybar_out, void* stream) {
  return shard_ybar_impl(ctx, rews_all, nullptr, 0, 0, nullptr, n_total, eps_all, 0, 0, 0, Ybar_in, noise_scale, ns, Ybar_out, stream, "dial_shard_ybar");
}

int dial_shard_ybar_rng(dial_ctx* ctx, const float* rews_all, int n_total, uint64_t seed, uint32_t counter,
                        const float* Ybar_in, const float* noise_scale, int ns, float* Ybar_out, void* stream) {
  return shard_ybar_impl(ctx, rews_all, nullptr, 0, 0, nullptr, n_total, nullptr, 1, seed, counter, Ybar_in, noise_scale, ns, Ybar_out, stream,
                         "dial_shard_ybar_rng");
}

int dial_shard_ybar_gathered(dial_ctx* ctx, const float* gathered, int world, int per, int n_total, const float* eps_all,
                             const float* Ybar_in, const float* noise_scale, int ns, float* rews_all_out, float* Ybar_out, void* stream) {
  return shard_ybar_impl(ctx, nullptr, gathered, world, per, rews_all_out, n_total, eps_all, 0, 0, 0, Ybar_in, noise_scale, ns, Ybar_out, stream,
                         "dial_shard_ybar_gathered");
}

int dial_shard_ybar_gathered_rng(dial_ctx* ctx, const float* gathered, int world, int per, int n_total, uint64_t seed, uint32_t counter,
                                 const float* Ybar_in, const float* noise_scale, int ns, float* rews_all_out, float* Ybar_out, void* stream) {
  return shard_ybar_impl(ctx, nullptr, gathered, world, per, rews_all_out, n_total, nullptr, 1, seed, counter, Ybar_in, noise_scale, ns, Ybar_out,
                         stream, "dial_shard_ybar_gathered_rng");
}

int dial_shard_pack_rewards(dial_ctx* ctx, const float* gathered, int world, int per, int n_total, float* rews_all,
                            void* stream) {
  if (!ctx || !gathered || !rews_all || world < 1 || per < 1 || n_total < 1 || (long long)world * per < n_total)
    return fail(ctx, DIAL_ERR_ARG, "dial_shard_pack_rewards: bad arguments");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(pack_rewards_kernel, dim3((n_total + 1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, gathered, per,
                     n_total, rews_all);
  HIP_TRY(ctx, hipGetLastError());
  return DIAL_OK;
}

static int reverse_once_impl(dial_ctx* ctx, const float* state, const float* Ybar_in, const float* noise_scale, int ns,
                             const float* eps, int use_rng, uint64_t seed, uint32_t counter, float* Ybar_out,
                             float* rews, float* qbar, float* qdbar, float* xbar, void* stream, const char* who) {
  if (!ctx || !Ybar_out || !rews) return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": null argument");
  if (!ctx->has_cfg) return fail(ctx, DIAL_ERR_ARG, std::string(who) + ": context was created without a dial_cfg");
  const int N = ctx->hc.Nsample;
  // qbar == qdbar == xbar == NULL: the caller wants the mean action only (every annealing iteration of a plan but the last,
  // dial_core.py:262-264 / dial_plan.py:214-215 read the bars of the LAST one) -- the rollouts then do not write their
  // per-step q / qd / x.pos rows at all (11 MB per launch for Go2) and K4b sums the candidate nodes only
  const bool bars = qbar || qdbar || xbar;
  int rc = shard_rollout_impl(ctx, state, Ybar_in, noise_scale, ns, eps, use_rng, seed, counter, 0, N, 1, rews, stream, who, bars);
  if (rc != DIAL_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(weights_kernel, dim3(1), dim3(WK_THREADS), 0, st, (const float*)rews, N + 1, ctx->hc.temp_sample, ctx->weights, (const float*)nullptr, 0, (float*)nullptr);
  HIP_TRY(ctx, hipGetLastError());
  return launch_wsum(ctx, ctx->weights, N + 1, 0, N, N, Ybar_out, qbar, qdbar, xbar, st, !bars);
}

int dial_reverse_once(dial_ctx* ctx, const float* state, const float* Ybar_in, const float* noise_scale, int ns,
                      const float* eps, float* Ybar_out, float* rews, float* qbar, float* qdbar, float* xbar,
                      void* stream) {
  return reverse_once_impl(ctx, state, Ybar_in, noise_scale, ns, eps, 0, 0, 0, Ybar_out, rews, qbar, qdbar, xbar, stream,
                           "dial_reverse_once");
}

int dial_reverse_once_rng(dial_ctx* ctx, const float* state, const float* Ybar_in, const float* noise_scale, int ns,
                          uint64_t seed, uint32_t counter, float* Ybar_out, float* rews, float* qbar, float* qdbar,
                          float* xbar, void* stream) {
  return reverse_once_impl(ctx, state, Ybar_in, noise_scale, ns, nullptr, 1, seed, counter, Ybar_out, rews, qbar, qdbar,
                           xbar, stream, "dial_reverse_once_rng");
}

int dial_shift(dial_ctx* ctx, float* Y, void* stream) {
  if (!ctx || !Y) return fail(ctx, DIAL_ERR_ARG, "dial_shift: null argument");
  if (!ctx->has_cfg) return fail(ctx, DIAL_ERR_ARG, "dial_shift: context was created without a dial_cfg");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(shift_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const dial_cfg*)ctx->dcfg, ctx->hm.nu, Y);
  HIP_TRY(ctx, hipGetLastError());
  return DIAL_OK;
}

int dial_env_step(dial_ctx* ctx, float* state, const float* action, float* xpos_out, float* xquat_out,
                  float* ctrl_out, void* stream) {
  if (!ctx || !state || !action) return fail(ctx, DIAL_ERR_ARG, "dial_env_step: null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
#define DIAL_LAUNCH_STEP(D)                                                                                         \
  hipLaunchKernelGGL(env_step_kernel<D>, dim3(1), dim3(64), ctx->lds_bytes, (hipStream_t)stream,                   \
                     (const CModel<D>*)ctx->dcm, (const dial_task*)ctx->dtask, state, action, xpos_out, xquat_out, \
                     ctrl_out)
  if (ctx->inst == 1) DIAL_LAUNCH_STEP(DimsGo2);
  else if (ctx->inst == 2) DIAL_LAUNCH_STEP(DimsH1);
  else if (ctx->inst == 3) DIAL_LAUNCH_STEP(DimsH1Loco);
  else if (ctx->inst == 4) DIAL_LAUNCH_STEP(DimsAllegro);
  else if (ctx->inst == 5) DIAL_LAUNCH_STEP(DimsGo2Crate);
  else if (ctx->inst == 6) DIAL_LAUNCH_STEP(DimsH1PushCrate);
  else DIAL_LAUNCH_STEP(DimsMax);
#undef DIAL_LAUNCH_STEP
  HIP_TRY(ctx, hipGetLastError());
  return DIAL_OK;
}

static int env_reset_launch(dial_ctx* ctx, const float* qpos, const float* qvel, float* state, float* xpos_out, float* xquat_out,
                            int n, void* stream, const char* who) {
  if (!ctx || !state || !qpos || !qvel || n < 1) return fail(ctx, DIAL_ERR_ARG, who);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
#define DIAL_LAUNCH_RESET(D)                                                                        \
  hipLaunchKernelGGL(env_reset_kernel<D>, dim3(n), dim3(64), ctx->lds_bytes, (hipStream_t)stream,   \
                     (const CModel<D>*)ctx->dcm, qpos, qvel, state, xpos_out, xquat_out)
  if (ctx->inst == 1) DIAL_LAUNCH_RESET(DimsGo2);
  else if (ctx->inst == 2) DIAL_LAUNCH_RESET(DimsH1);
  else if (ctx->inst == 3) DIAL_LAUNCH_RESET(DimsH1Loco);
  else if (ctx->inst == 4) DIAL_LAUNCH_RESET(DimsAllegro);
  else if (ctx->inst == 5) DIAL_LAUNCH_RESET(DimsGo2Crate);
  else if (ctx->inst == 6) DIAL_LAUNCH_RESET(DimsH1PushCrate);
  else DIAL_LAUNCH_RESET(DimsMax);
#undef DIAL_LAUNCH_RESET
  HIP_TRY(ctx, hipGetLastError());
  return DIAL_OK;
}
int dial_env_reset(dial_ctx* ctx, const float* qpos, const float* qvel, float* state, float* xpos_out,
                   float* xquat_out, void* stream) {
  return env_reset_launch(ctx, qpos, qvel, state, xpos_out, xquat_out, 1, stream, "dial_env_reset: null argument");
}
int dial_env_reset_batch(dial_ctx* ctx, const float* qpos, const float* qvel, float* states, float* xpos_out,
                         float* xquat_out, int n, void* stream) {
  return env_reset_launch(ctx, qpos, qvel, states, xpos_out, xquat_out, n, stream, "dial_env_reset_batch: null argument or n < 1");
}

// Internal diagnostics (not part of the public header): wave primitive self-test and the scratch
// tensors of the last reverse_once, used by the GPU parity tests for stage-wise comparison.
int dial_selftest(float* out3_host) {
  float* d = nullptr;
  if (hipMalloc(&d, 3 * sizeof(float)) != hipSuccess) return DIAL_ERR_HIP;
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, 0, d);
  hipError_t e = hipMemcpy(out3_host, d, 3 * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e == hipSuccess ? DIAL_OK : DIAL_ERR_HIP;
}
int dial_debug_scratch(dial_ctx* ctx, float** Y0s, float** rewss, float** qss, float** qdss, float** xss, float** weights) {
  if (!ctx) return DIAL_ERR_ARG;
  if (Y0s) *Y0s = ctx->Y0s;
  if (rewss) *rewss = ctx->rewss;
  if (qss) *qss = ctx->qss;
  if (qdss) *qdss = ctx->qdss;
  if (xss) *xss = ctx->xss;
  if (weights) *weights = ctx->weights;
  return DIAL_OK;
}
int dial_lds_bytes(dial_ctx* ctx) { return ctx ? (int)ctx->lds_rollout : -1; }
// tests: switch the relay-stall hook of DIAL_DEBUG_RELAY_STALL on (k >= 1: piece k - 1 never hands over) or off (0)
int dial_debug_set_stall(dial_ctx* ctx, int piece1) { if (!ctx) return DIAL_ERR_ARG; ctx->debug_stall_piece1 = piece1; return DIAL_OK; }
// wavefront slots of the rollout kernel on the whole chip for a batch of B rollouts (B > slots: the rollout queue runs)
int dial_debug_resident_rollouts(dial_ctx* ctx, int B) {
  if (!ctx) return -1;
  if (ctx->pair_ok && (ctx->opt.pair_mode == 2 || B > DIAL_GO2_PAIR_MIN_B)) {
    if (ctx->opt.no_queue) return 0;
    return (B + 1) / 2 <= ctx->resident_pair ? 2 * ctx->resident_pair : 2 * DIAL_GO2_PAIR_WPB * ctx->resident_pair_large;
  }
  const bool large = ctx->inst == 1 && B > DIAL_GO2_LARGE_B;
  return large ? ctx->resident_blocks_large * DIAL_GO2_WPB_LARGE : ctx->resident_blocks * ctx->wpb;
}
// DIAL_PROFILE builds: 6 words per rollout of the last launch -- start / end wall-clock timestamps (100 MHz), then
// the rollout's own event counters 27, 28, 30, 31
int dial_debug_wave_times(dial_ctx* ctx, unsigned long long* out, int n) {
  if (!ctx || !ctx->prof || n < 0 || n > ctx->B_cap) return DIAL_ERR_ARG;
  return hipMemcpy(out, ctx->prof + 32, sizeof(unsigned long long) * 6 * (size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? DIAL_OK : DIAL_ERR_HIP;
}
// DIAL_PROFILE builds: cycle counters of sample 0 of the last rollout launch (16 sections)
int dial_debug_prof(dial_ctx* ctx, unsigned long long* out16 /* 32 entries */) {
  if (!ctx || !ctx->prof) return DIAL_ERR_ARG;
  return hipMemcpy(out16, ctx->prof, sizeof(unsigned long long) * 32, hipMemcpyDeviceToHost) == hipSuccess ? DIAL_OK : DIAL_ERR_HIP;
}

}  // extern "C"
