// kernel_list.h -- which instantiations of the kernels of rollout_kernel.h the library carries, per robot FAMILY.  Each family
// is compiled in its own translation unit (kern_family.hip with -DDIAL_FAMILY=k; dial_mpc_amd/_lib.py builds them in
// parallel), dial_hip.hip declares them `extern template` and only holds the host side and the small K4 / K5 kernels.
#pragma once
#include "rollout_kernel.h"

// wavefronts per workgroup (they share one staged copy of the constants) / occupancy target of the Go2's ONE-rollout-per-wavefront
// instantiations: small batches (every sample co-resident at 1 wavefront per workgroup, 168 VGPRs, 0 scratch, 2 wavefronts per SIMD
// at N = 2048) and, since round 4, a large-batch variant of EIGHT wavefronts per workgroup (78 KB: two workgroups = 16 wavefronts per
// CU) compiled for FOUR wavefronts per SIMD (128 VGPRs; ISA probe of round 5: 13 spilled VGPRs / 32 B of scratch after the
// wavefront index became a scalar, 47 / 104 B before).  Since round 5 the default launch for batches beyond 2304 rollouts is the
// two-rollouts-per-wavefront kernel below (N = 65536: 7.15 -> 9.3 M rollouts/s); this variant is what dial_options.pair_mode = 1
// selects (the A/B arm of profiles/r05_ab_pair_kernel.txt).
#ifndef DIAL_GO2_WPB_LARGE
#define DIAL_GO2_WPB_LARGE 8
#endif
#ifndef DIAL_GO2_OCC_LARGE
#define DIAL_GO2_OCC_LARGE 4
#endif
// Round 5: TWO samples per wavefront (rollout_kernel2, wave.h: WaveH).  Small batches: one wavefront per workgroup (7.3 KB of
// constants + 2 x 8.9 KB), N + 1 = 2049 rollouts = 1025 wavefronts = ONE per SIMD; large ones: the rollout queue over workgroups
// of DIAL_GO2_PAIR_WPB wavefronts (76.5 KB: two per CU = two wavefronts = four rollouts per SIMD).  Compiled for two wavefronts
// per SIMD (<= 256 VGPRs; round 6: the plain grid uses 199, the queue 244, no scratch -- tools/isa/disasm_lib.py prints the table).
#ifndef DIAL_GO2_PAIR_WPB
#define DIAL_GO2_PAIR_WPB 4
#endif
#ifndef DIAL_GO2_PAIR_OCC
#define DIAL_GO2_PAIR_OCC 2
#endif
// Allegro: 15.3 KB of workspace per wavefront + 10.5 KB of shared constants.  9 wavefronts per workgroup = 148 KB = one
// workgroup per CU = 2304 resident rollouts: the example's N + 1 = 2049 run in ONE round (8 per CU would leave the
// 2049th rollout for a second round and double the launch time), BASELINE config 4 (4097) in two instead of three.
#ifndef DIAL_ALLEGRO_WPB
#define DIAL_ALLEGRO_WPB 9
#endif
// ... except when the batch is exactly N + 1 = 8 x CUs + 1: then 8 wavefronts per workgroup (one workgroup per CU, every CU
// equally loaded) and the mean-trajectory rollout as a one-wavefront workgroup of its own (26.6 KB: fits beside a 133 KB
// workgroup), launched on a side stream so that it runs concurrently
#define DIAL_ALLEGRO_WPB_EVEN 8
// H1: same idea with 4-wavefront workgroups (one wavefront per SIMD), two per CU: 1.012 -> 0.971 ms.  (H1 loco's
// two-wavefront workgroups already load every CU with 8 wavefronts; the split measured 0.7 % slower there.)
#define DIAL_H1_WPB_EVEN 4
// The even launch keeps exactly 2 wavefronts on every SIMD.  Allegro's 8-wavefront kernel is compiled for that occupancy
// (-3.3 % in the A/B, 7.70 -> 7.45 ms: the scheduler orders for latency at the lower occupancy target; the kernel still uses
// 150 VGPRs, so the one-wavefront mean-trajectory workgroup finds room beside two of its wavefronts).  H1's must stay at the 3-wavefront budget: at 2 its registers leave no SIMD for the mean-trajectory workgroup, which
// then runs AFTER the even launch (+50 %).
#ifndef DIAL_EVEN_OCC_ALLEGRO
#define DIAL_EVEN_OCC_ALLEGRO 2
#endif
#define DIAL_EVEN_OCC_H1 3
// Crate scenes' own instantiations (DimsGo2Crate / DimsH1PushCrate): generic feature set, compile-time dimensions, constants
// staged in LDS and shared by the NINE wavefronts of a workgroup -- one workgroup per CU (11-12 KB of constants + 9 x ~16.5 KB)
#ifndef DIAL_CRATE_WPB
#define DIAL_CRATE_WPB 9
#endif

// X(Dims, wavefronts per workgroup, occupancy target, rollout-queue variant, state-trace variant); XE(Dims): env.step / env.reset
#define DIAL_KERNELS_GO2(X, XE) \
  X(DimsGo2, 1, 3, false, false) X(DimsGo2, 1, 3, false, true) X(DimsGo2, DIAL_GO2_WPB_LARGE, DIAL_GO2_OCC_LARGE, true, false) XE(DimsGo2)
// X2(Dims, wavefronts per workgroup, occupancy target, rollout-queue variant, interleaved mean trajectory): the two-samples-per-wavefront kernels
#define DIAL_KERNELS2_GO2(X2) X2(DimsGo2, 1, DIAL_GO2_PAIR_OCC, false, false) X2(DimsGo2, DIAL_GO2_PAIR_WPB, DIAL_GO2_PAIR_OCC, true, true)
#define DIAL_KERNELS_H1(X, XE) \
  X(DimsH1, 3, 3, false, false) X(DimsH1, 3, 3, true, false) X(DimsH1, 3, 3, false, true) \
  X(DimsH1, DIAL_H1_WPB_EVEN, DIAL_EVEN_OCC_H1, false, false) X(DimsH1, 1, 3, false, false) XE(DimsH1)
#define DIAL_KERNELS_H1LOCO(X, XE) \
  X(DimsH1Loco, 2, 3, false, false) X(DimsH1Loco, 2, 3, true, false) X(DimsH1Loco, 2, 3, false, true) XE(DimsH1Loco)
#define DIAL_KERNELS_ALLEGRO(X, XE) \
  X(DimsAllegro, DIAL_ALLEGRO_WPB, 3, false, false) X(DimsAllegro, DIAL_ALLEGRO_WPB, 3, true, false) X(DimsAllegro, DIAL_ALLEGRO_WPB, 3, false, true) \
  X(DimsAllegro, DIAL_ALLEGRO_WPB_EVEN, DIAL_EVEN_OCC_ALLEGRO, false, false) X(DimsAllegro, 1, 3, false, false) XE(DimsAllegro)
#define DIAL_KERNELS_GENERIC(X, XE) \
  X(DimsMax, 1, 3, false, false) X(DimsMax, 1, 3, true, false) X(DimsMax, 1, 3, false, true) XE(DimsMax)
#define DIAL_KERNELS_GO2CRATE(X, XE) \
  X(DimsGo2Crate, DIAL_CRATE_WPB, 3, false, false) X(DimsGo2Crate, DIAL_CRATE_WPB, 3, true, false) X(DimsGo2Crate, DIAL_CRATE_WPB, 3, false, true) XE(DimsGo2Crate)
#define DIAL_KERNELS_H1PUSHCRATE(X, XE) \
  X(DimsH1PushCrate, DIAL_CRATE_WPB, 3, false, false) X(DimsH1PushCrate, DIAL_CRATE_WPB, 3, true, false) X(DimsH1PushCrate, DIAL_CRATE_WPB, 3, false, true) XE(DimsH1PushCrate)
#define DIAL_KERNELS_ALL(X, XE) \
  DIAL_KERNELS_GO2(X, XE) DIAL_KERNELS_H1(X, XE) DIAL_KERNELS_H1LOCO(X, XE) DIAL_KERNELS_ALLEGRO(X, XE) DIAL_KERNELS_GENERIC(X, XE) \
  DIAL_KERNELS_GO2CRATE(X, XE) DIAL_KERNELS_H1PUSHCRATE(X, XE)
#define DIAL_N_FAMILIES 8
