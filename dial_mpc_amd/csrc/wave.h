// wave.h -- the "one wavefront owns one sample" execution model.
//
// A rollout is owned by ONE 64-lane wavefront (workgroup = 64 threads).  The kernel body
// (rollout_body.h) is written as a sequence of *phases*:
//
//   w.items(count, [&](int i) { ... });   // work items i = lane, lane+64, ... ; LDS fence after
//   float s = w.sum(count, [&](int i) { return ...; });   // wave-uniform reduction (count <= 64)
//
// Rules the body obeys (checked by the CPU emulation build, see below):
//   * inside one items() call an item only reads LDS written by EARLIER phases and writes LDS
//     words no other item of the same call touches;
//   * variables declared outside the lambdas are wave-uniform (same value in every lane);
//     lambdas never assign to them;
//   * all cross-lane traffic goes through LDS or through sum()/sum3().
//
// Two implementations of the same interface:
//   * HIP/gfx950 (default): lane = threadIdx.x; items() ends in a workgroup barrier, which for a
//     single-wave workgroup is an LDS wait (s_waitcnt lgkmcnt(0)) -- DS operations of one wave
//     execute in order, so that is all the synchronisation needed; sum() is a DPP butterfly in
//     registers (no LDS traffic).
//   * DIAL_EMU (g++ on the host, TEST INFRASTRUCTURE): lanes are executed sequentially.  With
//     race checking on, every items() phase is executed twice from the same LDS snapshot, in
//     ascending and in descending item order, and the resulting LDS images must be bit-identical;
//     any intra-phase read-after-write / write-after-write dependence shows up as a mismatch.
//     This lets the exact kernel logic be verified against the oracle without a GPU.
#pragma once
#include "rollout_io.h"

#ifdef DIAL_EMU
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#define DIAL_DEV inline
#define DIAL_UNROLL


// ---- SPMD register values for the register-resident sections (rows-in-lanes linear algebra).
// On the GPU a `vfloat` is one VGPR (a float per lane) and bcast() is v_readlane; in the emulator it is
// an explicit 64-wide vector and every operator works lane-wise, so the same source runs in lock step.
struct vbool { bool x[64]; };
struct vfloat {
  float x[64];
};
inline vfloat vsplat(float v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = v; return r; }
inline vfloat operator+(const vfloat& a, const vfloat& b) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = a.x[l] + b.x[l]; return r; }
inline vfloat operator-(const vfloat& a, const vfloat& b) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = a.x[l] - b.x[l]; return r; }
inline vfloat operator*(const vfloat& a, const vfloat& b) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = a.x[l] * b.x[l]; return r; }
inline vfloat operator*(const vfloat& a, float b) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = a.x[l] * b; return r; }
inline vfloat vfma(const vfloat& a, const vfloat& b, const vfloat& c) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = std::fma(a.x[l], b.x[l], c.x[l]); return r; }
inline vfloat vsel(const vbool& c, const vfloat& a, const vfloat& b) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = c.x[l] ? a.x[l] : b.x[l]; return r; }
inline vbool vlt0(const vfloat& a) { vbool r; for (int l = 0; l < 64; l++) r.x[l] = a.x[l] < 0.f; return r; }
inline vbool veq0(const vfloat& a) { vbool r; for (int l = 0; l < 64; l++) r.x[l] = a.x[l] == 0.f; return r; }
inline float bcast(const vfloat& v, int lane) { return v.x[lane]; }
inline float lane_val(const vfloat& v, int lane) { return v.x[lane]; }
// two consecutive floats from an 8-byte aligned address (one ds_read_b64 on the GPU)
inline void load2(const float* p, float& a, float& b) { a = p[0]; b = p[1]; }
inline void load4(const float* p, float& a, float& b, float& c, float& d) { a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
// two / four consecutive floats to an 8- / 16-byte aligned address (one ds_write_b64 / b128 on the GPU)
inline void store2(float* p, float a, float b) { p[0] = a; p[1] = b; }
inline void store4(float* p, float a, float b, float c, float d) { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
inline float fast_rsqrt(float x) { return 1.0f / std::sqrt(x); }
inline float fast_rcp(float x) { return 1.0f / x; }
inline vfloat vrcp(const vfloat& a) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = 1.0f / a.x[l]; return r; }   // lane-wise fast_rcp

// The GPU's wave reductions are DPP butterflies with a fixed association (HIP section below: wave_sum_dpp).  The emulator sums
// sequentially by default (its results are compared with the oracle at tolerance); with `tree_sums` it reproduces the GPU's
// association bit for bit -- that is what lets a test compare two LANE LAYOUTS of the same arithmetic for exact equality.
inline float emu_row_tree(const float* v) {   // one row of 16 lanes: quad_perm xor 1, xor 2, row_half_mirror, row_mirror
  float s1[16], s2[16], s3[16];
  for (int l = 0; l < 16; l++) s1[l] = v[l] + v[l ^ 1];
  for (int l = 0; l < 16; l++) s2[l] = s1[l] + s1[l ^ 2];
  for (int l = 0; l < 16; l++) s3[l] = s2[l] + s2[(l & 8) | (7 - (l & 7))];
  return s3[15] + s3[0];
}
inline float emu_tree64(const float* v) {     // ... row_bcast15 (rows 1, 3 += rows 0, 2), row_bcast31 (rows 2, 3 += row 1): lane 63
  const float r0 = emu_row_tree(v), r1 = emu_row_tree(v + 16), r2 = emu_row_tree(v + 32), r3 = emu_row_tree(v + 48);
  return (r3 + r2) + (r1 + r0);
}
inline float emu_tree32(const float* v) {     // half-wave sum (WaveH): the two rows combined through v_permlane16_swap
  return emu_row_tree(v) + emu_row_tree(v + 16);
}

#define DIAL_MARK(w, id)
#define DIAL_LANE_SCOPE(w)
#define DIAL_LANE_SCOPE_IF(cond, w)
struct Wave {
  static constexpr bool half2 = false;   // (WaveH below: two samples per wavefront, this object is one 32-lane half)
  bool tree_sums = false;                // wave sums in the GPU's association (see emu_row_tree)
  bool launder = false;   // (GPU only: opaque lane id per step, see the HIP Wave)
  // the control step's rows of the launch's output tensors, for the phases that store what they produce (Dims::pre_ctrl)
  const dial::RolloutIO* out_io = nullptr;
  int out_row = 0;
  const float* jrow = nullptr;   // this control step's row of the joint-target table (see the HIP Wave)
  float* lds = nullptr;
  int lds_words = 0;
  bool check_races = false;
  int races = 0;
  int phase = 0;

  // DIAL_EMU_NANCHECK (tests/wave_emu/emu.cpp): report the first phase after which a non-finite value sits in LDS
  bool nancheck = false, nan_seen = false;
  void scan_nonfinite() {
    if (!nancheck || nan_seen) return;
    for (int k = 0; k < lds_words; k++)
      if (!std::isfinite(lds[k])) {
        std::fprintf(stderr, "[wave_emu] first non-finite LDS word after phase #%d: word %d = %g\n", phase, k, (double)lds[k]);
        nan_seen = true;
        return;
      }
  }
  template <class F>
  void items(int count, F f) {
    phase++;
    if (!check_races) {
      for (int i = 0; i < count; i++) f(i);
      scan_nonfinite();
      return;
    }
    std::vector<float> snap(lds, lds + lds_words);
    for (int i = 0; i < count; i++) f(i);
    std::vector<float> fwd(lds, lds + lds_words);
    std::memcpy(lds, snap.data(), sizeof(float) * lds_words);
    for (int i = count - 1; i >= 0; i--) f(i);
    if (std::memcmp(fwd.data(), lds, sizeof(float) * lds_words) != 0) {
      if (races < 10) {
        int w = 0;
        for (; w < lds_words; w++)
          if (std::memcmp(&fwd[w], &lds[w], 4) != 0) break;
        std::fprintf(stderr, "[wave_emu] intra-phase dependence in phase #%d (first differing LDS word %d)\n", phase, w);
      }
      races++;
    }
  }
  template <class F>
  float sum(int count, F f) {
    if (tree_sums) {   // lane-strided partial sums first, as the GPU does (HIP Wave::sum)
      float v[64];
      for (int l = 0; l < 64; l++) { v[l] = l < count ? f(l) : 0.f; for (int i = l + 64; i < count; i += 64) v[l] += f(i); }
      return emu_tree64(v);
    }
    float s = 0.f;
    for (int i = 0; i < count; i++) s += f(i);
    return s;
  }
  // three sums at once; f(i, a, b, c) adds its contribution to a, b, c
  template <class F>
  void sum3(int count, F f, float& a, float& b, float& c) {
    if (tree_sums) {
      float va[64], vb[64], vc[64];
      for (int l = 0; l < 64; l++) {
        va[l] = vb[l] = vc[l] = 0.f;
        if (l < count) f(l, va[l], vb[l], vc[l]);
        for (int i = l + 64; i < count; i += 64) { float x = 0.f, y = 0.f, z = 0.f; f(i, x, y, z); va[l] += x; vb[l] += y; vc[l] += z; }
      }
      a = emu_tree64(va); b = emu_tree64(vb); c = emu_tree64(vc);
      return;
    }
    a = b = c = 0.f;
    for (int i = 0; i < count; i++) {
      float x = 0.f, y = 0.f, z = 0.f;
      f(i, x, y, z);
      a += x; b += y; c += z;
    }
  }
  template <class F>
  float maxv(int count, F f) {
    float s = -INFINITY;
    for (int i = 0; i < count; i++) { float v = f(i); s = v > s ? v : s; }
    return s;
  }
  // SPMD helpers: value computed per lane; lane-index predicates
  template <class F>
  vfloat per_lane(F f) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = f(l); return r; }
  template <class F>
  vfloat per_lane_r(F f) { return per_lane(f); }
  // K values per lane at once: f(lane, float out[K])
  template <int K, class F>
  void per_lane_n(vfloat (&out)[K], F f) {
    for (int l = 0; l < 64; l++) { float o[K]; f(l, o); for (int k = 0; k < K; k++) out[k].x[l] = o[k]; }
  }
  // value of the lane whose index differs in bit 0 / bit 1 (neighbours inside a quad; DPP quad_perm on the GPU)
  vfloat quad_xor1(const vfloat& v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = v.x[l ^ 1]; return r; }
  vfloat quad_xor2(const vfloat& v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = v.x[l ^ 2]; return r; }
  // four consecutive floats from a per-lane, 16-byte aligned address (one ds_read_b128 on the GPU)
  template <class F>
  void per_lane4(F f, vfloat& a, vfloat& b, vfloat& c, vfloat& d) {
    for (int l = 0; l < 64; l++) { const float* p = f(l); a.x[l] = p[0]; b.x[l] = p[1]; c.x[l] = p[2]; d.x[l] = p[3]; }
  }
  template <class F>
  void per_lane4_r(F f, vfloat& a, vfloat& b, vfloat& c, vfloat& d) { per_lane4(f, a, b, c, d); }
  // value of the lane N below / above inside the aligned row of 16 lanes (0 where the row ends; DPP row_shr / row_shl on the GPU)
  template <int N>
  vfloat row_shr(const vfloat& v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = (l & 15) >= N ? v.x[l - N] : 0.f; return r; }
  template <int N>
  vfloat row_shl(const vfloat& v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = (l & 15) + N <= 15 ? v.x[l + N] : 0.f; return r; }
  // row_shr / row_shl for the lower half of every row only: lanes 0..7 receive, lanes 8..15 read 0 (DPP bank_mask 0x3)
  template <int N>
  vfloat row_shr_lo(const vfloat& v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = ((l & 15) < 8 && (l & 15) >= N) ? v.x[l - N] : 0.f; return r; }
  template <int N>
  vfloat row_shl_lo(const vfloat& v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = (l & 15) < 8 ? v.x[l + N] : 0.f; return r; }
  // value of lane N of the own row of 16 lanes (DPP row_newbcast on the GPU)
  template <int N>
  vfloat row_bcast(const vfloat& v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = v.x[(l & ~15) + N]; return r; }
  // the lanes where a predicate holds, as a bit mask (GPU: ballot)
  unsigned long long mask(const vbool& c) const { unsigned long long b = 0; for (int l = 0; l < 64; l++) b |= (unsigned long long)(c.x[l] ? 1 : 0) << l; return b; }
  vbool lane_gt(int k) const { vbool r; for (int l = 0; l < 64; l++) r.x[l] = l > k; return r; }
  vbool lane_eq(int k) const { vbool r; for (int l = 0; l < 64; l++) r.x[l] = l == k; return r; }
  vbool lane_lt(int k) const { vbool r; for (int l = 0; l < 64; l++) r.x[l] = l < k; return r; }
  void begin_region() {}
  void set_rollout(int) {}
  void redraw_priority() {}
  int work = 0;   // (GPU: solver iterations of this rollout so far, see the HIP Wave)
  // reverse the first n lanes: result[l] = v[n-1-l] for l < n (0 elsewhere)
  vfloat lane_reverse(const vfloat& v, int n) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = l < n ? v.x[n - 1 - l] : 0.f; return r; }
  // plain LDS fence between SPMD stores and later loads (the GPU needs the wait, the emulator nothing)
  void fence() {}
  // sum within each aligned group of 16 lanes, result replicated in every lane of the group
  vfloat row16_sum(const vfloat& v) {
    vfloat r;
    for (int g = 0; g < 4; g++) {
      float t = 0.f;
      for (int l = 0; l < 16; l++) t += v.x[16 * g + l];
      for (int l = 0; l < 16; l++) r.x[16 * g + l] = t;
    }
    return r;
  }
  // three independent 16-lane sums at once (the DPP stages interleave on the GPU)
  void row16_sum3(vfloat& a, vfloat& b, vfloat& c) { a = row16_sum(a); b = row16_sum(b); c = row16_sum(c); }
  // stream compaction: list[rank] = lane index of every lane < count whose predicate holds; returns how many
  template <class F>
  int compact(int count, F pred, float* list) {
    int n = 0;
    for (int l = 0; l < count && l < 64; l++) if (pred(l)) list[n++] = (float)l;
    return n;
  }
  // K independent 16-lane sums (results replicated in every lane of the group)
  template <int K>
  void row16_sumN(vfloat (&v)[K]) { for (int k = 0; k < K; k++) v[k] = row16_sum(v[k]); }
  // K independent sums within every aligned group of EIGHT lanes, replicated in the group's lanes (GPU: quad_perm xor 1, xor 2,
  // row_half_mirror -- the association (((a0+a1)+(a2+a3)) + ((a7+a6)+(a5+a4))) is reproduced here)
  template <int K>
  void seg8_sumN(vfloat (&v)[K]) {
    for (int k = 0; k < K; k++) {
      vfloat s1, s2, s3;
      for (int l = 0; l < 64; l++) s1.x[l] = v[k].x[l] + v[k].x[l ^ 1];
      for (int l = 0; l < 64; l++) s2.x[l] = s1.x[l] + s1.x[l ^ 2];
      for (int l = 0; l < 64; l++) s3.x[l] = s2.x[l] + s2.x[(l & ~7) | (7 - (l & 7))];
      v[k] = s3;
    }
  }
  // value of lane src(l), per lane (GPU: ds_bpermute); src outside 0..63 is the caller's bug
  template <class F>
  vfloat gather64(const vfloat& v, F src) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = v.x[src(l) & 63]; return r; }
  // wave-uniform sum of a register value over all 64 lanes (idle lanes must hold 0)
  float vsum(const vfloat& v) { if (tree_sums) return emu_tree64(v.x); float s = 0.f; for (int l = 0; l < 64; l++) s += v.x[l]; return s; }
  template <int K>
  void vsumN(vfloat (&v)[K], float (&out)[K]) { for (int k = 0; k < K; k++) out[k] = vsum(v[k]); }
  // value of lane K, as a wave-uniform scalar (same as the free function bcast; WaveH: of the own half)
  template <int K>
  float bc(const vfloat& v) { return v.x[K]; }
  // ---- DPP-operand broadcasts (used by the 32-lane layouts of WaveH and by the register L D L^T of every robot: solver_reg2.h)
  // X = the half's even row in both of its rows, Y = its odd row in both (GPU: one v_permlane16_swap)
  void dup_rows(const vfloat& v, vfloat& X, vfloat& Y) {
    for (int l = 0; l < 64; l++) { X.x[l] = v.x[(l & 32) | (l & 15)]; Y.x[l] = v.x[(l & 32) | 16 | (l & 15)]; }
  }
  // LO = the lower 32 lanes of v in BOTH halves of the wavefront, HI = its upper 32 lanes in both (GPU: one v_permlane32_swap).
  // One-sample layouts only: dup_halves + dup_rows put any lane of v within reach of every lane's row_newbcast.
  void dup_halves(const vfloat& v, vfloat& LO, vfloat& HI) {
    for (int l = 0; l < 64; l++) { LO.x[l] = v.x[l & 31]; HI.x[l] = v.x[32 | (l & 31)]; }
  }
  // lane K of the own ROW, as a (half-uniform) scalar -- for values every row holds a copy of (GPU: one DPP row_newbcast)
  template <int K>
  float rowbc(const vfloat& v) { return v.x[K]; }
  template <int K>
  vfloat mul_pick(const vfloat& X, const vfloat& Y, const vfloat& other) { return other * pick<K>(X, Y); }
  // acc +- other * (lane K of the half, from its duplicated rows X | Y) and 1 / that lane: on the GPU's product build ONE instruction
  // each -- the DPP row broadcast is an operand modifier of v_fmac_f32 / v_rcp_f32 (HIP WaveH below)
  template <int K>
  vfloat pick(const vfloat& X, const vfloat& Y) { if constexpr (K < 16) return row_bcast<K>(X); else return row_bcast<K - 16>(Y); }
  template <int K>
  vfloat fma_pick(const vfloat& acc, const vfloat& X, const vfloat& Y, const vfloat& other) { return acc + other * pick<K>(X, Y); }
  template <int K>
  vfloat fnma_pick(const vfloat& acc, const vfloat& X, const vfloat& Y, const vfloat& other) { return acc - other * pick<K>(X, Y); }
  template <int K>
  vfloat rcp_pick(const vfloat& X, const vfloat& Y) { return vrcp(pick<K>(X, Y)); }
};

// ---- WaveH: TWO samples per wavefront, one per 32-lane half (HIP section below).  The emulator runs ONE half: logical lanes
// 0..31 (lanes 32..63 of a vfloat mirror them), so that the 32-lane layouts of smooth_quad2.h / solver_reg2.h can be compared
// with the oracle -- and bit for bit with the 64-lane layouts -- without a GPU.  Sums are always in the GPU's association.
struct WaveH : Wave {
  static constexpr bool half2 = true;
  template <class F>
  float sum(int count, F f) {
    float v[32];
    for (int l = 0; l < 32; l++) { v[l] = l < count ? f(l) : 0.f; for (int i = l + 32; i < count; i += 32) v[l] += f(i); }
    return emu_tree32(v);
  }
  template <class F>
  void sum3(int count, F f, float& a, float& b, float& c) {
    float va[32], vb[32], vc[32];
    for (int l = 0; l < 32; l++) {
      va[l] = vb[l] = vc[l] = 0.f;
      if (l < count) f(l, va[l], vb[l], vc[l]);
      for (int i = l + 32; i < count; i += 32) { float x = 0.f, y = 0.f, z = 0.f; f(i, x, y, z); va[l] += x; vb[l] += y; vc[l] += z; }
    }
    a = emu_tree32(va); b = emu_tree32(vb); c = emu_tree32(vc);
  }
  template <class F>
  vfloat per_lane(F f) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = f(l & 31); return r; }
  template <class F>
  vfloat per_lane_r(F f) { return per_lane(f); }
  template <int K, class F>
  void per_lane_n(vfloat (&out)[K], F f) {
    for (int l = 0; l < 64; l++) { float o[K]; f(l & 31, o); for (int k = 0; k < K; k++) out[k].x[l] = o[k]; }
  }
  template <class F>
  void per_lane4(F f, vfloat& a, vfloat& b, vfloat& c, vfloat& d) {
    for (int l = 0; l < 64; l++) { const float* p = f(l & 31); a.x[l] = p[0]; b.x[l] = p[1]; c.x[l] = p[2]; d.x[l] = p[3]; }
  }
  unsigned long long mask(const vbool& c) const { unsigned long long b = 0; for (int l = 0; l < 32; l++) b |= (unsigned long long)(c.x[l] ? 1 : 0) << l; return b; }
  vbool lane_gt(int k) const { vbool r; for (int l = 0; l < 64; l++) r.x[l] = (l & 31) > k; return r; }
  vbool lane_eq(int k) const { vbool r; for (int l = 0; l < 64; l++) r.x[l] = (l & 31) == k; return r; }
  vbool lane_lt(int k) const { vbool r; for (int l = 0; l < 64; l++) r.x[l] = (l & 31) < k; return r; }
  vfloat lane_reverse(const vfloat& v, int n) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = (l & 31) < n ? v.x[(l & 32) + n - 1 - (l & 31)] : 0.f; return r; }
  float vsum(const vfloat& v) { return emu_tree32(v.x); }
  template <int K>
  void vsumN(vfloat (&v)[K], float (&out)[K]) { for (int k = 0; k < K; k++) out[k] = vsum(v[k]); }
  // value of (logical) lane src(l) of the own half, per lane (GPU: ds_bpermute)
  template <class F>
  vfloat gather(const vfloat& v, F src) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = v.x[(l & 32) | (src(l & 31) & 31)]; return r; }
  // lane 3 of the own group of 8 lanes, to the whole group (GPU: two row_newbcast + select)
  vfloat grp8_bcast3(const vfloat& v) { vfloat r; for (int l = 0; l < 64; l++) r.x[l] = v.x[(l & ~7) | 3]; return r; }
};

#else  // ------------------------------------------------------------------ HIP / gfx950
#include <hip/hip_runtime.h>
#define DIAL_DEV __device__ __forceinline__

namespace dialwave {
// DPP controls (gfx9/CDNA): quad_perm 0x00-0xff, row_shr n 0x110+n, row_mirror 0x140,
// row_half_mirror 0x141, row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, BOUND);
  return v + __builtin_bit_cast(float, moved);
}
// Full-wave sum.  After the six steps lane 63 holds the total; broadcast it as a scalar.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add<0xb1>(v);         // quad_perm [1,0,3,2]
  v = dpp_add<0x4e>(v);         // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);        // row_half_mirror
  v = dpp_add<0x140>(v);        // row_mirror  -> every lane of a 16-row holds the row sum
  v = dpp_add<0x142, 0xa>(v);   // row_bcast15 -> rows 1,3 += rows 0,2
  v = dpp_add<0x143, 0xc>(v);   // row_bcast31 -> rows 2,3 += row 1(=0+1)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum_shfl(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float wave_max_shfl(float v) {
  for (int o = 32; o > 0; o >>= 1) { float t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
#ifdef DIAL_NO_DPP
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_shfl(v); }
#else
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
#endif
}  // namespace dialwave


// ---- SPMD register values (see the emulator section): one VGPR per vfloat, v_readlane broadcasts.
using vfloat = float;
using vbool = bool;
__device__ __forceinline__ vfloat vsplat(float v) { return v; }
__device__ __forceinline__ vfloat vfma(vfloat a, vfloat b, vfloat c) { return __builtin_fmaf(a, b, c); }   // single rounding
__device__ __forceinline__ vfloat vsel(vbool c, vfloat a, vfloat b) { return c ? a : b; }
__device__ __forceinline__ vbool vlt0(vfloat a) { return a < 0.f; }
__device__ __forceinline__ vbool veq0(vfloat a) { return a == 0.f; }
__device__ __forceinline__ float bcast(vfloat v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float lane_val(vfloat v, int) { return v; }
__device__ __forceinline__ void load2(const float* p, float& a, float& b) { const float2 t = *reinterpret_cast<const float2*>(p); a = t.x; b = t.y; }
__device__ __forceinline__ void load4(const float* p, float& a, float& b, float& c, float& d) { const float4 t = *reinterpret_cast<const float4*>(p); a = t.x; b = t.y; c = t.z; d = t.w; }
__device__ __forceinline__ void store2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ vfloat vrcp(vfloat a) { return __builtin_amdgcn_rcpf(a); }

// An opaque copy of the lane id for the enclosing scope (restored at its end): what is derived from it -- role masks, table
// addresses -- stays inside the scope instead of being hoisted out of the T-step loop as long-lived SGPR masks / address VGPRs.
template <class W>
struct LaneScope {
  W& w;
  int keep, keep_r;
  __device__ __forceinline__ explicit LaneScope(W& w_) : w(w_), keep(w_.lane), keep_r(w_.lane_r) {
    int lq = w.lane;
    asm volatile("" : "+v"(lq));
    w.lane = lq;
    w.lane_r = lq;
  }
  __device__ __forceinline__ ~LaneScope() { w.lane = keep; w.lane_r = keep_r; }
};
#define DIAL_LANE_SCOPE(w) LaneScope<std::remove_reference_t<decltype(w)>> dial_lane_scope_(w)
// the same, compiled in only where COND (a compile-time condition) holds
template <bool COND, class W>
struct LaneScopeIf {
  __device__ __forceinline__ explicit LaneScopeIf(W&) {}
};
template <class W>
struct LaneScopeIf<true, W> : LaneScope<W> {
  __device__ __forceinline__ explicit LaneScopeIf(W& w_) : LaneScope<W>(w_) {}
};
#define DIAL_LANE_SCOPE_IF(cond, w) LaneScopeIf<(cond), std::remove_reference_t<decltype(w)>> dial_lane_scope_if_(w)
#ifdef DIAL_PROFILE
#define DIAL_NSEC 32
#define DIAL_MARK(w, id) (w).mark(id)
#elif defined(DIAL_ISA_MARKS)
// ISA probes (tools/isa/probe.sh -DDIAL_ISA_MARKS): a comment line per section boundary in the assembly, so that tools/isa/section_hist.py
// can count the instructions of every section by class; no instruction is emitted
#define DIAL_MARK(w, id) asm volatile("; DIAL_MARK %0" ::"n"(id))
#else
#define DIAL_MARK(w, id)
#endif
struct Wave {
  static constexpr bool half2 = false;   // (WaveH below: two samples per wavefront)
  int lane;
  // The control step's rows of the launch's output tensors (row = rollout x T + step), for the phases that store what they
  // produce (Dims::pre_ctrl; rollout_driver.h sets them before env_step): one kernarg pointer + one index, not four row pointers
  // kept live through the solver.  WaveH: the index is a per-lane value (each half has its own rollout).
  const dial::RolloutIO* out_io = nullptr;
  int out_row = 0;
  // this control step's row of the rollout's joint-target table (Ws::jtab), or nullptr: the position stage's actuation lanes then
  // run act2tau themselves (base_env.py:51-66) instead of reading s.ctrl from a phase of its own
  const float* jrow = nullptr;
#ifdef DIAL_PROFILE
  // accumulators live in LDS (written by lane 0) so that the profiling build does not eat the scalar
  // registers the measured code is short of
  unsigned long long tprev = 0;
  unsigned long long* acc = nullptr;
  __device__ __forceinline__ void mark(int id) {
    unsigned long long t = __builtin_readcyclecounter();
    if (lane == 0 && acc) acc[id] += t - tprev;
    tprev = t;
  }
#endif
  // Phase boundary.  DS (LDS) instructions of ONE wave execute in order, so a later ds_read observes an
  // earlier ds_write of any lane of the same wave without waiting for the write to retire: all that is
  // needed is that the compiler does not move LDS accesses across the boundary (wavefront-scope fence +
  // scheduling barrier).  -DDIAL_BLOCK_SYNC restores the conservative workgroup barrier.
  __device__ __forceinline__ void sync() {
#ifdef DIAL_BLOCK_SYNC
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
  }

  template <class F>
  __device__ __forceinline__ void items(int count, F f) {
    for (int i = lane; i < count; i += 64) f(i);
    sync();
  }
  // (count may exceed the wavefront: the crate scene sums over 220 constraint rows -- lane-strided partial sums first)
  template <class F>
  __device__ __forceinline__ float sum(int count, F f) {
    float v = lane < count ? f(lane) : 0.f;
    for (int i = lane + 64; i < count; i += 64) v += f(i);
    return dialwave::wave_sum(v);
  }
  template <class F>
  __device__ __forceinline__ void sum3(int count, F f, float& a, float& b, float& c) {
    float x = 0.f, y = 0.f, z = 0.f;
    if (lane < count) f(lane, x, y, z);
    for (int i = lane + 64; i < count; i += 64) {
      float x2 = 0.f, y2 = 0.f, z2 = 0.f;
      f(i, x2, y2, z2);
      x += x2; y += y2; z += z2;
    }
    vfloat t3[3] = {x, y, z};
    float r3[3];
    vsumN(t3, r3);   // the three reductions with their DPP stages interleaved
    a = r3[0]; b = r3[1]; c = r3[2];
  }
  template <class F>
  __device__ __forceinline__ float maxv(int count, F f) {
    float v = lane < count ? f(lane) : -INFINITY;
    for (int i = lane + 64; i < count; i += 64) { const float u = f(i); v = u > v ? u : v; }
    return dialwave::wave_max_shfl(v);
  }
  template <class F>
  __device__ __forceinline__ vfloat per_lane(F f) { return f(lane); }
  // same with the region-laundered lane id: address arithmetic derived from it stays inside the region instead of
  // being hoisted out of the T-step loop as dozens of long-lived address VGPRs (use where that causes spills)
  template <class F>
  __device__ __forceinline__ vfloat per_lane_r(F f) { return f(lane_r); }
  template <int K, class F>
  __device__ __forceinline__ void per_lane_n(vfloat (&out)[K], F f) {
    float o[K];
    f(lane, o);
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = o[k];
  }
  __device__ __forceinline__ vfloat quad_xor1(vfloat v) {   // quad_perm [1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));
  }
  __device__ __forceinline__ vfloat quad_xor2(vfloat v) {   // quad_perm [2,3,0,1]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));
  }
  template <int N>
  __device__ __forceinline__ vfloat row_shr(vfloat v) {   // DPP row_shr:N, lanes without a source read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
  }
  template <int N>
  __device__ __forceinline__ vfloat row_shl(vfloat v) {   // DPP row_shl:N
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
  }
  template <int N>
  __device__ __forceinline__ vfloat row_shr_lo(vfloat v) {   // bank_mask 0x3: lanes 8..15 of every row keep `old` = 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0x3, true));
  }
  template <int N>
  __device__ __forceinline__ vfloat row_shl_lo(vfloat v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0x3, true));
  }
  template <int N>
  __device__ __forceinline__ vfloat row_bcast(vfloat v) {   // DPP row_newbcast:N (gfx90a+): lane N of the own row
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xf, 0xf, true));
  }
  template <class F>
  __device__ __forceinline__ void per_lane4(F f, vfloat& a, vfloat& b, vfloat& c, vfloat& d) {
    const float4 t = *reinterpret_cast<const float4*>(f(lane));
    a = t.x; b = t.y; c = t.z; d = t.w;
  }
  // (address from the region-laundered lane id, see per_lane_r)
  template <class F>
  __device__ __forceinline__ void per_lane4_r(F f, vfloat& a, vfloat& b, vfloat& c, vfloat& d) {
    const float4 t = *reinterpret_cast<const float4*>(f(lane_r));
    a = t.x; b = t.y; c = t.z; d = t.w;
  }
  // Lane-index predicates compare against `lane_r`, a copy of the lane id that begin_region() launders
  // through an empty asm.  The comparisons are loop invariants of the T-step rollout loop; left alone, LICM
  // hoists dozens of them out of it as 64-bit SGPR masks that live for the whole kernel and get spilled,
  // while the v_readlane broadcasts of the register-resident linear algebra starve for scalar registers.
  // Refreshing lane_r once per region (one solve, one line search) keeps the masks short-lived.
  int lane_r;
  __device__ __forceinline__ void begin_region() { int l = lane; asm volatile("" : "+v"(l)); lane_r = l; }
  // Opaque copy of the lane id once per control step / physics frame (rollout_driver.h, rollout_body.h): set by the kernel for
  // the instantiations whose hoisted lane-derived addresses would not fit the register budget (generic feature set; the
  // 128-VGPR large-batch build of the Go2).  A compile-time constant after inlining.
  bool launder = false;
  // Issue priority, re-drawn pseudo-randomly (4 levels, hash of rollout index and draw count) twice per physics step.
  // The SIMD's arbiter serves the OLDEST ready wavefront first: of the two or three wavefronts that share a SIMD the
  // oldest runs at its solo pace and the youngest on what is left, so rollouts of equal length finish 410 ... 615 us
  // apart and the launch lasts as long as the youngest.  Random priorities make the sharing fair over a rollout: the
  // wavefronts of a SIMD finish together (437 ... 570 us) and the launch is 9 % shorter.
  // Results do not depend on it.  -DDIAL_FIXED_PRIORITY keeps the hardware default (measurement switch).
  unsigned prio_seed = 0, prio_ctr = 0;
  bool prio_held = false;
  // Rollouts of data-dependent length (elliptic solver: Newton iterations until convergence): fair sharing makes the launch as
  // long as the slowest rollout AT ITS FAIR SHARE of its SIMD.  Instead a rollout that is behind -- more solver iterations per
  // control step so far than the launch's running average (RolloutIO::work_stat) -- gets the higher issue priority: it runs
  // closer to its solo pace while its faster SIMD-mates, which have slack, give way.  `work`: this rollout's Newton iterations.
  int work = 0;
  int prio_level = -1;   // >= 0: the lag-based level, set once per control step (rollout_driver.h); redraws re-apply it
  __device__ __forceinline__ void apply_level(int L) {
    if (L <= 0) __builtin_amdgcn_s_setprio(0);
    else if (L == 1) __builtin_amdgcn_s_setprio(1);
    else if (L == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
  }
  __device__ __forceinline__ void set_rollout(int n) { prio_seed = (unsigned)n * 2654435761u; prio_ctr = 0; prio_held = false; work = 0; prio_level = -1; }
  __device__ __forceinline__ void hold_priority(int) { prio_held = true; __builtin_amdgcn_s_setprio(3); }   // relay pieces
  __device__ __forceinline__ void redraw_priority() {
#ifndef DIAL_FIXED_PRIORITY
    if (prio_held) return;
    if (prio_level >= 0) { apply_level(prio_level); return; }
    prio_ctr++;
    const unsigned h = (prio_seed + prio_ctr * 0x9E3779B1u) >> 30;
    if (h == 0) __builtin_amdgcn_s_setprio(0);
    else if (h == 1) __builtin_amdgcn_s_setprio(1);
    else if (h == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
#endif
  }
  __device__ __forceinline__ unsigned long long mask(vbool c) const { return __builtin_amdgcn_ballot_w64(c); }
  __device__ __forceinline__ vbool lane_gt(int k) const { return lane_r > k; }
  __device__ __forceinline__ vbool lane_eq(int k) const { return lane_r == k; }
  __device__ __forceinline__ vbool lane_lt(int k) const { return lane_r < k; }
  __device__ __forceinline__ vfloat lane_reverse(vfloat v, int n) {
    const float r = __shfl(v, n - 1 - lane, 64);   // ds_bpermute_b32
    return lane < n ? r : 0.f;
  }
  __device__ __forceinline__ void fence() { sync(); }
  __device__ __forceinline__ float vsum(vfloat v) { return dialwave::wave_sum(v); }
  // K independent wave sums with their DPP stages interleaved (a dependent DPP add costs ~16 cycles, the K
  // chains hide each other's latency); results are wave-uniform
  template <int K>
  __device__ __forceinline__ void vsumN(vfloat (&v)[K], float (&out)[K]) {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0xb1>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x4e>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x141>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x140>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x142, 0xa>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x143, 0xc>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), 63));
  }
  __device__ __forceinline__ vfloat row16_sum(vfloat v) {
    v = dialwave::dpp_add<0xb1>(v);    // quad_perm [1,0,3,2]
    v = dialwave::dpp_add<0x4e>(v);    // quad_perm [2,3,0,1]
    v = dialwave::dpp_add<0x141>(v);   // row_half_mirror
    v = dialwave::dpp_add<0x140>(v);   // row_mirror
    return v;
  }
  template <class F>
  __device__ __forceinline__ int compact(int count, F pred, float* list) {
    const bool p = lane < count && pred(lane);
    const unsigned long long b = __builtin_amdgcn_ballot_w64(p);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    if (p) list[rank] = (float)lane;
    sync();
    return __builtin_popcountll(b);
  }
  template <int K>
  __device__ __forceinline__ void row16_sumN(vfloat (&v)[K]) {   // K chains, DPP stages interleaved
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0xb1>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x4e>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x141>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x140>(v[k]);
  }
  template <int K>
  __device__ __forceinline__ void seg8_sumN(vfloat (&v)[K]) {   // sums within aligned groups of eight lanes, K chains interleaved
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0xb1>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x4e>(v[k]);
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = dialwave::dpp_add<0x141>(v[k]);
  }
  template <class F>
  __device__ __forceinline__ vfloat gather64(vfloat v, F src) { return __shfl(v, src(lane) & 63, 64); }   // ds_bpermute_b32
  __device__ __forceinline__ void row16_sum3(vfloat& a, vfloat& b, vfloat& c) {   // stage-interleaved: no DPP hazard stalls
    a = dialwave::dpp_add<0xb1>(a); b = dialwave::dpp_add<0xb1>(b); c = dialwave::dpp_add<0xb1>(c);
    a = dialwave::dpp_add<0x4e>(a); b = dialwave::dpp_add<0x4e>(b); c = dialwave::dpp_add<0x4e>(c);
    a = dialwave::dpp_add<0x141>(a); b = dialwave::dpp_add<0x141>(b); c = dialwave::dpp_add<0x141>(c);
    a = dialwave::dpp_add<0x140>(a); b = dialwave::dpp_add<0x140>(b); c = dialwave::dpp_add<0x140>(c);
  }
  template <int K>
  __device__ __forceinline__ float bc(vfloat v) { return bcast(v, K); }
  // ---- DPP-operand broadcasts: X = every 32-lane half's even row in both of its rows, Y = its odd row (one v_permlane16_swap, gfx950)
  __device__ __forceinline__ void dup_rows(vfloat v, vfloat& X, vfloat& Y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    X = __builtin_bit_cast(float, r0);
    Y = __builtin_bit_cast(float, r1);
#ifdef DIAL_FUSED_DPP
    asm("s_nop 1" : "+v"(X), "+v"(Y));   // (the hand-written DPP consumers of X | Y: see fma_pick)
#endif
  }
  // LO = the lower 32 lanes of v in both halves of the wavefront, HI = its upper 32 lanes in both (one v_permlane32_swap, gfx950)
  __device__ __forceinline__ void dup_halves(vfloat v, vfloat& LO, vfloat& HI) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const unsigned r0 = r[0], r1 = r[1];
    LO = __builtin_bit_cast(float, r0);
    HI = __builtin_bit_cast(float, r1);
  }
  template <int K>
  __device__ __forceinline__ float rowbc(vfloat v) { return row_bcast<K>(v); }
  template <int K>
  __device__ __forceinline__ vfloat pick(vfloat X, vfloat Y) { if constexpr (K < 16) return row_bcast<K>(X); else return row_bcast<K - 16>(Y); }
  // acc +- other * (lane K of the half) and 1 / (lane K of the half).  hipcc keeps `v_mov_b32_dpp` + `v_fma_f32` apart (its DPP
  // combine does not see through the three-address FMA), so the product build (-DDIAL_FUSED_DPP) spells the fused instruction:
  // v_fmac_f32_dpp dst, src0 (the DPP operand, with its neg modifier), src1 -- the same single-rounding fma the compiler's
  // contraction produces, one VALU issue slot instead of two.  Inline asm is opaque to the hazard recogniser: a DPP source must not
  // have been written by one of the two preceding VALU instructions.  The sources are always the X | Y of dup_rows, which (in this
  // build) ends in an `s_nop 1` that every later use depends on; tools/isa/check_dpp_hazards.py verifies the emitted ISA.  The build without
  // contraction (libdialhip_ieee.so) keeps the plain expression: there the multiply and the add round separately, as in the
  // one-sample kernel it is compared with bit for bit.
#ifdef DIAL_FUSED_DPP
#define DIAL_DPP_TAIL " row_newbcast:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
  template <int K>
  __device__ __forceinline__ vfloat fma_pick(vfloat acc, vfloat X, vfloat Y, vfloat other) {
    asm("v_fmac_f32_dpp %0, %1, %2" DIAL_DPP_TAIL : "+v"(acc) : "v"(K < 16 ? X : Y), "v"(other), "n"(K & 15));
    return acc;
  }
  template <int K>
  __device__ __forceinline__ vfloat fnma_pick(vfloat acc, vfloat X, vfloat Y, vfloat other) {
    asm("v_fmac_f32_dpp %0, -%1, %2" DIAL_DPP_TAIL : "+v"(acc) : "v"(K < 16 ? X : Y), "v"(other), "n"(K & 15));
    return acc;
  }
  template <int K>
  __device__ __forceinline__ vfloat rcp_pick(vfloat X, vfloat Y) {
    vfloat r;
    asm("v_rcp_f32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(K < 16 ? X : Y), "n"(K & 15));
    return r;
  }
  template <int K>
  __device__ __forceinline__ vfloat mul_pick(vfloat X, vfloat Y, vfloat other) {   // other * (lane K): v_mul_f32_dpp
    vfloat r;
    asm("v_mul_f32_dpp %0, %1, %2" DIAL_DPP_TAIL : "=v"(r) : "v"(K < 16 ? X : Y), "v"(other), "n"(K & 15));
    return r;
  }
#undef DIAL_DPP_TAIL
#else
  template <int K>
  __device__ __forceinline__ vfloat mul_pick(vfloat X, vfloat Y, vfloat other) { return other * pick<K>(X, Y); }
  template <int K>
  __device__ __forceinline__ vfloat fma_pick(vfloat acc, vfloat X, vfloat Y, vfloat other) { return acc + other * pick<K>(X, Y); }
  template <int K>
  __device__ __forceinline__ vfloat fnma_pick(vfloat acc, vfloat X, vfloat Y, vfloat other) { return acc - other * pick<K>(X, Y); }
  template <int K>
  __device__ __forceinline__ vfloat rcp_pick(vfloat X, vfloat Y) { return vrcp(pick<K>(X, Y)); }
#endif
};

// ---- WaveH: TWO samples per wavefront.  Each 32-lane half owns one sample; the kernel body is the same per-lane program,
// `lane` is the LOGICAL lane 0..31 inside the half and every value the one-sample kernel keeps wave-uniform (reduction results,
// broadcast pivots, the solver's control flow) is simply a per-lane value that agrees within a half: where the two samples
// take different branches the hardware's EXEC mask does what it does for any divergent SIMT code.  What that needs is that no
// cross-lane operation leaves the half:
//   * reductions: the DPP butterfly inside each row of 16 lanes, then ONE v_permlane16_swap (gfx950) that puts the half's
//     even row next to its odd row in every lane -- no v_readlane, no SGPR;
//   * broadcasts: dup_rows (the same swap: X = the half's even row in both rows, Y = its odd row) + DPP row_newbcast, i.e. a
//     broadcast is a VGPR operand of the consuming instruction's DPP mov instead of a v_readlane -> SGPR -> VALU hazard chain;
//   * ballots are split per half; the LDS workspace base and every global row pointer are per-lane values.
// The lane layouts that need more than 32 lanes per sample in the one-sample kernel (smooth_quad.h: four DPP rows; solver_reg.h:
// dof lanes + contact lanes + three line-search groups) have 32-lane versions in smooth_quad2.h / solver_reg2.h.
struct WaveH : Wave {
  static constexpr bool half2 = true;
  int half;     // 0 / 1: which half of the wavefront this lane belongs to (a VGPR value)
  __device__ __forceinline__ void init(int tid) { lane = tid & 31; lane_r = lane; half = (tid >> 5) & 1; }
  template <class F>
  __device__ __forceinline__ void items(int count, F f) {
    for (int i = lane; i < count; i += 32) f(i);
    sync();
  }
  // even row + odd row of the own half, in every lane of the half
  static __device__ __forceinline__ float half_combine(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const unsigned r0 = r[0], r1 = r[1];   // (element temporaries: bit_cast straight from r[k] folds both to element 0, clang 22)
    return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
  }
  template <int K>
  __device__ __forceinline__ float bc(vfloat v) {
    static_assert(K >= 0 && K < 32, "logical lane");
    vfloat X, Y;
    dup_rows(v, X, Y);
    if constexpr (K < 16) return row_bcast<K>(X);
    else return row_bcast<K - 16>(Y);
  }
  template <class F>
  __device__ __forceinline__ vfloat gather(vfloat v, F src) { return __shfl(v, (src(lane) & 31) + 32 * half, 64); }   // ds_bpermute_b32
  __device__ __forceinline__ vfloat grp8_bcast3(vfloat v) {
    const vfloat a = row_bcast<3>(v), b = row_bcast<11>(v);
    return (lane & 8) ? b : a;
  }

  // Reductions whose result every lane USES (there is no v_readlane to make it wave-uniform): all lanes of the half must end
  // up with the same bits.  The butterfly is symmetric -- lane l forms v[l] + v[l ^ 1], lane l ^ 1 the same two operands the other
  // way round -- unless hipcc contracts the multiply that produced v into the first add: fma(a_l, b_l, round(a_l' b_l')) in one
  // lane and fma(a_l', b_l', round(a_l b_l)) in the other differ in the last bit, the lanes of a half then take different
  // branches of the solver and the rollout is garbage (the first GPU run of the product build; the build without contraction was
  // bit-identical to the one-sample kernel).  So the summands enter the butterfly through an opaque copy: no fusion across it.
  static __device__ __forceinline__ float opaque(float v) { asm("" : "+v"(v)); return v; }
  __device__ __forceinline__ vfloat row16_sum(vfloat v) { return Wave::row16_sum(opaque(v)); }
  template <int K>
  __device__ __forceinline__ void row16_sumN(vfloat (&v)[K]) {
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = opaque(v[k]);
    Wave::row16_sumN(v);
  }
  __device__ __forceinline__ void row16_sum3(vfloat& a, vfloat& b, vfloat& c) { a = opaque(a); b = opaque(b); c = opaque(c); Wave::row16_sum3(a, b, c); }
  __device__ __forceinline__ float vsum(vfloat v) { return half_combine(row16_sum(v)); }
  template <int K>
  __device__ __forceinline__ void vsumN(vfloat (&v)[K], float (&out)[K]) {
    row16_sumN(v);
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = half_combine(v[k]);
  }
  template <class F>
  __device__ __forceinline__ float sum(int count, F f) {
    float v = lane < count ? f(lane) : 0.f;
    for (int i = lane + 32; i < count; i += 32) v += f(i);
    return vsum(v);
  }
  template <class F>
  __device__ __forceinline__ void sum3(int count, F f, float& a, float& b, float& c) {
    float x = 0.f, y = 0.f, z = 0.f;
    if (lane < count) f(lane, x, y, z);
    for (int i = lane + 32; i < count; i += 32) {
      float x2 = 0.f, y2 = 0.f, z2 = 0.f;
      f(i, x2, y2, z2);
      x += x2; y += y2; z += z2;
    }
    vfloat t3[3] = {x, y, z};
    float r3[3];
    vsumN(t3, r3);
    a = r3[0]; b = r3[1]; c = r3[2];
  }
  // ballot of the own half, in the low 32 bits
  __device__ __forceinline__ unsigned long long mask(vbool c) const {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(c);
    return half ? (unsigned)(b >> 32) : (unsigned)b;
  }
  __device__ __forceinline__ vfloat lane_reverse(vfloat v, int n) {
    const float r = __shfl(v, ((n - 1 - lane) & 31) + 32 * half, 64);
    return lane < n ? r : 0.f;
  }
  // issue priority is a property of the wavefront: keyed by the pair's first rollout
  // (a priority held by the kernel -- rollout_kernel2: the odd wavefront of a batch -- survives the start of the rollout)
  __device__ __forceinline__ void set_rollout(int n) { const bool held = prio_held; Wave::set_rollout(__builtin_amdgcn_readfirstlane(n)); prio_held = held; }
  __device__ __forceinline__ void redraw_priority() {
#ifndef DIAL_FIXED_PRIORITY
    if (prio_held) return;
    prio_ctr++;
    const unsigned h = __builtin_amdgcn_readfirstlane((prio_seed + prio_ctr * 0x9E3779B1u) >> 30);
    if (h == 0) __builtin_amdgcn_s_setprio(0);
    else if (h == 1) __builtin_amdgcn_s_setprio(1);
    else if (h == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
#endif
  }
#ifdef DIAL_PROFILE
  __device__ __forceinline__ void mark(int id) {   // the lower half's lane 0 keeps the wavefront's section clock
    unsigned long long t = __builtin_readcyclecounter();
    if (lane == 0 && half == 0 && acc) acc[id] += t - tprev;
    tprev = t;
  }
#endif
};
#endif
