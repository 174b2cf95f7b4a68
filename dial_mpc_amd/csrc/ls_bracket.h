// ls_bracket.h -- the bracket bookkeeping of MJX's `solver._linesearch` on INTEGER keys.
//
// One wavefront owns the sample, so the line search's points (alpha, cost, slope) are wave-uniform.  gfx950 has no
// scalar floating-point ALU: written with floats, every comparison of the bracket update became a VALU v_cmp into a
// 64-bit lane mask + s_and_b64 with exec + s_cbranch -- ~20 dependent VALU->SALU->branch round trips per line-search
// iteration (build/go2.s, round 2), the longest serial stretch of the iteration.  Floats order like sign-magnitude
// integers, so each point is carried as four 32-bit words in SGPRs instead --
//     alpha   raw bits                          nalpha  raw bits of the point's own Newton step alpha - d0 / d1
//     cost    monotone key of the cost          d0      monotone key of the slope
// -- produced LANE-WISE where the point is evaluated (the three trial points of an iteration sit in three 16-lane
// groups) and broadcast with v_readlane.  The whole update is then s_cmp_*_i32 / s_cselect_b32 on the scalar unit, and
// the next iteration's trial steps are already there (lo.nalpha, hi.nalpha; only the mid-point needs one VALU add).
// fkey(a) < fkey(b)  <=>  a < b for all non-NaN floats (-0 and +0 share key 0, as they compare equal); the device code
// does not honour NaNs anywhere (-fno-honor-nans).  Same code in the host wave emulator.
//
// Reference: mujoco.mjx._src.solver._linesearch (both bracket rules, see include/dial_mpc.h DIAL_LS_*); the
// CPU checker restates the same rules with plain float comparisons.
#pragma once

namespace dial {

DIAL_DEV int fbits(float x) { return __builtin_bit_cast(int, x); }
DIAL_DEV float bitsf(int b) { return __builtin_bit_cast(float, b); }
// monotone integer key of a float
DIAL_DEV int fkey(float x) {
  const int b = fbits(x + 0.f);   // -0 + 0 = +0: one key for both zeros
  return b ^ ((b >> 31) & 0x7fffffff);
}
// the same key carried in a float-typed register (per-lane results travel as vfloat)
DIAL_DEV float fkeyf(float x) { return bitsf(fkey(x)); }

struct LsPt { int alpha, nalpha, cost, d0; };

DIAL_DEV LsPt ls_pick(bool c, const LsPt& a, const LsPt& b) {
  LsPt r;
  r.alpha = c ? a.alpha : b.alpha; r.nalpha = c ? a.nalpha : b.nalpha; r.cost = c ? a.cost : b.cost; r.d0 = c ? a.d0 : b.d0;
  return r;
}

// opening bracket: p0 = point at 0, p1 = its Newton step
DIAL_DEV void ls_open(const LsPt& p0, const LsPt& p1, LsPt& lo, LsPt& hi) {
  const bool lesser = p1.d0 < p0.d0;
  lo = ls_pick(lesser, p1, p0);
  hi = ls_pick(lesser, p0, p1);
}

// done-test of an iteration (ls_iter < max_ls is checked by the caller): kg = fkey(gtol), kng = fkey(-gtol)
DIAL_DEV bool ls_converged(const LsPt& lo, const LsPt& hi, int kg, int kng) {
  return ((lo.d0 < 0) & (lo.d0 > kng)) | ((hi.d0 > 0) & (hi.d0 < kg));
}

// The same test as two unsigned range compares (s_sub + s_cmp_lt_u32 each, straight into a branch), where the form above
// materialises four compares as 64-bit masks: lo.d0 in (kng, 0)  <=>  lo.d0 - (kng + 1) <u -(kng + 1),  hi.d0 in (0, kg)  <=>
// hi.d0 - 1 <u kg - 1; an empty interval (gtol = 0) gets range 0.
struct LsGate { int lo_off; unsigned lo_rng, hi_rng; };
DIAL_DEV LsGate ls_gate(int kg, int kng) {
  LsGate g;
  g.lo_off = kng + 1;
  g.lo_rng = kng < 0 ? (unsigned)(-(kng + 1)) : 0u;
  g.hi_rng = kg > 0 ? (unsigned)(kg - 1) : 0u;
  return g;
}
DIAL_DEV bool ls_converged_lo(const LsPt& lo, const LsGate& g) { return (unsigned)lo.d0 - (unsigned)g.lo_off < g.lo_rng; }
DIAL_DEV bool ls_converged_hi(const LsPt& hi, const LsGate& g) { return (unsigned)hi.d0 - 1u < g.hi_rng; }

// one bracket update; returns whether any end moved (`swap` of the reference)
DIAL_DEV bool ls_update(bool rule_swap, LsPt& lo, LsPt& hi, const LsPt& lo_next, const LsPt& hi_next, const LsPt& mid) {
  if (rule_swap) {   // MJX <= 3.1.3
    const bool swap_lo_next = (lo.d0 > 0) | (lo.d0 < lo_next.d0);
    lo = ls_pick(swap_lo_next, lo_next, lo);
    const bool swap_lo_mid = (mid.d0 < 0) & (lo.d0 < mid.d0);
    lo = ls_pick(swap_lo_mid, mid, lo);
    const bool swap_hi_next = (hi.d0 < 0) | (hi.d0 > hi_next.d0);
    hi = ls_pick(swap_hi_next, hi_next, hi);
    const bool swap_hi_mid = (mid.d0 > 0) & (hi.d0 > mid.d0);
    hi = ls_pick(swap_hi_mid, mid, hi);
    return swap_lo_next | swap_lo_mid | swap_hi_next | swap_hi_mid;
  }
  // MJX >= 3.1.4 `_in_bracket`: y replaces the bracket end x only if it lies on the same side of the minimum and closer
  // to it; each end is offered its own Newton step, the mid-point and the other end's Newton step
  const auto in_bracket = [](const LsPt& x, const LsPt& y) { return ((x.d0 < y.d0) & (y.d0 < 0)) | ((x.d0 > y.d0) & (y.d0 > 0)); };
  const bool s1 = in_bracket(lo, lo_next);
  lo = ls_pick(s1, lo_next, lo);
  const bool s2 = in_bracket(lo, mid);
  lo = ls_pick(s2, mid, lo);
  const bool s3 = in_bracket(lo, hi_next);
  lo = ls_pick(s3, hi_next, lo);
  const bool s4 = in_bracket(hi, hi_next);
  hi = ls_pick(s4, hi_next, hi);
  const bool s5 = in_bracket(hi, mid);
  hi = ls_pick(s5, mid, hi);
  const bool s6 = in_bracket(hi, lo_next);
  hi = ls_pick(s6, lo_next, hi);
  return s1 | s2 | s3 | s4 | s5 | s6;
}

// One bracket update with LAZY fetches; returns whether any end moved (`swap` of the reference).  The three trial points
// of the iteration stay where they were evaluated -- `pk[4]` holds (alpha, nalpha, cost key, d0 key) of group g's point
// in the lanes of group g (lane_of: lo_next / hi_next / mid -> first lane of its group) -- and only their three slope keys
// are broadcast up front.  The update runs on the slope keys alone and yields, per bracket end, WHICH point it ends up
// with; the winner's other three words are fetched afterwards with a lane-indexed v_readlane.  (Carrying all 12 words
// of the three candidates through the select chain cost ~28 live SGPRs and four s_cselect per pick: the kernels sit at
// the SGPR limit, and every SGPR spilled to a VGPR lane is a v_writelane / v_readlane pair on the VALU.)
// MINMAX = false keeps the boolean form of `_in_bracket` (the same decisions): the capacity-dimension kernel's translation unit trips an
// LLVM address-space bug ("Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base", cf. rollout_body.h: the overflow workspace)
// whenever this function's shape changes; its call sites pass !D::gen.
template <bool MINMAX = true, class BC>
DIAL_DEV bool ls_update_lazy(bool rule_swap, LsPt& lo, LsPt& hi, int k_lo_next, int k_hi_next, int k_mid, int lane_lo_next,
                             int lane_hi_next, int lane_mid, BC&& fetch /* (word 0..2, lane) -> int */) {
  // ("none" is -2, not -1: `taken ? 0 : -1` is a sign extension of the condition, which the compiler routes through the VALU --
  //  s_cselect_b64, v_cndmask, v_readfirstlane -- where `taken ? 0 : -2` is one s_cselect_b32)
  int lo_d0 = lo.d0, hi_d0 = hi.d0, lo_sel = -2, hi_sel = -2;
  bool any;
  if (rule_swap) {   // MJX <= 3.1.3
    const bool swap_lo_next = (lo_d0 > 0) | (lo_d0 < k_lo_next);
    lo_sel = swap_lo_next ? lane_lo_next : lo_sel; lo_d0 = swap_lo_next ? k_lo_next : lo_d0;
    const bool swap_lo_mid = (k_mid < 0) & (lo_d0 < k_mid);
    lo_sel = swap_lo_mid ? lane_mid : lo_sel; lo_d0 = swap_lo_mid ? k_mid : lo_d0;
    const bool swap_hi_next = (hi_d0 < 0) | (hi_d0 > k_hi_next);
    hi_sel = swap_hi_next ? lane_hi_next : hi_sel; hi_d0 = swap_hi_next ? k_hi_next : hi_d0;
    const bool swap_hi_mid = (k_mid > 0) & (hi_d0 > k_mid);
    hi_sel = swap_hi_mid ? lane_mid : hi_sel; hi_d0 = swap_hi_mid ? k_mid : hi_d0;
    any = swap_lo_next | swap_lo_mid | swap_hi_next | swap_hi_mid;
  } else if constexpr (!MINMAX) {
    const auto in_bracket = [](int x, int y) { return ((x < y) & (y < 0)) | ((x > y) & (y > 0)); };
    const bool s1 = in_bracket(lo_d0, k_lo_next);
    lo_sel = s1 ? lane_lo_next : lo_sel; lo_d0 = s1 ? k_lo_next : lo_d0;
    const bool s2 = in_bracket(lo_d0, k_mid);
    lo_sel = s2 ? lane_mid : lo_sel; lo_d0 = s2 ? k_mid : lo_d0;
    const bool s3 = in_bracket(lo_d0, k_hi_next);
    lo_sel = s3 ? lane_hi_next : lo_sel; lo_d0 = s3 ? k_hi_next : lo_d0;
    const bool s4 = in_bracket(hi_d0, k_hi_next);
    hi_sel = s4 ? lane_hi_next : hi_sel; hi_d0 = s4 ? k_hi_next : hi_d0;
    const bool s5 = in_bracket(hi_d0, k_mid);
    hi_sel = s5 ? lane_mid : hi_sel; hi_d0 = s5 ? k_mid : hi_d0;
    const bool s6 = in_bracket(hi_d0, k_lo_next);
    hi_sel = s6 ? lane_lo_next : hi_sel; hi_d0 = s6 ? k_lo_next : hi_d0;
    any = s1 | s2 | s3 | s4 | s5 | s6;
  } else {
    // MJX >= 3.1.4 `_in_bracket`: y replaces the bracket end x only if it lies on the same side of the minimum and closer
    // to it -- ((x < y) & (y < 0)) | ((x > y) & (y > 0)); each end is offered its own Newton step, the mid-point and the other end's
    // Newton step, in that order.  On the integer keys "x after the offer" is max(x, y) for y < 0, min(x, y) for y > 0 and x for
    // y = 0, i.e. min(max(x, a), b) with a = y < 0 ? y : INT_MIN, b = y > 0 ? y : INT_MAX (neither sentinel is the key of a
    // non-NaN float), and the offer was taken exactly when that differs from x: two compares + two selects per CANDIDATE and
    // s_max / s_min / s_cmp_lg / s_cselect per offer -- 39 scalar instructions for the six offers where the boolean form (every
    // compare materialised as a 64-bit mask, two s_and and an s_or per offer) took 85 (build/isa/DimsAllegro_9_3_false.s, round 5).
    // The decisions are the same for every triple of keys (tests/test_ls_bracket.py runs both forms over edge and random keys).
    constexpr int KMIN = (int)0x80000000u, KMAX = 0x7fffffff;
    const int a1 = k_lo_next < 0 ? k_lo_next : KMIN, b1 = k_lo_next > 0 ? k_lo_next : KMAX;
    const int a2 = k_mid < 0 ? k_mid : KMIN, b2 = k_mid > 0 ? k_mid : KMAX;
    const int a3 = k_hi_next < 0 ? k_hi_next : KMIN, b3 = k_hi_next > 0 ? k_hi_next : KMAX;
    const auto offer = [](int& x, int& sel, int a, int b, int lane) {
      const int m = x > a ? x : a, n = m < b ? m : b;
      sel = n != x ? lane : sel;
      x = n;
    };
    offer(lo_d0, lo_sel, a1, b1, lane_lo_next);
    offer(lo_d0, lo_sel, a2, b2, lane_mid);
    offer(lo_d0, lo_sel, a3, b3, lane_hi_next);
    offer(hi_d0, hi_sel, a3, b3, lane_hi_next);
    offer(hi_d0, hi_sel, a2, b2, lane_mid);
    offer(hi_d0, hi_sel, a1, b1, lane_lo_next);
    any = (lo_sel >= 0) | (hi_sel >= 0);
  }
  const bool lo_new = lo_sel >= 0, hi_new = hi_sel >= 0;
  const int ll = lo_new ? lo_sel : 0, hl = hi_new ? hi_sel : 0;
  const int la = fetch(0, ll), ln = fetch(1, ll), lc = fetch(2, ll);
  const int ha = fetch(0, hl), hn = fetch(1, hl), hc = fetch(2, hl);
  lo.alpha = lo_new ? la : lo.alpha; lo.nalpha = lo_new ? ln : lo.nalpha; lo.cost = lo_new ? lc : lo.cost; lo.d0 = lo_d0;
  hi.alpha = hi_new ? ha : hi.alpha; hi.nalpha = hi_new ? hn : hi.nalpha; hi.cost = hi_new ? hc : hi.cost; hi.d0 = hi_d0;
  return any;
}

// result of the search: improved?  and the step of the better end
DIAL_DEV bool ls_result(const LsPt& p0, const LsPt& lo, const LsPt& hi, float& alpha) {
  alpha = bitsf(lo.cost < hi.cost ? lo.alpha : hi.alpha);
  return (lo.cost < p0.cost) | (hi.cost < p0.cost);
}

// lane-wise: pack a point's four words from its float values (alpha, cost, slope d0, curvature d1 != 0)
DIAL_DEV void ls_pack(float alpha, float cost, float d0, float d1, float& o_alpha, float& o_nalpha, float& o_cost, float& o_d0) {
  o_alpha = alpha;
  o_nalpha = alpha - d0 / d1;
  o_cost = fkeyf(cost);
  o_d0 = fkeyf(d0);
}

}  // namespace dial
