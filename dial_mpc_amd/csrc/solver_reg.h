// solver_reg.h -- register-resident Newton solver (solver.solve of MJX) for the dimension-specialised
// instantiations (all of which use the square LDS layout, cmodel.h: Dims::square).
//
// Lane roles inside the wavefront that owns the sample:
//   lanes [0, NV)            dof i  -- and the joint-limit row of dof i, if it has one ("row slot")
//   lanes [32, 32 + 4*NC)    pyramid edge e of contact c (lane 32 + 4c + e)     ("row slot")
// Persistent registers:
//   R[NV]   dof lane i: row i of M          contact lane: the constraint row J_r = Jn +- mu*Jt
//   per dof lane: qfs, qas, qacc, Ma, grad, search, mv     per row slot: D, aref, lsign, Jaref, jv
// One sweep  acc += R[j] * readlane(v, j), j < NV  yields M v in the dof lanes AND J v in the contact lanes;
// J^T f is formed from the dof-major pyramid rows in LDS (one 16-byte fetch per contact) and 4*NC readlanes;
// reductions are DPP butterflies, independent ones batched with interleaved stages.  LDS is used to assemble
// H = M + J^T D J (contact-sparse work list), for the transposed copy of the factor inside the L D L^T solve
// and to re-layout the constraint rows for the line search.
//
// Cost model (tools/ubench/latency.hip, profiles/r01_ubench_latency.txt): an independent VALU instruction issues
// every ~2 cycles, a dependent one every ~10.6, v_readlane -> VALU ~20, a dependent DPP add ~16, an LDS round
// trip ~60: the code below is arranged so that independent chains sit next to each other.
//
// Semantics are those of rollout_body.h's LDS solver (same iterates: warm-start choice, MJX line search
// with <= ls_iterations bracket refinements, skipped final factorisation); the wave emulator runs both
// against the oracle.
#pragma once
// (included from rollout_body.h after msym / tri_idx / kbi are defined)

namespace dial {

// Register-resident sparse L D L^T solve, x = A^-1 b with b / x in registers (lane i <-> dof i).
//
// Layout: lane l owns row l of the REVERSED matrix A' = P A P^T (P = order reversal), i.e. dof N-1-l.
// Eliminating the dofs leaves-first (MuJoCo's L^T D L order) produces no fill-in: L'[j'][k'] != 0 only if dof
// j is an ancestor of dof k, which is known at compile time (D::Topo) -- the (k', j') update is simply not
// emitted otherwise (Go2: 99 of 153 pairs, H1: 169 of 300).  Per step k': pivot and column entries are
// broadcast with v_readlane (wave-uniform scalars) and every lane updates its own row.  Both substitutions
// are column oriented (broadcast one entry, one FMA per lane, unit diagonals so no per-step scaling): the
// forward one is fused into the factorisation, the backward one runs on the rows of L'^T, which are fetched
// from an LDS copy of the factor (N ds_writes, N/4 ds_read_b128 per lane; `scratch` needs N * kCholStride<N>
// floats and may be the storage of A itself).  A row-oriented backward pass needs no transpose but costs one
// full DPP wave reduction per unknown -- 3x the instructions.  Right-hand side and solution are lane-reversed
// with one ds_bpermute each.
template <int N> constexpr int kCholStride = (N + 3) & ~3;

// Elimination sequence: any leaves-first order is fill-free; taking the dofs by decreasing depth interleaves the
// branches (Go2: the four calves, then the four thighs, the four hips, then the base chain), so consecutive
// columns -- pivot reciprocal, scaling, updates, forward-substitution step -- are independent of each other and
// their ~10-cycle dependent-issue latencies overlap.  seq[t] is the reversed index k' handled at step t.
template <class Topo, int N>
struct ElimOrder {
  int seq[N > 0 ? N : 1] = {};
  int nlevel = 0;
  int lvl[N > 0 ? N + 1 : 2] = {};   // steps lvl[g] .. lvl[g+1]-1 handle the dofs of one depth: mutually independent
  constexpr ElimOrder() {
    int depth[N > 0 ? N : 1] = {};
    int maxd = 0;
    for (int i = 0; i < N; i++) {
      int d = 0;
      for (int j = 0; j < i; j++) d += Topo::anc(i, j) ? 1 : 0;
      depth[i] = d;
      maxd = d > maxd ? d : maxd;
    }
    int t = 0;
    for (int d = maxd; d >= 0; d--) {
      lvl[nlevel++] = t;
      for (int i = N - 1; i >= 0; i--)
        if (depth[i] == d) seq[t++] = N - 1 - i;
    }
    lvl[nlevel] = t;
  }
};

// compile-time list of the reversed column indices j' > k' whose dof is an ancestor of dof k = N-1-k'
template <class Topo, int N, int KP>
struct AncList {
  int jp[N > 0 ? N : 1] = {};
  int n = 0;
  constexpr AncList() {
    for (int q = KP + 1; q < N; q++)
      if (Topo::anc(N - 1 - KP, N - 1 - q)) jp[n++] = q;
  }
};

// TopoT: the fill pattern of A -- the robot's dof tree for M (and M + dt B), TopoDense for a matrix that couples all dofs
// dinv_io: where the pivots' reciprocals go (lane = reversed dof), for a later REUSE solve.
// REUSE: a second right-hand side on the factor the previous call left in `scratch` (nothing has written there since) and the
// reciprocals it returned: the unit columns are re-read from the LDS copy (N strided fetches per lane), forward substitution in
// the same order with the same operands -- bit for bit what a second factorisation of the same matrix gives -- then the shared
// backward pass.  (The Newton solver's second iteration when the active set did not change: H is the same matrix.)
template <class D, class TopoT = typename D::Topo, bool REUSE = false, class W, class M>
DIAL_DEV vfloat reg_chol_solve_v(W& w, const M* m, const float* A, vfloat bvec, float* scratch, vfloat* dinv_io = nullptr) {
  constexpr int N = D::NV, S = kCholStride<N>;
  using Topo = TopoT;
  w.begin_region();
  vfloat a[N];
  static_assert(D::square, "the register solver reads M / H from the square LDS layout");
  const auto own_i = [&](int l) { return N - 1 - l; };   // the lane's dof
  (void)m;
  vfloat b = w.lane_reverse(bvec, N);
  vfloat dinv = vsplat(0.f);
  constexpr ElimOrder<Topo, N> EO{};
  if constexpr (REUSE) {
    (void)A;
    dinv = *dinv_io;
    // column k' of L' (unit lower, 0 in lanes <= k') sits in row N-1-k' of the LDS copy, this lane's entry at its own dof
    static_for<0, N>([&](auto KP) {
      constexpr int kp = KP;
      a[kp] = w.per_lane([&](int l) { return scratch[(N - 1 - kp) * S + (l < N ? own_i(l) : 0)]; });
    });
    static_for<0, EO.nlevel>([&](auto LV) {
      constexpr int l0 = EO.lvl[LV], l1 = EO.lvl[LV + 1];
      float bk[l1 - l0 > 0 ? l1 - l0 : 1];
      static_for<l0, l1>([&](auto STEP) { bk[STEP - l0] = bcast(b, EO.seq[STEP]); });
      static_for<l0, l1>([&](auto STEP) { b = b - a[EO.seq[STEP]] * bk[STEP - l0]; });
    });
  } else {
  // A is a full symmetric square with exact zeros off the sparsity pattern: lane l fetches row N-1-l with
  // S/4 ds_read_b128.  Entries above the diagonal of A' (and everything in lanes >= N) are never used: a
  // column is masked when it is finalised, broadcasts only read lanes k' <= l < N.
  static_for<0, S / 4>([&](auto Q) {
    constexpr int q = Q;
    vfloat t[4];
    // (large models: the row addresses from the region's own copy of the lane id -- hoisted out of the T-step loop they are
    //  S / 4 more VGPRs that live through the whole kernel; the H1's kernel spilled them)
    if constexpr (N > 20) w.per_lane4_r([&](int l) { return A + (l < N ? own_i(l) : 0) * S + 4 * q; }, t[0], t[1], t[2], t[3]);
    else w.per_lane4([&](int l) { return A + (l < N ? own_i(l) : 0) * S + 4 * q; }, t[0], t[1], t[2], t[3]);
    static_for<0, 4>([&](auto E) {
      constexpr int j = 4 * q + E;
      if constexpr (j < N) a[N - 1 - j] = t[E];
    });
  });
  static_for<0, EO.nlevel>([&](auto LV) {
    constexpr int l0 = EO.lvl[LV], l1 = EO.lvl[LV + 1];
    float bk[l1 - l0 > 0 ? l1 - l0 : 1];
    static_for<l0, l1>([&](auto STEP) {
      constexpr int kp = EO.seq[STEP];
      const vfloat col = a[kp];                                   // d_k l_ik (unscaled column, lanes >= k')
      const float rinv = fast_rcp(bcast(col, kp));
      const vfloat lik = vsel(w.lane_gt(kp), col * rinv, vsplat(0.f));   // unit lower column: 0 in lanes <= k'
      a[kp] = lik;
      dinv = vsel(w.lane_eq(kp), vsplat(rinv), dinv);
      bk[STEP - l0] = bcast(b, kp);   // forward substitution L' z = b, fused in: the columns of one depth do not touch
                                      // each other's b entries, so all their broadcasts precede the updates of b
      // l'_{j'k'} != 0 only when dof j is an ancestor of dof k.  The broadcasts run two updates ahead of the FMAs
      // that consume them, so that the v_readlane -> VALU scalar-operand hazard is covered by useful work
      constexpr AncList<Topo, N, kp> L{};
      float sb[3] = {0.f, 0.f, 0.f};
      static_for<0, L.n + 2>([&](auto IDX) {
        constexpr int idx = IDX;
        if constexpr (idx < L.n) sb[idx % 3] = bcast(col, L.jp[idx]);
        if constexpr (idx >= 2) { constexpr int jp = L.jp[idx - 2]; a[jp] = a[jp] - lik * sb[(idx - 2) % 3]; }
      });
    });
    static_for<l0, l1>([&](auto STEP) { b = b - a[EO.seq[STEP]] * bk[STEP - l0]; });
  });
  // LDS copy of the factor in ORIGINAL dof order: scratch[k * S + i] = L'[i'][k'] (i = N-1-i', k = N-1-k'), so the
  // lane of dof k finds row k' of L'^T contiguously.  Stored this way the non-zeros fall on the sparsity pattern
  // of A (i an ancestor of k) and everything else is written as an exact 0 -- which is what lets the square
  // layout reuse the storage of H for it: the next assembly of H only rewrites the pattern.
  w.items(N, [&](int l) {
    static_for<0, N>([&](auto KP) { constexpr int kp = KP; scratch[(N - 1 - kp) * S + own_i(l)] = lane_val(a[kp], l); });
  });
  if (dinv_io) *dinv_io = dinv;
  }   // (!REUSE)
  vfloat x = b * dinv;
  // backward substitution L'^T x = D^-1 z: lane i' needs u[j'] = L'[j'][i'] = scratch[i * S + j] (0 unless j' > i'):
  // its row of the LDS copy, fetched into the registers the factor no longer needs; steps in reverse elimination
  // order (root first, then the branches interleaved)
  static_for<0, S / 4>([&](auto Q) {
    constexpr int q = Q;
    vfloat t[4];
    if constexpr (N > 20) w.per_lane4_r([&](int l) { return scratch + (l < N ? own_i(l) : 0) * S + 4 * q; }, t[0], t[1], t[2], t[3]);
    else w.per_lane4([&](int l) { return scratch + (l < N ? own_i(l) : 0) * S + 4 * q; }, t[0], t[1], t[2], t[3]);
    static_for<0, 4>([&](auto E) {
      constexpr int j = 4 * q + E;
      if constexpr (j < N) a[N - 1 - j] = t[E];
    });
  });
  static_for<0, EO.nlevel>([&](auto LVR) {
    constexpr int lv = EO.nlevel - 1 - LVR, l0 = EO.lvl[lv], l1 = EO.lvl[lv + 1];
    float xk[l1 - l0 > 0 ? l1 - l0 : 1];
    static_for<l0, l1>([&](auto STEP) { xk[STEP - l0] = bcast(x, EO.seq[STEP]); });
    static_for<l0, l1>([&](auto STEP) { x = x - a[EO.seq[STEP]] * xk[STEP - l0]; });
  });
  return w.lane_reverse(x, N);
}

// The same solve with DPP-OPERAND broadcasts (round 5): the rows live in lanes 0 .. N - 1 <= 31, i.e. in the first two DPP rows, so
// one v_permlane16_swap per vector (wave.h: dup_rows) puts both rows within reach of every consumer lane's row_newbcast, and on the
// product build the broadcast is an operand of the consuming v_fmac_f32 / v_rcp_f32 itself (wave.h: fma_pick) -- one instruction
// where reg_chol_solve_v issues v_readlane + the SGPR-operand hazard + v_fma.  Same elimination order, same operations on the
// same operands: bit-identical results.  Written for the two-rollouts-per-wavefront kernel (every 32-lane half runs its own
// solve; nothing here leaves the half) and used by every register solver since (reg_chol below).
template <class D, class TopoT = typename D::Topo, bool REUSE = false, class W, class M>
DIAL_DEV vfloat reg_chol_solve2(W& w, const M* m, const float* A, vfloat bvec, float* scratch, vfloat* dinv_io = nullptr) {
  constexpr int N = D::NV, S = kCholStride<N>;
  using Topo = TopoT;
  static_assert(N <= 32 && D::square, "the rows fit the first two DPP rows of a 32-lane half");
#ifdef DIAL_PAIR_SOLVER_SCOPE
  DIAL_LANE_SCOPE(w);
#endif
  w.begin_region();
  vfloat a[N];
  const auto own_i = [&](int l) { return N - 1 - l; };   // the lane's dof
  (void)m;
  vfloat b = w.lane_reverse(bvec, N);
  vfloat dinv = vsplat(0.f);
  constexpr ElimOrder<Topo, N> EO{};
  if constexpr (REUSE) {
    (void)A;
    dinv = *dinv_io;
    static_for<0, N>([&](auto KP) {
      constexpr int kp = KP;
      a[kp] = w.per_lane([&](int l) { return scratch[(N - 1 - kp) * S + (l < N ? own_i(l) : 0)]; });
    });
    static_for<0, EO.nlevel>([&](auto LV) {
      constexpr int l0 = EO.lvl[LV], l1 = EO.lvl[LV + 1];
      vfloat bX, bY;
      w.dup_rows(b, bX, bY);
      static_for<l0, l1>([&](auto STEP) { constexpr int kp = EO.seq[STEP]; b = w.template fnma_pick<kp>(b, bX, bY, a[kp]); });
    });
  } else {
  static_for<0, S / 4>([&](auto Q) {
    constexpr int q = Q;
    vfloat t[4];
    w.per_lane4([&](int l) { return A + (l < N ? own_i(l) : 0) * S + 4 * q; }, t[0], t[1], t[2], t[3]);
    static_for<0, 4>([&](auto E) {
      constexpr int j = 4 * q + E;
      if constexpr (j < N) a[N - 1 - j] = t[E];
    });
  });
  static_for<0, EO.nlevel>([&](auto LV) {
    constexpr int l0 = EO.lvl[LV], l1 = EO.lvl[LV + 1];
    vfloat bX, bY;
    w.dup_rows(b, bX, bY);   // (the columns of one depth do not touch each other's b entries: one swap per level)
    static_for<l0, l1>([&](auto STEP) {
      constexpr int kp = EO.seq[STEP];
      const vfloat col = a[kp];                                   // d_k l_ik (unscaled column, lanes >= k')
      vfloat cX, cY;
      w.dup_rows(col, cX, cY);
      const vfloat rinv = w.template rcp_pick<kp>(cX, cY);
      const vfloat lik = vsel(w.lane_gt(kp), col * rinv, vsplat(0.f));   // unit lower column: 0 in lanes <= k'
      a[kp] = lik;
      dinv = vsel(w.lane_eq(kp), rinv, dinv);
      constexpr AncList<Topo, N, kp> L{};
      static_for<0, L.n>([&](auto IDX) {
        constexpr int jp = L.jp[IDX];
        a[jp] = w.template fnma_pick<jp>(a[jp], cX, cY, lik);
      });
    });
    // forward substitution L' z = b: the level's pivots' entries of b were duplicated before its columns were touched
    static_for<l0, l1>([&](auto STEP) { constexpr int kp = EO.seq[STEP]; b = w.template fnma_pick<kp>(b, bX, bY, a[kp]); });
  });
  w.items(N, [&](int l) {
    static_for<0, N>([&](auto KP) { constexpr int kp = KP; scratch[(N - 1 - kp) * S + own_i(l)] = lane_val(a[kp], l); });
  });
  if (dinv_io) *dinv_io = dinv;
  }   // (!REUSE)
  vfloat x = b * dinv;
  static_for<0, S / 4>([&](auto Q) {
    constexpr int q = Q;
    vfloat t[4];
    w.per_lane4([&](int l) { return scratch + (l < N ? own_i(l) : 0) * S + 4 * q; }, t[0], t[1], t[2], t[3]);
    static_for<0, 4>([&](auto E) {
      constexpr int j = 4 * q + E;
      if constexpr (j < N) a[N - 1 - j] = t[E];
    });
  });
  static_for<0, EO.nlevel>([&](auto LVR) {
    constexpr int lv = EO.nlevel - 1 - LVR, l0 = EO.lvl[lv], l1 = EO.lvl[lv + 1];
    vfloat xX, xY;
    w.dup_rows(x, xX, xY);
    static_for<l0, l1>([&](auto STEP) { constexpr int kp = EO.seq[STEP]; x = w.template fnma_pick<kp>(x, xX, xY, a[kp]); });
  });
  return w.lane_reverse(x, N);
}


// Which formulation: measured per robot on one box (profiles/r05_ab_chol_dpp.txt, rollout kernel ms, DPP vs v_readlane): H1 jog
// 0.864 vs 0.879, H1 loco 0.510 vs 0.514, crate climb 1.119 vs 1.130, push crate 1.581 vs 1.630, Allegro within noise -- and the
// Go2's one-rollout-per-wavefront kernel 0.386 vs 0.380: the headline batch is as much bound by the chain as by issue slots, and
// v_permlane16_swap (quarter rate) sits on every pivot's chain.  So: DPP everywhere but there.
#ifndef DIAL_CHOL_DPP
#define DIAL_CHOL_DPP 1   // A/B switch: 0 = the v_readlane formulation (reg_chol_solve_v) for all one-rollout-per-wavefront kernels
#endif
template <class D, class TopoT = typename D::Topo, bool REUSE = false, class W, class M>
DIAL_DEV vfloat reg_chol(W& w, const M* m, const float* A, vfloat bvec, float* scratch, vfloat* dinv_io = nullptr) {
  constexpr bool go2_one = std::is_same<D, DimsGo2>::value && !W::half2;
  if constexpr (W::half2 || (DIAL_CHOL_DPP && !go2_one)) return reg_chol_solve2<D, TopoT, REUSE>(w, m, A, bvec, scratch, dinv_io);
  else return reg_chol_solve_v<D, TopoT, REUSE>(w, m, A, bvec, scratch, dinv_io);
}

template <class W, class M>
DIAL_DEV void solver_reg(W& w, const M* m, const Ws& s) {
  constexpr int NV = M::D::NV, NC = M::D::NC, NL = M::D::NL;
  constexpr int C0 = 32;
  static_assert(NV <= 32 && 4 * NC <= 32, "lane layout needs nv <= 32 and 4*ncon <= 32");
  w.begin_region();
  const vbool isdof = w.lane_lt(NV);
  const vfloat vzero = vsplat(0.f);

  // ---- row-slot index of every lane (limit row of the dof, or contact edge row), -1 = none
  auto row_of = [&](int l) -> int {
    if (l < NV) return m->dof_limrow[l];
    if (l >= C0 && l < C0 + 4 * NC) return NL + (l - C0);
    return -1;
  };
  // ---- persistent registers
  vfloat R[NV];
  {
    // one strided fetch per register: dof lane i walks row i of the square M (stride 1), contact lane r walks
    // column r of the dof-major pyramid Jacobian (stride T); idle lanes re-read a word that holds 0 (lsign of a contact row)
    constexpr int S = M::D::S, T = M::D::T;
    const auto fetch = [&](int l, int j) {
      // (offsets from one base pointer rather than a select of pointers: keeps the accesses in the LDS address space)
      const int ojc = (int)(s.Jc - s.M), ozero = (int)(s.lsign - s.M) + NL;   // lsign of a contact row is 0
      const int off = l < NV ? l * S : ((l >= C0 && l < C0 + 4 * NC) ? ojc + (l - C0) : ozero);
      const int stride = l < NV ? 1 : ((l >= C0 && l < C0 + 4 * NC) ? T : 0);
      return s.M[off + j * stride];
    };
#pragma unroll
    for (int j = 0; j < NV; j++) {
      // large models: keep the NV strided addresses from being hoisted out of the step loop (they would be spilled)
      if constexpr (NV > 20) R[j] = w.per_lane_r([&](int l) { return fetch(l, j); });
      else R[j] = w.per_lane([&](int l) { return fetch(l, j); });
    }
  }
  const vfloat vD = w.per_lane([&](int l) { int r = row_of(l); return r >= 0 ? s.D[r] : 0.f; });
  const vfloat varef = w.per_lane([&](int l) { int r = row_of(l); return r >= 0 ? s.aref[r] : 0.f; });
  const vfloat vls = w.per_lane([&](int l) { int r = l < NV ? m->dof_limrow[l] : -1; return r >= 0 ? s.lsign[r] : 0.f; });
  const vfloat vqfs = w.per_lane([&](int l) { return l < NV ? s.qfs[l] : 0.f; });
  const vfloat vqas = w.per_lane([&](int l) { return l < NV ? s.qas[l] : 0.f; });
  const vfloat vwarm = w.per_lane([&](int l) { return l < NV ? s.warm[l] : 0.f; });

  // acc[dof lane i] = (M v)_i, acc[contact lane r] = (J v)_r
  // (a dependent fp32 FMA issues every ~10 cycles on gfx950, an independent one every 2: three partial sums;
  //  the broadcasts run two FMAs ahead of their use)
  // DIAL_SWEEP_DPP (round 6, default): the entries of v reach every lane as DPP operands of the multiply-adds -- one v_permlane32_swap +
  // one v_permlane16_swap put v's lanes 0..31 into every row (wave.h: dup_halves, dup_rows), then one v_fmac_f32_dpp per column --
  // instead of one v_readlane (-> SGPR) + one v_fmac per column: 20 VALU instructions per sweep for 36, and an SGPR-operand FMA
  // issues no faster than a DPP one (profiles/r06_ubench_issue.txt: 4.2 vs 4.4 cycles, v_readlane 4.5).  Same products, same three
  // partial sums, same order: bit-identical.
#ifndef DIAL_SWEEP_DPP
#define DIAL_SWEEP_DPP 1
#endif
  auto dotR = [&](const vfloat& v) {
    vfloat acc[3] = {vzero, vzero, vzero};
#if DIAL_SWEEP_DPP
    vfloat vlo, vhi, X, Y;
    w.dup_halves(v, vlo, vhi);
    w.dup_rows(vlo, X, Y);
    static_for<0, NV>([&](auto IDX) { constexpr int j = IDX; acc[j % 3] = w.template fma_pick<j>(acc[j % 3], X, Y, R[j]); });
#else
    float sb[3] = {0.f, 0.f, 0.f};
    static_for<0, NV + 2>([&](auto IDX) {
      constexpr int idx = IDX;
      if constexpr (idx < NV) sb[idx % 3] = bcast(v, idx);
      if constexpr (idx >= 2) acc[(idx - 2) % 3] = acc[(idx - 2) % 3] + R[idx - 2] * sb[(idx - 2) % 3];
    });
#endif
    return (acc[0] + acc[1]) + acc[2];
  };
  // row-slot product J_r . v: limit rows are +-e_dof, contact rows come out of the sweep
  auto row_prod = [&](const vfloat& v, const vfloat& sweep) { return vsel(isdof, vls * v, sweep); };
  auto row_cost_terms = [&](const vfloat& ja) { return vsel(vlt0(ja), vD * ja * ja, vzero); };

  // ---- warm-start selection (solver.solve): cost at qacc_warmstart vs cost at qacc_smooth
  const vfloat pW = dotR(vwarm), pS = dotR(vqas);
  const vfloat jaW = row_prod(vwarm, pW) - varef, jaS = row_prod(vqas, pS) - varef;
  const vfloat maW = vsel(isdof, pW, vzero), maS = vsel(isdof, pS, vzero);
  float cw, gw, cs, gs;
  {
    vfloat t[4] = {row_cost_terms(jaW), (maW - vqfs) * (vwarm - vqas), row_cost_terms(jaS), (maS - vqfs) * (vqas - vqas)};
    float r[4];
    w.vsumN(t, r);
    cw = r[0]; gw = r[1]; cs = r[2]; gs = r[3];
  }
  const float cost_w = 0.5f * cw + 0.5f * gw, cost_s = 0.5f * cs + 0.5f * gs;
  const bool use_warm = cost_w < cost_s;
  vfloat vqacc = use_warm ? vwarm : vqas;
  vfloat vMa = use_warm ? maW : maS;
  vfloat vJa = use_warm ? jaW : jaS;
  float cost = use_warm ? cost_w : cost_s;
  float gauss = use_warm ? 0.5f * gw : 0.5f * gs;
  float prev_cost = INFINITY;
  const float scale = 1.f / (m->meaninertia * (float)(NV > 1 ? NV : 1));
  const bool rule_swap = m->ls_rule == DIAL_LS_SWAP;   // fetched once: the constants live in LDS
  const int max_iter = DM_UNIFORM_I(m->iterations), max_ls = DM_UNIFORM_I(m->ls_iterations);
  const float tol = m->tolerance, ls_tol = m->ls_tolerance, meaninertia = m->meaninertia;

#ifdef DIAL_PROFILE
  unsigned long long prof_prev_act = 0;
#endif
  int niter = 0;
  // H = M + J^T diag(D act) J depends on the iterate through the active set alone: when the second Newton iteration finds the
  // set of the first (27 % of the Go2's control steps at N = 2048, profiles/r04_sections_unitree_go2_trot_cycles.txt), its H is
  // the same matrix -- assembly and factorisation are skipped, the factor left in s.H by the first solve is used again
  unsigned long long act_prev = 0;
  vfloat h_dinv = vzero;
  bool h_valid = false;
  for (;;) {
    // ---- _update_constraint: forces; _update_gradient: grad = Ma - qfrc_smooth - J^T f
    const vbool act = vlt0(vJa);
#ifdef DIAL_PROFILE
    {   // how often does the active set (rows with D > 0 and Jaref < 0) survive from one Newton iteration to the next?
      const unsigned long long cur = __builtin_amdgcn_ballot_w64(act && (vD > 0.f));
      static_assert(sizeof(cur) == 8, "");
      if (niter == 1 && w.lane == 0 && w.acc) { w.acc[28] += 1; if (cur == prof_prev_act) w.acc[29] += 1; }
      prof_prev_act = cur;
    }
#endif
    const vfloat vf = vsel(act, vD * (vzero - vJa), vzero);
    vfloat qfc = vls * vf;  // limit row of the own dof
#if DIAL_SWEEP_DPP
    vfloat flo, fhi, fX, fY;   // the contact lanes' forces (lanes 32 ..) within reach of the dof lanes' row broadcasts
    w.dup_halves(vf, flo, fhi);
    w.dup_rows(fhi, fX, fY);
#endif
    static_for<0, NC>([&](auto Cc) {   // J^T f from the dof-major pyramid rows: one 16-byte fetch per contact
      constexpr int c = Cc;
      vfloat jr[4];
      w.per_lane4([&](int l) { return s.Jc + (l < NV ? l : 0) * M::D::T + 4 * c; }, jr[0], jr[1], jr[2], jr[3]);
#if DIAL_SWEEP_DPP
      const vfloat t01 = w.template fma_pick<4 * c + 1>(w.template mul_pick<4 * c>(fX, fY, jr[0]), fX, fY, jr[1]);
      const vfloat t23 = w.template fma_pick<4 * c + 3>(w.template mul_pick<4 * c + 2>(fX, fY, jr[2]), fX, fY, jr[3]);
      qfc = qfc + (t01 + t23);
#else
      qfc = qfc + ((jr[0] * bcast(vf, C0 + 4 * c) + jr[1] * bcast(vf, C0 + 4 * c + 1)) +
                   (jr[2] * bcast(vf, C0 + 4 * c + 2) + jr[3] * bcast(vf, C0 + 4 * c + 3)));
#endif
    });
    const vfloat vgrad = vsel(isdof, vMa - vqfs - qfc, vzero);
    float gn = 0.f;
    if (niter > 0) {
      vfloat t[3] = {row_cost_terms(vJa), (vMa - vqfs) * (vqacc - vqas), vgrad * vgrad};
      float r[3];
      w.vsumN(t, r);
      gauss = 0.5f * r[1];
      prev_cost = cost;
      cost = 0.5f * r[0] + gauss;
      gn = r[2];
    } else if (max_iter != 1) {
      gn = w.vsum(vgrad * vgrad);
    }
    DIAL_MARK(w, 4);
    bool done;
    if (max_iter != 1) {
      const float improvement = scale * (prev_cost - cost), gradient = scale * DM_SQRT(gn);
      done = niter >= max_iter || improvement < tol || gradient < tol;
    } else {
      done = niter >= 1;
    }
    if (done) break;

    // ---- Newton direction: H = M + J^T diag(D*active) J in LDS (lane per entry), Cholesky in registers
    const vfloat vwgt = vsel(act, vD, vzero);
#ifdef DIAL_NO_FACTOR_REUSE
    const bool reuse = false;
#else
    const unsigned long long act_now = w.mask(vlt0(vzero - vwgt));   // rows that carry weight: active and D > 0
    // (also in the 128-VGPR large-batch build, where the reciprocals kept across the line search cost 29 more spilled VGPRs: it
    //  is VALU-issue-bound, and the instructions saved weigh more -- N = 65536 9.38 -> 9.17 ms, N = 8192 unchanged,
    //  profiles/r04_ab_call_x_factor_reuse.txt)
    const bool reuse = h_valid && act_now == act_prev;
    act_prev = act_now;
    h_valid = true;
#endif
    vfloat vsearch;
    if (reuse) {
      vsearch = vzero - reg_chol<typename M::D, typename M::D::Topo, true>(w, m, s.H, vgrad, s.H, &h_dinv);
      DIAL_MARK(w, 5);
    } else {
    {
      // row weights: limit rows at frc[0, NL), contact rows 16-byte aligned at frc[NLP, NLP + 4 NC), then a zero word
      constexpr int S = M::D::S, T = M::D::T, NLP = M::D::NLP, NP = M::D::NHI / 64;
      (void)S;
      w.items(64, [&](int l) {
        const int r = row_of(l);
        if (r >= 0) s.frc[r < NL ? r : NLP + (r - NL)] = lane_val(vwgt, l);
        if (l == 63) s.frc[NLP + 4 * NC] = 0.f;
      });
      // Every lane accumulates its item of every pass (branch-free records: the LDS latencies of the passes
      // overlap), partial sums of split entries are combined inside quads, then one phase writes the entries.
      vfloat part[NP], tot[NP];
      static_for<0, 4>([&](auto Qq) {
        constexpr int q = Qq;
        bool any = false;
        static_for<0, NP>([&](auto PASS) { any = any || q < m->hpass_n[PASS]; });
        if (any) {   // wave-uniform: round q of the contact lists (Go2: one round)
          // issue every 16-byte fetch of the round before the first use
          vfloat ji[NP][4], jj[NP][4], dd[NP][4];
          static_for<0, NP>([&](auto PASS) {
            constexpr int pass = PASS;
            const auto c4 = [&](int l) { return (int)((m->hrec[pass * 64 + l][0] >> (20 + 3 * q)) & 7u) * 4; };
            w.per_lane4([&](int l) { return s.Jc + (m->hrec[pass * 64 + l][0] & 1023u) + c4(l); }, ji[pass][0], ji[pass][1], ji[pass][2], ji[pass][3]);
            w.per_lane4([&](int l) { return s.Jc + ((m->hrec[pass * 64 + l][0] >> 10) & 1023u) + c4(l); }, jj[pass][0], jj[pass][1], jj[pass][2], jj[pass][3]);
            w.per_lane4([&](int l) { return s.frc + NLP + c4(l); }, dd[pass][0], dd[pass][1], dd[pass][2], dd[pass][3]);
          });
          static_for<0, NP>([&](auto PASS) {
            constexpr int pass = PASS;
            const vfloat t = ((ji[pass][0] * dd[pass][0]) * jj[pass][0] + (ji[pass][1] * dd[pass][1]) * jj[pass][1]) +
                             ((ji[pass][2] * dd[pass][2]) * jj[pass][2] + (ji[pass][3] * dd[pass][3]) * jj[pass][3]);
            const vfloat tq = w.per_lane([&](int l) { return q < (int)(m->hrec[pass * 64 + l][1] >> 29) ? lane_val(t, l) : 0.f; });
            if constexpr (q == 0) part[pass] = tq; else part[pass] = part[pass] + tq;
          });
        }
      });
      static_for<0, NP>([&](auto PASS) {
        constexpr int pass = PASS;
        const vfloat pair = part[pass] + w.quad_xor1(part[pass]);
        const vfloat quad = pair + w.quad_xor2(pair);
        // group total by group size (values first, then select: a select between the captured registers'
        // addresses would pin them, the closure and with it the whole workspace descriptor to scratch memory)
        tot[pass] = w.per_lane([&](int l) {
          const float t1 = lane_val(part[pass], l), t2 = lane_val(pair, l), t4 = lane_val(quad, l);
          const uint32_t pc = (m->hrec[pass * 64 + l][1] >> 26) & 3u;
          return pc == 0 ? t1 : (pc == 1 ? t2 : t4);
        });
      });
      w.items(64, [&](int l) {
        static_for<0, NP>([&](auto PASS) {
          constexpr int pass = PASS;
          const uint32_t w1 = m->hrec[pass * 64 + l][1];
          const float v = (s.M[w1 & 1023u] + lane_val(tot[pass], l)) + s.frc[(w1 >> 20) & 63u];
          if ((w1 >> 28) & 1u) {
            s.H[w1 & 1023u] = v;
            s.H[(w1 >> 10) & 1023u] = v;
          }
        });
      });
    }
    DIAL_MARK(w, 5);
    vsearch = vzero - reg_chol<typename M::D>(w, m, s.H, vgrad, s.H, &h_dinv);
    }
    DIAL_MARK(w, 6);

    // ---- solver._linesearch
    w.begin_region();
    const vfloat pv = dotR(vsearch);
    const vfloat vmv = vsel(isdof, pv, vzero);
    const vfloat vjv = row_prod(vsearch, pv);
    float sn2, s1, s2;
    {
      vfloat t[3] = {vsearch * vsearch, vsearch * vMa - vsearch * vqfs, vsearch * vmv};
      float r[3];
      w.vsumN(t, r);
      sn2 = r[0]; s1 = r[1]; s2 = r[2];
    }
    const float smag = DM_SQRT(sn2) * meaninertia * (float)(NV > 1 ? NV : 1);
    const float gtol = tol * ls_tol * smag;
    const float qg0 = gauss, qg1 = s1, qg2 = 0.5f * s2;
    // Line-search layout: lane (g, l) = (lane >> 4, lane & 15) owns rows l, l + 16, .. for trial point g: the
    // three points of one bracketing iteration (lo_next, hi_next, mid) are evaluated by three 16-lane groups in
    // one pass -- per row one FMA, one compare/select and three FMAs into the sums (0.5 D Jaref^2, D jv Jaref,
    // 0.5 D jv^2 over the active rows), three interleaved 16-lane DPP reductions, and the cost / derivatives of
    // the group's point computed lane-wise.  Re-layout through LDS (rows are indexed by r there).
    w.items(64, [&](int l) {
      const int r = row_of(l);
      if (r >= 0) { s.Jaref[r] = lane_val(vJa, l); s.jv[r] = lane_val(vjv, l); }
    });
    constexpr int NE = M::D::NE, RPL = (NE + 15) / 16;   // rows per line-search lane (Go2: 2, H1: 3)
    const vbool g0 = w.lane_lt(16), g01 = w.lane_lt(32);
    vfloat lJa[RPL], ljv[RPL], Q0[RPL], Q1[RPL], Q2[RPL];
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      lJa[q] = w.per_lane([&](int l) { const int r = (l & 15) + 16 * q; return (l < 48 && r < NE) ? s.Jaref[r] : 0.f; });
      ljv[q] = w.per_lane([&](int l) { const int r = (l & 15) + 16 * q; return (l < 48 && r < NE) ? s.jv[r] : 0.f; });
      const vfloat lD = w.per_lane([&](int l) { const int r = (l & 15) + 16 * q; return (l < 48 && r < NE) ? s.D[r] : 0.f; });
      const vfloat dja = lD * lJa[q], djv = lD * ljv[q];
      Q0[q] = (lJa[q] * 0.5f) * dja;
      Q1[q] = ljv[q] * dja;
      Q2[q] = (ljv[q] * 0.5f) * djv;
    }
    // evaluate the points a0, a1, a2 (group g evaluates a_g).  Every lane of a group finishes its point -- cost, slope,
    // the point's own Newton step, and the integer keys of ls_bracket.h -- so that what is broadcast (4 words per
    // point) is all the scalar bracket logic needs
    // pk: the packed words of the three points, each in the lanes of its group (see ls_bracket.h)
    vfloat pk[4];
    auto ls_eval3 = [&](float a0, float a1, float a2) {
      const vfloat va = vsel(g0, vsplat(a0), vsel(g01, vsplat(a1), vsplat(a2)));
      vfloat s0 = vzero, s1v = vzero, s2v = vzero;
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        const vfloat act = vsel(vlt0(lJa[q] + ljv[q] * va), vsplat(1.f), vzero);
        s0 = s0 + act * Q0[q];
        s1v = s1v + act * Q1[q];
        s2v = s2v + act * Q2[q];
      }
      w.row16_sum3(s0, s1v, s2v);
      const vfloat q0 = s0 + vsplat(qg0), q1 = s1v + vsplat(qg1), q2 = s2v + vsplat(qg2);
      const vfloat vcost = (va * va) * q2 + va * q1 + q0;
      const vfloat vd0 = vfma(va * 2.f, q2, q1);   // single rounding: see the line search of rollout_body.h
      const vfloat vd1 = q2 * 2.f + vsel(veq0(q2), vsplat(MJ_MINVAL), vzero);
      w.per_lane_n(pk, [&](int l, float* o) {
        ls_pack(lane_val(va, l), lane_val(vcost, l), lane_val(vd0, l), lane_val(vd1, l), o[0], o[1], o[2], o[3]);
      });
    };
    auto point_at = [&](int lane) {   // all four words of the point held by the group that starts at `lane`
      LsPt p;
      p.alpha = fbits(bcast(pk[0], lane)); p.nalpha = fbits(bcast(pk[1], lane)); p.cost = fbits(bcast(pk[2], lane)); p.d0 = fbits(bcast(pk[3], lane));
      return p;
    };
    ls_eval3(0.f, 0.f, 0.f);
    const LsPt p0 = point_at(0);
    ls_eval3(bitsf(p0.nalpha), bitsf(p0.nalpha), bitsf(p0.nalpha));
    LsPt lo, hi;
    ls_open(p0, point_at(0), lo, hi);
    const int kg = DM_UNIFORM_I(fkey(gtol)), kng = DM_UNIFORM_I(fkey(-gtol));
    // the reference's `while (ls_iter < max) & swap & !converged`, as one scalar compare + branch per condition (a combined
    // `done` flag cost 18 scalar instructions per iteration: every compare became a 64-bit mask)
    const LsGate gate = ls_gate(kg, kng);
    int ls_iter = 0;
    while (ls_iter < max_ls) {
      DM_NOFOLD();
      if (ls_converged_lo(lo, gate)) break;
      DM_NOFOLD();
      if (ls_converged_hi(hi, gate)) break;
      ls_eval3(bitsf(lo.nalpha), bitsf(hi.nalpha), 0.5f * (bitsf(lo.alpha) + bitsf(hi.alpha)));   // groups: lo_next, hi_next, mid
      const bool swap = ls_update_lazy(rule_swap, lo, hi, fbits(bcast(pk[3], 0)), fbits(bcast(pk[3], 16)), fbits(bcast(pk[3], 32)), 0, 16, 32,
                                       [&](int word, int lane) { return fbits(bcast(pk[word], lane)); });
      ls_iter++;
      if (!swap) break;
    }
    float alpha;
    const bool improved = ls_result(p0, lo, hi, alpha);
    if (improved) {
      vqacc = vqacc + vsearch * alpha;
      vMa = vMa + vmv * alpha;
      vJa = vJa + vjv * alpha;
    }
    niter++;
#ifdef DIAL_PROFILE
    if (w.lane == 0 && w.acc) { w.acc[30] += ls_iter; w.acc[31] += 1; }
#endif
    DIAL_MARK(w, 7);
  }
  w.items(NV, [&](int i) {
    const float q = lane_val(vqacc, i);
    s.qacc[i] = q;
    s.warm[i] = q;
  });
  DIAL_MARK(w, 8);
}

}  // namespace dial
